#!/bin/bash
# r06 final: everything the driver line quotes, collected on the FINAL sources (every file carries tools/srchash.py's hash and bench.py refuses figures from
# other sources): 1. the GPU suite; 2. FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes) of three latency-mode Dense-4M
# proofs and of the 2^24 sumcheck; 3. the SQ instruction pass of the cohort regime + the compress probe; 4. the member timing of the diagnostic builds
# (entry -> exit, and the permutation count) -> tail_roofline; 5. rocprofv3 --kernel-trace --stats of the cohort regime; 6. the default bench.
N=${N:-704}  # proofs in flight of the cohort regime that is profiled = bench.py's default
o=gpurun_out/r06_final; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
step() { echo "== $1 ($(date +%H:%M:%S))"; }
step "GPU suite"
timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > $o/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $o/pytest_gpu.txt | tail -3
step "PMC: Dense-4M latency-mode proofs"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $c -d "$R/$o/proof_$c" -o x -- python "$R/tools/proof_only.py" dense_4m 3 > "$R/$o/proof_$c.log" 2>&1; echo "$c rc=$?"
done
step "PMC: 2^24 sumcheck"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $c -d "$R/$o/sc24_$c" -o x -- python "$R/tools/sumcheck24_only.py" 3 > "$R/$o/sc24_$c.log" 2>&1; echo "sc24 $c rc=$?"
done
cd "$R"
f=$(find "$o/proof_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/proof_WRITE_SIZE" -name '*_results.db' | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then
  python tools/pmc_summary.py --after-marker k_merkle_paths --population dense_4m_latency_proofs --units 3 "$f" "$w" "$o/r06_pmc_dense4m_proofs.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/proof_only.py dense_4m 3 (launches after the k_merkle_paths marker: 3 latency-mode proofs, no setup; final build of round 6, tools/r06/final.sh)" > "$o/pmc_dense4m.txt" 2>&1
  cp "$o/r06_pmc_dense4m_proofs.json" profiles/ && echo "pmc dense4m written"
fi
step "PMC: 2^26 sumcheck"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c -d "$R/$o/sc26_$c" -o x -- python "$R/tools/sumcheck24_only.py" 3 26 > "$R/$o/sc26_$c.log" 2>&1; echo "sc26 $c rc=$?"
done
cd "$R"
f=$(find "$o/sc26_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/sc26_WRITE_SIZE" -name '*_results.db' | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then
  python tools/pmc_summary.py --population sumcheck26 --units 3 "$f" "$w" "$o/r06_pmc_sumcheck26.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/sumcheck24_only.py 3 26 (3 repetitions of the 2^26 sumcheck; final build of round 6, tools/r06/final.sh)" k_sc > "$o/pmc_sc26.txt" 2>&1
  cp "$o/r06_pmc_sumcheck26.json" profiles/ && echo "pmc sumcheck26 written"; tail -3 "$o/pmc_sc26.txt" | cut -c1-300
fi
f=$(find "$o/sc24_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/sc24_WRITE_SIZE" -name '*_results.db' | head -1)
if [ -n "$f" ] && [ -n "$w" ]; then
  python tools/pmc_summary.py --population sumcheck24 --units 3 "$f" "$w" "$o/r06_pmc_sumcheck24.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/sumcheck24_only.py 3 (3 repetitions of the 2^24 sumcheck; final build of round 6, tools/r06/final.sh)" k_sc > "$o/pmc_sc24.txt" 2>&1
  cp "$o/r06_pmc_sumcheck24.json" profiles/ && echo "pmc sumcheck24 written"; tail -3 "$o/pmc_sc24.txt" | cut -c1-300
fi
step "diagnostic builds: member timing of k_logup_tail"
DP_LIB_VARIANT=wgtimes DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py ${N} > $o/r06_wgtimes_${N}.txt 2>&1; echo "wgtimes rc=$?"; grep -E "wg-times|proofs/s" $o/r06_wgtimes_${N}.txt | tail -2 | cut -c1-400
DP_LIB_VARIANT=wgphases DP_TIMING=1 timeout -s KILL 300 python tools/archive/conc_hoststats.py ${N} > $o/r06_wgphases_${N}.txt 2>&1; echo "wgphases rc=$?"; grep -E "wg-times" $o/r06_wgphases_${N}.txt | tail -4 | cut -c1-400
grep -E "wg-times|proofs/s|prove_batch:" $o/r06_wgtimes_${N}.txt > $o/r06_wgtimes_${N}_summary.txt; grep -E "wg-times|proofs/s" $o/r06_wgphases_${N}.txt > $o/r06_wgphases_${N}_summary.txt
python tools/tail_roofline.py $o/r06_wgtimes_${N}_summary.txt $o/r06_wgphases_${N}_summary.txt $o/r06_tail_roofline.json && cp $o/r06_tail_roofline.json $o/r06_wgtimes_${N}_summary.txt $o/r06_wgphases_${N}_summary.txt profiles/
step "rocprofv3 kernel trace of the cohort regime"
cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$R/$o/prof" -o b${N} -- python "$R/tools/profile_batch.py" dense_4m ${N} > "$R/$o/prof.log" 2>&1; echo "rocprof rc=$?"
cd "$R"; tail -1 $o/prof.log | cut -c1-200
db=$(find $o/prof -name "*.db" | head -1)
if [ -n "$db" ]; then
  python tools/rocpd_summary.py "$db" $o/r06_bench${N}_kernel_stats.csv > $o/summary.err 2>&1; head -8 $o/r06_bench${N}_kernel_stats.csv | cut -c1-120
  python tools/trace_analyze.py "$db" > $o/r06_trace_analysis_${N}.txt 2>&1; sed -n 1,14p $o/r06_trace_analysis_${N}.txt | cut -c1-160
  python tools/timeline_occupancy.py "$db" 5 > $o/r06_timeline_${N}.txt 2>&1; sed -n 1,12p $o/r06_timeline_${N}.txt | cut -c1-200
fi
rate=$(grep -E 'proofs/s' $o/prof.log | tail -1 | sed 's/.* \([0-9.]*\) proofs\/s.*/\1/')
step "SQ instruction pass (rate of the un-profiled job: $rate proofs/s)"
cd /tmp
timeout -s KILL 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d "$R/$o/sq" -o x -- python "$R/tools/profile_batch.py" dense_4m ${N} > "$R/$o/sq.log" 2>&1; echo "sq rc=$?"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d "$R/$o/sqp" -o x -- python "$R/tools/r04/probe_compress.py" > "$R/$o/sqp.log" 2>&1; echo "sqp rc=$?"
cd "$R"
f=$(find $o/sq -name '*_results.db' | head -1); g=$(find $o/sqp -name '*_results.db' | head -1)
[ -n "$f" ] && python tools/pmc_sq_job.py "$f" $((2*N)) "$rate" $o/r06_pmc_sq_bench${N}.json "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- python tools/profile_batch.py dense_4m ${N} (cohort launches of the two ${N}-proof batches; final build of round 6)" "$g" 2097152 > $o/pmc_sq.txt 2>&1 && cp $o/r06_pmc_sq_bench${N}.json profiles/
head -6 $o/pmc_sq.txt | cut -c1-250
find $o -name '*.db' -size +2M -delete
step "single-proof latency A/B (after a warm-up process: the first GPU process of a box runs ~4 ms slower)"
timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_warmup.txt 2>&1
for rep in 1 2; do
  timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_default_$rep.txt 2>&1; echo "default:            $(grep -E 'proof [3-5]' $o/lat_default_$rep.txt | sed 's/.*library //' | tr '\n' ' ')"
  DP_LP_MAX=4096 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_lp4096_$rep.txt 2>&1; echo "DP_LP_MAX=4096:     $(grep -E 'proof [3-5]' $o/lat_lp4096_$rep.txt | sed 's/.*library //' | tr '\n' ' ')"
  DP_MAILBOX_VRAM=0 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_hostmail_$rep.txt 2>&1; echo "DP_MAILBOX_VRAM=0:  $(grep -E 'proof [3-5]' $o/lat_hostmail_$rep.txt | sed 's/.*library //' | tr '\n' ' ')"
  DP_NUMA_PIN=0 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_nopin_$rep.txt 2>&1; echo "DP_NUMA_PIN=0:      $(grep -E 'proof [3-5]' $o/lat_nopin_$rep.txt | sed 's/.*library //' | tr '\n' ' ')"
done
timeout -s KILL 200 python tools/archive/latency_probe.py cnn_264k > $o/lat_cnn.txt 2>&1; echo "cnn_264k default:   $(grep -E 'proof [3-5]' $o/lat_cnn.txt | sed 's/.*library //' | tr '\n' ' ')"
DP_TIMING=2 timeout -s KILL 200 python tools/archive/latency_probe.py > $o/lat_t2.txt 2>&1; grep -E "sc-debug|sumcheck rounds" $o/lat_t2.txt | tail -2 | cut -c1-260
( for f in default_1 lp4096_1 hostmail_1 nopin_1 default_2 lp4096_2 hostmail_2 nopin_2; do echo "$f: $(grep -E 'proof [3-5]' $o/lat_$f.txt | sed 's/.*library //' | tr '\n' ' ')"; done; echo "cnn_264k: $(grep -E 'proof [3-5]' $o/lat_cnn.txt | sed 's/.*library //' | tr '\n' ' ')"; grep -E "sc-debug|sumcheck rounds" $o/lat_t2.txt | tail -2 ) > $o/r06_latency_ab.txt
step "bench"
timeout -s KILL 1500 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -3 $o/bench.err | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_final/bench.json').read().strip().split('\n')[-1])
print('dense4m', d['value'], 'golden', d['config']['golden_sha256_ok'], 'lat', d['config']['single_proof_latency_ms'], 'steps', d.get('step_ms_min_median_max'))
print('cnn', d['cnn_264k']['value'], d['cnn_264k'].get('single_proof_latency_ms'), d['cnn_264k'].get('golden_sha256_ok'))
print('sc24', d['sumcheck24']['wall_ms'], d['sumcheck24']['golden_sha256_ok'], d['sumcheck24'].get('roofline') and {k: d['sumcheck24']['roofline'].get(k) for k in ('frac','traffic','traffic_source','avg_launch_us')})
print('sc26', {k: (d.get('sumcheck26') or {}).get(k) for k in ('wall_ms','verified','end_to_end_hbm_frac','error')})
print('batch64', d.get('batch64'))
print('tail', {k: (d.get('tail_roofline') or {}).get(k) for k in ('frac_member','frac_merged_launch','median_member_us','merged_launch_us','source')})
t=d.get('transformer_layer') or {}
print('tl', {k:t.get(k) for k in ('value','proofs_in_flight','single_proof_latency_ms','golden_sha256_ok','error')})
print('seam', {k:(v.get('seam_level_proofs_per_s') if isinstance(v,dict) else v) for k,v in d['seam_level'].items() if k!='note'})
r=d['roofline']
print('roofline', {k:r.get(k) for k in ('achieved','peak','frac','job_frac','job_frac_of_sustained_peak','traffic','traffic_source','avg_launch_us','peak_valu_bound','frac_of_valu_bound','valu_issue_util','valu_issue_util_at_sampled_clock','valu_source')})
print('cpu', d['cpu_baseline'] and {k:d['cpu_baseline'].get(k) for k in ('value','cores','kind')})
PY
