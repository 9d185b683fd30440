#!/bin/bash
# Round-2 evidence run (profiles/r02_*): the default bench line under rocprofv3 --kernel-trace --stats, the 2^24 sumcheck the
# same way, then the counter passes (separate runs, --kernel-trace only next to --pmc): FETCH_SIZE / WRITE_SIZE of the sumcheck
# and of one latency-mode proof, and one SQ pass each (VALU instructions / busy cycles) for the "VALU-integer bound" statements.
o=${1:-gpurun_out/r02_profile}; mkdir -p "$o"; export TMPDIR=/tmp
DP_BENCH_NO_TORCH=1 timeout 420 rocprofv3 --kernel-trace --stats -d "$o/bench_kt" -o x -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$o/bench_under_rocprof.json" 2> "$o/bench_under_rocprof.err"
db=$(find "$o/bench_kt" -name '*_results.db' | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" "$o/bench_kernel_stats.csv" > "$o/bench_kernel_stats.txt" 2>&1 && python tools/trace_analyze.py "$db" > "$o/bench_trace_analysis.txt" 2>&1
[ -n "$db" ] && rm -f "$db"
timeout 100 rocprofv3 --kernel-trace --stats -d "$o/sc24_kt" -o x -- python tools/sumcheck24_only.py 5 > "$o/sc24_kt.log" 2>&1
db=$(find "$o/sc24_kt" -name '*_results.db' | head -1); [ -n "$db" ] && python tools/rocpd_summary.py "$db" "$o/sumcheck24_kernel_stats.csv" > "$o/sumcheck24_kernel_stats.txt" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 60 rocprofv3 --kernel-trace --pmc $c -d "$o/sc24_$c" -o x -- python tools/sumcheck24_only.py 2 > "$o/sc24_$c.log" 2>&1
  timeout 60 rocprofv3 --kernel-trace --pmc $c -d "$o/proof_$c" -o x -- python tools/one_proof_cwd.py > "$o/proof_$c.log" 2>&1
done
f=$(find "$o/sc24_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/sc24_WRITE_SIZE" -name '*_results.db' | head -1)
[ -n "$f" ] && [ -n "$w" ] && python tools/pmc_summary.py "$f" "$w" "$o/pmc_sumcheck24.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/sumcheck24_only.py 2" k_sc > "$o/pmc_sumcheck24.txt" 2>&1
f=$(find "$o/proof_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/proof_WRITE_SIZE" -name '*_results.db' | head -1)
[ -n "$f" ] && [ -n "$w" ] && python tools/pmc_summary.py "$f" "$w" "$o/pmc_dense4m_proof.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/one_proof_cwd.py" > "$o/pmc_dense4m_proof.txt" 2>&1
SQC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
timeout 60 rocprofv3 --kernel-trace --pmc $SQC -d "$o/sc24_SQ" -o x -- python tools/sumcheck24_only.py 2 > "$o/sc24_SQ.log" 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc $SQC -d "$o/proof_SQ" -o x -- python tools/one_proof_cwd.py > "$o/proof_SQ.log" 2>&1
d=$(find "$o/sc24_SQ" -name '*_results.db' | head -1); [ -n "$d" ] && python tools/pmc_generic.py "$d" "$o/pmc_sq_sumcheck24.json" "rocprofv3 --kernel-trace --pmc $SQC -- python tools/sumcheck24_only.py 2" k_sc > "$o/pmc_sq_sumcheck24.txt" 2>&1
d=$(find "$o/proof_SQ" -name '*_results.db' | head -1); [ -n "$d" ] && python tools/pmc_generic.py "$d" "$o/pmc_sq_dense4m_proof.json" "rocprofv3 --kernel-trace --pmc $SQC -- python tools/one_proof_cwd.py" k_merkle k_logup k_sc > "$o/pmc_sq_dense4m_proof.txt" 2>&1
find "$o" -name '*_results.db' -size +8M -delete
tail -2 "$o/bench_under_rocprof.err" | cut -c1-200; head -c 600 "$o/bench_under_rocprof.json"; echo; head -12 "$o/bench_kernel_stats.txt"; head -8 "$o/sumcheck24_kernel_stats.txt"; cat "$o/pmc_sumcheck24.txt" "$o/pmc_sq_sumcheck24.txt" "$o/pmc_sq_dense4m_proof.txt" 2>/dev/null | cut -c1-330 | head -30; tail -3 "$o/sc24_SQ.log" | cut -c1-200
