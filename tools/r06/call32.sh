#!/bin/bash
# r06 call 32: the FETCH_SIZE / WRITE_SIZE passes of the 2^26 sumcheck (the block final.sh now has), so that sumcheck26.roofline.traffic comes from a pass of ITS size
o=gpurun_out/r06_final; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c -d "$R/$o/sc26_$c" -o x -- python "$R/tools/sumcheck24_only.py" 3 26 > "$R/$o/sc26_$c.log" 2>&1; echo "sc26 $c rc=$?"
done
cd "$R"
f=$(find "$o/sc26_FETCH_SIZE" -name '*_results.db' | head -1); w=$(find "$o/sc26_WRITE_SIZE" -name '*_results.db' | head -1)
python tools/pmc_summary.py --population sumcheck26 --units 3 "$f" "$w" "$o/r06_pmc_sumcheck26.json" "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/sumcheck24_only.py 3 26 (3 repetitions of the 2^26 sumcheck; final build of round 6, tools/r06/call32.sh = the block of final.sh)" k_sc > "$o/pmc_sc26.txt" 2>&1
tail -6 "$o/pmc_sc26.txt" | cut -c1-300
find $o -name '*.db' -size +2M -delete
