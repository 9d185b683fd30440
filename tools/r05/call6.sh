#!/bin/bash
# r05 call 6: the tails ALONE on the chip (one proof at a time, throughput-mode kernels): rocprofv3 kernel stats, round-5 library and the round-4 k_logup_tail (v1)
o=gpurun_out/r05_call6; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for v in release v1; do
  if [ $v = release ]; then unset DP_LIB_VARIANT; else export DP_LIB_VARIANT=$v; fi
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d "$R/$o/solo_$v" -o x -- python "$R/tools/r05/solo_tails.py" dense_4m 6 > "$R/$o/solo_$v.log" 2>&1; echo "$v rc=$?"; tail -1 "$R/$o/solo_$v.log"
  f=$(find "$R/$o/solo_$v" -name '*_results.db' | head -1)
  if [ -n "$f" ]; then python "$R/tools/rocpd_summary.py" "$f" "$R/$o/solo_${v}_kernel_stats.csv" > /dev/null 2>&1; grep -E "tail|persist" "$R/$o/solo_${v}_kernel_stats.csv" | cut -c1-120; fi
done
cd "$R"; find $o -name '*_results.db' -size +20M -delete; ls -la $o
