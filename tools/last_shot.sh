#!/bin/bash
# kernel trace of the cohort scheme + the two PMC passes of the 2^24 sumcheck, sized for a ~70 s slot
export TMPDIR=/tmp; o=gpurun_out/shot; mkdir -p $o
timeout 38 rocprofv3 --kernel-trace --stats -d $o/kt -o x -- python tools/profile_batch.py dense_4m 192 > $o/kt.log 2>&1
db=$(find $o/kt -name '*_results.db' | head -1)
[ -n "$db" ] && python tools/rocpd_summary.py "$db" $o/cohort_kernel_stats.csv > $o/kt_summary.txt 2>&1
[ -n "$db" ] && [ $(stat -c %s "$db") -gt 20000000 ] && rm -f "$db"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 16 rocprofv3 --kernel-trace --pmc $c -d $o/pmc_$c -o x -- python tools/sumcheck24_only.py 2 > $o/pmc_$c.log 2>&1
done
tail -2 $o/kt.log; head -8 $o/kt_summary.txt; ls -la $o $o/pmc_* | head -30
