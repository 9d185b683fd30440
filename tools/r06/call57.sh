#!/bin/bash
# r06 call 57: plain rocprofv3 --kernel-trace --stats of the bench command itself (python bench.py --steps 2 --warmup 1), the trace kept in /tmp, only the summary comes back
o=gpurun_out/r06_call57; mkdir -p $o; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout -s KILL 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python "$R/bench.py" --steps 2 --warmup 1 > "$R/$o/bench_under_rocprof.txt" 2> "$R/$o/bench_under_rocprof.err"; echo "rocprof rc=$?"
cd "$R"
db=$(find /tmp/prof_bench -name "*.db" | head -1); ls -la $db
python tools/rocpd_summary.py "$db" $o/r06_bench_command_kernel_stats.csv > $o/summary.err 2>&1; head -14 $o/r06_bench_command_kernel_stats.csv | cut -c1-140
tail -1 $o/bench_under_rocprof.txt | cut -c1-400
