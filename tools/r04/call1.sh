#!/bin/bash
# r04 call 1: single-wave issue price list, the hand-written multiplication against the compiler's, and an SQ instruction-count pass over 3 latency-mode Dense-4M proofs (round-3 library)
o=gpurun_out/r04_call1; mkdir -p $o; export TMPDIR=/tmp
timeout -s KILL 120 tools/_build/issue_rate > $o/issue_rate.txt 2>&1; echo "issue_rate rc=$?"
timeout -s KILL 180 tools/_build/mulbench2_cc > $o/mulbench2_cc.txt 2>&1; echo "mulbench2_cc rc=$?"
timeout -s KILL 180 tools/_build/mulbench2_asm > $o/mulbench2_asm.txt 2>&1; echo "mulbench2_asm rc=$?"
cat $o/mulbench2_cc.txt $o/mulbench2_asm.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $o/sq -o x -- python tools/proof_only.py dense_4m 3 > $o/sq.log 2>&1; echo "sq rc=$?"
f=$(find $o/sq -name '*_results.db' | head -1)
[ -n "$f" ] && python tools/pmc_generic.py "$f" $o/pmc_sq_dense4m_r03lib.json "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- python tools/proof_only.py dense_4m 3 (round-3 library)" > $o/pmc_sq.txt 2>&1
find $o -name '*_results.db' -size +8M -delete
cat $o/issue_rate.txt
