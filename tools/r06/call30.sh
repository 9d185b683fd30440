#!/bin/bash
# r06 call 30: the download copy with AVX-512 and the query section in a recycled, uninitialised block (host work at the end of a proof), then the final pipeline
o=gpurun_out/r06_call30; mkdir -p $o; export TMPDIR=/tmp
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-200)"; }
run new1 dense_4m 704 8 X=1
run noavx dense_4m 704 8 DP_NO_AVX512=1
run new2 dense_4m 704 8 X=1
DP_TIMING=3 timeout -s KILL 300 python tools/archive/conc_hoststats.py 704 2>&1 | grep -E "proofs/s|host phase before the fire of.*(k_query_gather|k_download|k_copy_words)" | sort | uniq -c | sort -rn | head -0
bash tools/r06/final.sh
