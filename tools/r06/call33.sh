#!/bin/bash
# r06 call 33: hash waves that own a tail-sized share of the register file (DP_HASH_VGPRS builds: h168 = 3 hash waves per SIMD + tails compiled for <= 168 VGPRs,
# h256 = 2 hash waves per SIMD). (a) does the compress probe still reach the VALU rate with 3 / 2 waves per SIMD? (b) Dense-4M at 704 in flight, in phase and staggered
o=gpurun_out/r06_call33; mkdir -p $o; export TMPDIR=/tmp
probe() { DP_LIB_VARIANT=$1 timeout -s KILL 200 python - > $o/probe_$1.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import deep_prove_amd as dpa
dev = dpa.Device(0)
for n in (21, 17):
    r = [dev.probe_compress_rate(1 << n, 8) for _ in range(6)]
    print(f"variant={os.environ.get('DP_LIB_VARIANT') or 'release'} 2^{n}-node layer: best {max(r) / 1e9:.4f} median {sorted(r)[3] / 1e9:.4f} Gcompress/s")
PY
cat $o/probe_$1.txt | tail -2; }
run() { tag=$1; wl=$2; n=$3; nb=$4; shift 4; env "$@" timeout -s KILL 300 python tools/r04/ab_batch.py $wl $n $nb > $o/ab_$tag.txt 2>&1; echo "$tag: $(tail -1 $o/ab_$tag.txt | cut -c1-140)"; }
probe ""
probe h168
probe h256
ST="DP_COHORT_GROUPS=22 DP_COHORT_STAGGER_MS=25"
run rel_a dense_4m 704 8 X=1
run h168_a dense_4m 704 8 DP_LIB_VARIANT=h168
run h256_a dense_4m 704 8 DP_LIB_VARIANT=h256
run rel_st dense_4m 704 8 $ST
run h168_st dense_4m 704 8 DP_LIB_VARIANT=h168 $ST
run h256_st dense_4m 704 8 DP_LIB_VARIANT=h256 $ST
run rel_b dense_4m 704 8 X=1
run h168_b dense_4m 704 8 DP_LIB_VARIANT=h168
run h168_cnn cnn_264k 674 4 DP_LIB_VARIANT=h168
run rel_cnn cnn_264k 674 4 X=1
