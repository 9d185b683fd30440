export TMPDIR=/tmp
for v in "DP_MAILBOX_VRAM=0" "X=1"; do echo "== $v"; env $v DP_POLL_TIMEOUT_S=2 DP_TIMING=2 timeout -s KILL 100 python tools/sumcheck24_only.py 1 22 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -6; done
