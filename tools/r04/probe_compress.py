"""the compress-rate probe alone (k_merkle_layer on a 2^21-node layer, 8 launches): wrapped by rocprofv3 --pmc for the VALU instructions per compress"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))  # (the repository root, wherever the command is started from)
import deep_prove_amd as dpa
dev = dpa.Device(0)
print("compress/s:", max(dev.probe_compress_rate(1 << 21, 8) for _ in range(2)), flush=True)
