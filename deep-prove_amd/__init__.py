"""deep-prove hot path, MI355X-native: sumcheck / logup-GKR / Basefold prover on hand-written gfx950 HIP kernels behind
the C ABI of include/deep_prove_hip.h. See DESIGN.md and INTEGRATION.md."""
from . import _lib, models, sharded, wire
from ._lib import DeepProveError, LIB_PATH
from .api import (P, AsyncEngine, Basefold, Commitment, Context, Device, Mle, Prover, Ticket, Transcript, VirtualPolynomial, build_eq_x_r,
                  infer_host, logup_batch_prove, prove_parallel, verify, verify_batch, verify_logup, verify_sumcheck)

__all__ = ["P", "AsyncEngine", "Ticket", "Basefold", "Commitment", "Context", "Device", "Mle", "Prover", "Transcript", "VirtualPolynomial",
           "build_eq_x_r", "infer_host", "logup_batch_prove", "prove_parallel", "verify", "verify_batch", "verify_logup", "verify_sumcheck", "models", "sharded", "wire", "DeepProveError", "LIB_PATH"]
