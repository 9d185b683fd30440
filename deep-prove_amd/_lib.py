"""ctypes binding of libdeepprove_hip.so (include/deep_prove_hip.h). The library is mandatory: there is no Python or
CPU fallback for any device op — a missing .so or a missing HIP device raises."""
import ctypes as C
import os

# Every proof in flight (and every shard of a sharded sumcheck) owns a HIP stream whose persistent kernels wait for the
# host: two such streams must never share a hardware queue (the second one's kernels would queue behind a kernel that waits
# for a challenge the host only sends after the second stream made progress). ROCm multiplexes streams onto
# GPU_MAX_HW_QUEUES queues (default 4), read when the HIP runtime initialises — so it has to be set before the first HIP
# call of the process. 24 is the most an MI355X serves without time-slicing the queues (tools/queue_oversub.hip).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

_HERE = os.path.dirname(os.path.abspath(__file__))
# DP_LIB_VARIANT=<tag> loads libdeepprove_hip_<tag>.so: a diagnostic build (DP_HIPCC_EXTRA, __graft_entry__.build_hip(variant=...)) next to the release library
LIB_PATH = os.path.join(_HERE, "libdeepprove_hip" + ("_" + os.environ["DP_LIB_VARIANT"] if os.environ.get("DP_LIB_VARIANT") else "") + ".so")

u64p = C.POINTER(C.c_uint64)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/deep_prove_hip.h
SIGNATURES = {
    "dp_last_error": (C.c_char_p, []),
    "dp_free": (None, [vp]),
    "dp_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(vp)]),
    "dp_ctx_destroy": (C.c_int32, [vp]),
    "dp_ctx_name": (C.c_char_p, [vp]),
    "dp_ctx_set_throughput_mode": (C.c_int32, [vp, C.c_int32]),
    "dp_async_create": (C.c_int32, [vp, C.c_int32, C.c_size_t, C.POINTER(vp)]),
    "dp_ctx_route_to_engine": (C.c_int32, [vp, vp]),
    "dp_async_destroy": (C.c_int32, [vp]),
    "dp_async_stats": (C.c_int32, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "dp_pcs_commit_submit": (C.c_int32, [vp, vp, C.POINTER(vp)]),
    "dp_pcs_commit_host_submit": (C.c_int32, [vp, u64p, C.c_size_t, C.c_int32, C.POINTER(vp)]),
    "dp_mle_fix_high_submit": (C.c_int32, [vp, vp, C.c_size_t, C.c_size_t, u64p, C.POINTER(vp)]),
    "dp_mle_eval_submit": (C.c_int32, [vp, vp, u64p, C.c_uint32, C.POINTER(vp)]),
    "dp_ticket_buf": (C.c_int32, [vp, C.POINTER(vp)]),
    "dp_sumcheck_prove_submit": (C.c_int32, [vp, C.c_uint32, C.POINTER(vp), C.c_int32, i32p, i32p, u64p, C.c_int32, vp, C.POINTER(vp)]),
    "dp_logup_prove_submit": (C.c_int32, [vp, C.POINTER(vp), C.c_int32, C.c_int32, vp, u64p, u64p, vp, C.POINTER(vp)]),
    "dp_pcs_batch_open_submit": (C.c_int32, [vp, C.POINTER(vp), C.c_int32, u64p, u64p, vp, C.POINTER(vp)]),
    "dp_poll": (C.c_int32, [vp]),
    "dp_wait": (C.c_int32, [vp]),
    "dp_ticket_words": (C.c_int32, [vp, C.c_int32, C.POINTER(u64p), C.POINTER(C.c_size_t)]),
    "dp_ticket_values": (C.c_int32, [vp, u64p, C.c_size_t]),
    "dp_ticket_commit": (C.c_int32, [vp, C.POINTER(vp), u64p]),
    "dp_ticket_free": (C.c_int32, [vp]),
    "dp_profile_enable": (C.c_int32, [vp, C.c_int32]),
    "dp_profile_report": (C.c_int32, [vp, C.POINTER(C.c_void_p)]),
    "dp_probe_compress_rate": (C.c_int32, [vp, C.c_size_t, C.c_int32, C.POINTER(C.c_double)]),
    "dp_buf_from_i64": (C.c_int32, [vp, i64p, C.c_size_t, C.POINTER(vp)]),
    "dp_buf_upload": (C.c_int32, [vp, u64p, C.c_size_t, C.c_int32, C.POINTER(vp)]),
    "dp_buf_download": (C.c_int32, [vp, vp, u64p]),
    "dp_buf_len": (C.c_size_t, [vp]),
    "dp_buf_is_ext": (C.c_int32, [vp]),
    "dp_buf_free": (C.c_int32, [vp, vp]),
    "dp_transcript_new": (vp, [C.c_char_p]),
    "dp_transcript_free": (None, [vp]),
    "dp_transcript_append_elements": (C.c_int32, [vp, u64p, C.c_size_t]),
    "dp_transcript_append_message": (C.c_int32, [vp, C.c_char_p, C.c_size_t]),
    "dp_transcript_challenge": (C.c_int32, [vp, C.c_char_p, u64p]),
    "dp_eq_table": (C.c_int32, [vp, u64p, C.c_uint32, C.POINTER(vp)]),
    "dp_mle_eval": (C.c_int32, [vp, vp, u64p, C.c_uint32, u64p]),
    "dp_mle_fix_high": (C.c_int32, [vp, vp, C.c_size_t, C.c_size_t, u64p, C.POINTER(vp)]),
    "dp_sumcheck_prove": (C.c_int32, [vp, C.c_uint32, C.POINTER(vp), C.c_int32, i32p, i32p, u64p, C.c_int32, vp,
                                      C.POINTER(u64p), C.POINTER(C.c_size_t), u64p]),
    "dp_sc_session_new": (C.c_int32, [vp, C.c_uint32, C.POINTER(vp), C.c_int32, i32p, i32p, C.c_int32, C.POINTER(vp)]),
    "dp_sc_session_round": (C.c_int32, [vp, u64p, u64p, C.POINTER(C.c_size_t)]),
    "dp_sc_session_finish": (C.c_int32, [vp, u64p, u64p]),
    "dp_sc_session_free": (C.c_int32, [vp]),
    "dp_dist_unique_id": (C.c_int32, [C.POINTER(C.c_uint8)]),
    "dp_dist_init": (C.c_int32, [vp, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, C.POINTER(vp)]),
    "dp_dist_free": (C.c_int32, [vp]),
    "dp_sumcheck_prove_sharded": (C.c_int32, [vp, vp, C.c_uint32, C.POINTER(vp), C.c_int32, i32p, i32p, u64p, C.c_int32, vp, C.POINTER(u64p), C.POINTER(C.c_size_t), u64p]),
    "dp_sumcheck_prove_sharded_local": (C.c_int32, [C.POINTER(vp), C.c_int32, C.c_uint32, C.POINTER(vp), C.c_int32, i32p, i32p, u64p, C.c_int32, C.POINTER(vp),
                                                    C.POINTER(u64p), C.POINTER(C.c_size_t), u64p]),
    "dp_sumcheck_verify": (C.c_int32, [C.c_uint32, C.c_uint32, u64p, u64p, C.c_size_t, vp, u64p, u64p]),
    "dp_logup_verify": (C.c_int32, [u64p, C.c_size_t, C.c_int32, u64p, u64p, vp, u64p, u64p, C.POINTER(u64p), C.POINTER(C.c_size_t)]),
    "dp_logup_prove": (C.c_int32, [vp, C.POINTER(vp), C.c_int32, C.c_int32, vp, u64p, u64p, vp, C.POINTER(u64p),
                                   C.POINTER(C.c_size_t)]),
    "dp_pcs_setup": (C.c_int32, [vp, C.c_size_t]),
    "dp_pcs_commit": (C.c_int32, [vp, vp, C.POINTER(vp), u64p]),
    "dp_pcs_commit_free": (C.c_int32, [vp, vp]),
    "dp_pcs_commitment": (C.c_int32, [vp, u64p, u32p, i32p]),
    "dp_model_infer_host": (C.c_int32, [i64p, C.c_size_t, i64p, C.c_size_t, i64p, C.POINTER(C.c_size_t)]),
    "dp_host_poseidon2": (C.c_int32, [u64p, C.c_int32, C.POINTER(C.c_int32)]),
    "dp_pcs_open": (C.c_int32, [vp, vp, u64p, C.c_uint32, u64p, vp, C.POINTER(u64p), C.POINTER(C.c_size_t)]),
    "dp_pcs_verify": (C.c_int32, [C.c_size_t, u64p, C.c_uint32, C.c_int32, u64p, u64p, u64p, C.c_size_t, vp]),
    "dp_pcs_batch_open": (C.c_int32, [vp, C.POINTER(vp), C.c_int32, u64p, u64p, vp, C.POINTER(u64p),
                                      C.POINTER(C.c_size_t)]),
    "dp_pcs_batch_verify": (C.c_int32, [C.c_size_t, u64p, u32p, i32p, C.c_int32, u64p, u64p, u64p, C.c_size_t, vp]),
    "dp_pcs_batch_open_evals": (C.c_int32, [vp, C.POINTER(vp), C.c_int32, u64p, u32p, C.c_int32, u32p, u32p, u64p, C.c_int32, vp, C.POINTER(u64p), C.POINTER(C.c_size_t)]),
    "dp_pcs_batch_verify_evals": (C.c_int32, [C.c_size_t, u64p, u32p, i32p, C.c_int32, u64p, u32p, C.c_int32, u32p, u32p, u64p, C.c_int32, u64p, C.c_size_t, vp]),
    "dp_pcs_batch_commit": (C.c_int32, [vp, C.POINTER(vp), C.c_int32, C.POINTER(vp), u64p]),
    "dp_pcs_batch_commit_free": (C.c_int32, [vp, vp]),
    "dp_pcs_simple_batch_open": (C.c_int32, [vp, vp, u64p, C.c_uint32, vp, C.POINTER(u64p), C.POINTER(C.c_size_t)]),
    "dp_pcs_simple_batch_verify": (C.c_int32, [C.c_size_t, u64p, C.c_uint32, C.c_int32, u64p, u64p, C.c_int32, u64p, C.c_size_t, vp]),
    "dp_model_setup": (C.c_int32, [vp, i64p, C.c_size_t, C.POINTER(vp)]),
    "dp_model_free": (C.c_int32, [vp]),
    "dp_model_prove": (C.c_int32, [vp, i64p, C.c_size_t, C.POINTER(u64p), C.POINTER(C.c_size_t), i64p,
                                   C.POINTER(C.c_size_t), C.POINTER(C.c_double)]),
    "dp_model_prove_batch": (C.c_int32, [vp, i64p, C.c_size_t, C.c_size_t, C.c_int32, C.POINTER(u64p), C.POINTER(C.c_size_t), i64p,
                                         C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_double)]),
    "dp_model_verifier_blob": (C.c_int32, [vp, C.POINTER(u64p), C.POINTER(C.c_size_t)]),
    "dp_model_in_flight": (C.c_int32, [vp, C.POINTER(C.c_size_t)]),
    "dp_model_output_len": (C.c_int32, [vp, C.POINTER(C.c_size_t)]),
    "dp_host_cpu_budget": (C.c_double, []),
    "dp_verify_batch": (C.c_int32, [vp, u64p, C.c_size_t, C.POINTER(u64p), C.POINTER(C.c_size_t), i64p, C.c_size_t, i64p, C.c_size_t, C.c_size_t, C.c_int32, i32p, C.POINTER(C.c_double)]),
    "dp_verify": (C.c_int32, [u64p, C.c_size_t, u64p, C.c_size_t, i64p, C.c_size_t, i64p, C.c_size_t]),
}

_lib = None


def load():
    """Load the native library (raises if it has not been built: run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension is mandatory (build it with __graft_entry__.build())")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class DeepProveError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"[{code}] {msg}")
        self.code = code


def check(rc):
    if rc != 0:
        raise DeepProveError(rc, load().dp_last_error().decode())
