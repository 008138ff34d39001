"""ctypes binding of libsr_hip.so (C ABI in include/sr_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails,
this module raises.  Build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C scaling_retriever_amd/csrc`.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SR_HIP_LIB") or os.path.join(_HERE, "libsr_hip.so")     # SR_HIP_LIB: diagnostic builds (tools/micro)

SR_OK, SR_ERR_INVALID, SR_ERR_HIP, SR_ERR_NOMEM, SR_ERR_UNSUPPORTED = 0, 1, 2, 3, 4
SR_DTYPE_F32, SR_DTYPE_BF16 = 0, 1

c_void_p, c_int, c_int32, c_int64, c_float, c_char_p = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int32,
                                                         ctypes.c_int64, ctypes.c_float, ctypes.c_char_p)


class SrModelConfig(ctypes.Structure):
    _fields_ = [
        ("vocab_size", c_int32), ("hidden_size", c_int32), ("intermediate_size", c_int32), ("num_layers", c_int32),
        ("num_heads", c_int32), ("num_kv_heads", c_int32), ("head_dim", c_int32),
        ("rms_norm_eps", c_float), ("rope_theta", c_float), ("rope_llama3", c_int32),
        ("rope_factor", c_float), ("rope_low_freq_factor", c_float), ("rope_high_freq_factor", c_float),
        ("rope_original_max_pos", c_int32), ("tie_word_embeddings", c_int32), ("has_lm_head", c_int32),
        ("max_batch_tokens", c_int32), ("max_batch_seqs", c_int32), ("fp32_planes", c_int32),
    ]


# name -> (restype, argtypes); every function declared in include/sr_hip.h
SIGNATURES = {
    "sr_last_error": (c_char_p, []),
    "sr_version": (c_int, []),
    "sr_max_topk": (c_int, []),
    "sr_dense_index_create": (c_int, [ctypes.POINTER(c_void_p), c_int]),
    "sr_dense_index_add": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64]),
    "sr_dense_index_ntotal": (c_int64, [c_void_p]),
    "sr_dense_search": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "sr_dense_index_set_workspace_limit": (c_int, [c_void_p, c_int64]),
    "sr_dense_index_set_batch_invariant": (c_int, [c_void_p, c_int]),
    "sr_dense_index_set_precision": (c_int, [c_void_p, c_int]),
    "sr_dense_search_begin": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "sr_dense_search_finish": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sr_dense_index_filter_stats": (c_int, [c_void_p, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "sr_dense_index_filter_query_stats": (c_int, [c_void_p, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "sr_dense_index_destroy": (c_int, [c_void_p]),
    "sr_dense_index_profile": (c_int, [c_void_p, c_int]),
    "sr_dense_index_profile_read": (c_int, [c_void_p, ctypes.POINTER(c_int64), ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "sr_sparse_index_work_counters": (c_int, [c_void_p, c_int, ctypes.POINTER(ctypes.c_uint64)]),
    "sr_sparse_index_profile": (c_int, [c_void_p, c_int]),
    "sr_sparse_index_profile_read": (c_int, [c_void_p, ctypes.POINTER(c_int64), ctypes.POINTER(ctypes.c_double),
                                             ctypes.POINTER(ctypes.c_double)]),
    "sr_sparse_index_create": (c_int, [ctypes.POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                       c_void_p]),
    "sr_sparse_search": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int64, c_int64,
                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "sr_sparse_index_set_workspace_limit": (c_int, [c_void_p, c_int64]),
    "sr_sparse_index_block_stats": (c_int, [c_void_p, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64),
                                            ctypes.POINTER(c_int64)]),
    "sr_sparse_index_destroy": (c_int, [c_void_p]),
    "sr_sparse_index_cert_stats": (c_int, [c_void_p, ctypes.POINTER(c_int64)]),
    "sr_sparse_index_cert_debug": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64,
                                           ctypes.POINTER(c_float), ctypes.POINTER(ctypes.c_int32)]),
    "sr_sparse_csr_expand_terms": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "sr_sparse_csr_build": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "sr_topk_merge": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "sr_model_create": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(SrModelConfig)]),
    "sr_model_set_weight": (c_int, [c_void_p, c_char_p, c_void_p, c_int, c_int64, c_int64, c_void_p]),
    "sr_model_finalize": (c_int, [c_void_p]),
    "sr_encode_dense": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "sr_encode_sparse": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "sr_encode_dense_fp32": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "sr_encode_sparse_fp32": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "sr_encode_both": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "sr_encode_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "sr_model_last_hidden": (c_int, [c_void_p, c_void_p, c_int64, ctypes.POINTER(c_int64), c_void_p]),
    "sr_model_destroy": (c_int, [c_void_p]),
    "sr_lora_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_float, c_void_p]),
    "sr_run_writer_mapped_rounds": (c_int64, []),
    "sr_write_run_json": (c_int, [c_char_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                  c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, ctypes.POINTER(c_int64)]),
    "sr_write_run_json_part": (c_int, [c_char_p, c_int32, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                       c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, ctypes.POINTER(c_int64)]),
    "sr_gemm_bf16": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "sr_gemm_f16_scaled": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "sr_gemm_qkv_rope": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int32, c_int32, c_void_p]),
    "sr_attention_varlen": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                    c_int32, c_int32, c_int32, c_void_p]),
    "sr_sparse_compact": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                  ctypes.POINTER(c_int64), c_void_p]),
}

_lib = None
_lock = threading.Lock()


class SrHipError(RuntimeError):
    pass


def load():
    """Load libsr_hip.so and bind every entry point.  Raises if the library is missing."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SrHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU fallback; "
                "run `make -C scaling_retriever_amd/csrc` (needs hipcc, targets gfx950).")
        # torch first: it ships its own libamdhip64; were the system runtime pulled in by our DT_NEEDED before torch's, the
        # process would hold two HIP runtimes and the second would find "no ROCm-capable device"
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc, what=""):
    """Map a C-ABI status to the Python exception the reference would raise for the same condition."""
    if rc == SR_OK:
        return
    msg = load().sr_last_error().decode("utf-8", "replace")
    if rc == SR_ERR_INVALID:
        raise ValueError(f"{what}: {msg}" if what else msg)
    if rc == SR_ERR_NOMEM:
        raise MemoryError(f"{what}: {msg}" if what else msg)
    raise SrHipError(f"{what}: [{rc}] {msg}" if what else f"[{rc}] {msg}")


def stream_ptr():
    """hipStream_t of torch's current stream (so C-ABI work is ordered with torch ops)."""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise SrHipError("no ROCm device visible: scaling_retriever_amd runs its hot path on MI355X only "
                         "(there is no CPU fallback)")
