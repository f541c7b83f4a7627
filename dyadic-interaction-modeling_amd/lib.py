"""ctypes binding of libdimx_hip.so (the C-ABI of include/dimx.h).

torch is used only for device memory and streams: tensors are handed to the library as
raw ``data_ptr()`` values plus the current HIP stream.  There is NO fallback: if the
shared library is missing, cannot be built, or a call fails, an exception is raised.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint64,
                    c_void_p)

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DIMX_LIB", os.path.join(HERE, "libdimx_hip.so"))   # DIMX_LIB: A/B a saved build

MODE_PARITY_F32 = 0
MODE_PERF_BF16 = 1
F32, BF16 = 0, 1
ACT_NONE, ACT_LEAKY, ACT_GELU_TANH, ACT_GELU_ERF = 0, 1, 2, 3


class DimxError(RuntimeError):
    pass


class Dims(ctypes.Structure):
    _fields_ = [(n, c_int) for n in (
        "vq_in_dim", "vq_hidden", "vq_layers", "vq_heads", "vq_inter", "vq_n_embed", "vq_zdim",
        "dim_in", "dim", "dim_a", "enc_depth", "dec_depth", "heads", "dim_head", "num_tokens",
        "max_seq_len", "ff_mult", "variant", "spk_in_dim", "spk_hidden", "spk_heads", "spk_inter",
        "spk_face_quan_num")]


class WeightDesc(ctypes.Structure):
    _fields_ = [("name", c_char_p), ("data", POINTER(c_float)), ("ndim", c_int), ("shape", c_int64 * 4)]


# every symbol include/dimx.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "dimx_version": (c_int, []),
    "dimx_last_error": (c_char_p, []),
    "dimx_default_dims": (None, [POINTER(Dims)]),
    "dimx_legacy_dims": (None, [POINTER(Dims)]),
    "dimx_create": (c_int, [POINTER(c_void_p), c_int, POINTER(Dims), c_int]),
    "dimx_destroy": (c_int, [c_void_p]),
    "dimx_numeric_mode": (c_int, [c_void_p]),
    "dimx_load_weights": (c_int, [c_void_p, POINTER(WeightDesc), c_int]),
    "dimx_begin_checkpoint": (c_int, [c_void_p]),
    "dimx_missing_weights": (c_int, [c_void_p]),
    "dimx_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "dimx_workspace_bytes_samples": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "dimx_vq_encode": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int32,
                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dimx_vq_argmin": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "dimx_vq_decode": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                               c_void_p]),
    "dimx_vq_decode_latent": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                      c_void_p]),
    "dimx_encode_speaker": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t,
                                    c_void_p]),
    "dimx_set_shard": (c_int, [c_void_p, c_int, c_int]),
    "dimx_encode_ctx": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "dimx_legacy_speaker_features": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                             c_void_p, c_size_t, c_void_p]),
    "dimx_slm_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dimx_set_context": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t,
                                 c_void_p]),
    "dimx_decode_tf": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "dimx_generate": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p, c_uint64,
                              c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dimx_train_num_params": (c_int, [c_void_p]),
    "dimx_train_total": (c_int64, [c_void_p]),
    "dimx_train_param_info": (c_int, [c_void_p, c_int, POINTER(c_char_p), POINTER(c_int64), POINTER(c_int64)]),
    "dimx_train_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "dimx_train_forward_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                            c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dimx_train_graph_stats": (c_int, [c_void_p, c_void_p]),
    "dimx_train_legacy_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "dimx_train_legacy_forward_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dimx_train_slm_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "dimx_train_slm_forward_backward": (c_int, [c_void_p] + [c_void_p] * 14 + [c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "dimx_train_adamw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float, c_float,
                                 c_int, c_float, c_void_p, c_void_p]),
    "dimx_op_train_attention": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "dimx_chain_faults": (c_int, [c_void_p]),
    "dimx_debug_chain_fault": (c_int, [c_void_p, c_int]),
    "dimx_op_gemm_slabs": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "dimx_op_split_x3": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_void_p]),
    "dimx_op_gemm_x3": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                c_void_p]),
    "dimx_op_gemm": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                             c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "dimx_op_gemm_headmajor": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_void_p]),
    "dimx_op_layernorm": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dimx_op_instnorm": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dimx_op_attention": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_void_p,
                                  c_void_p]),
    "dimx_op_attention_rowv": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "dimx_op_decode_attn": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_float, c_void_p, c_int, c_int, c_void_p]),
    "dimx_op_fused_probe": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "dimx_op_decode_attn_self": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_void_p, c_float, c_int, c_void_p]),
    "dimx_op_mlp_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dimx_mlp_fused_packed_bytes": (c_size_t, [c_int, c_int]),
    "dimx_mlp_fused_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "dimx_op_mlp_fused_packed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dimx_op_add_slabs_layernorm": (c_int, [c_int, c_void_p, c_void_p, c_int, ctypes.c_long, c_void_p, c_void_p, c_int,
                                            c_int, c_void_p]),
    "dimx_op_chain": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                              c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dimx_op_chain_ln": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                 c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dimx_op_layer_chain": (c_int, [c_void_p, c_int, ctypes.c_long, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "dimx_op_gemm_ln": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                c_void_p, c_void_p]),
    "dimx_op_sample": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_uint64, c_uint64, c_void_p,
                               c_void_p]),
}

_lib = None


def load(build_if_missing=True):
    """dlopen the library (building it in-tree with hipcc first if it is absent)."""
    global _lib
    if _lib is not None:
        return _lib
    in_tree = "DIMX_LIB" not in os.environ
    if not os.path.exists(LIB_PATH):
        if not build_if_missing or not in_tree:
            raise DimxError("libdimx_hip.so not found at %s (run python __graft_entry__.py build)" % LIB_PATH)
        from . import build as _build
        _build.build()
    elif in_tree:
        # The in-tree library must have been built from the sources next to it (content hash, build.stale()): the ctypes
        # signatures below describe THESE sources.  Stale + hipcc present -> rebuild (one process at a time, build.py
        # takes a file lock); stale without a compiler -> refuse: calling changed entry points through old signatures
        # corrupts arguments silently.  DIMX_ALLOW_STALE_LIB=1 overrides (A/B runs against a saved build use DIMX_LIB).
        from . import build as _build
        if _build.stale():
            if build_if_missing and _build.have_hipcc():
                _build.build()
            elif not os.environ.get("DIMX_ALLOW_STALE_LIB"):
                raise DimxError("libdimx_hip.so at %s was not built from the sources next to it and %s; run "
                                "python __graft_entry__.py build where hipcc exists"
                                % (LIB_PATH, "hipcc is not available" if build_if_missing else "rebuilding was not requested"))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        if not in_tree and not hasattr(lib, name):
            # A/B run against an older saved build (DIMX_LIB): a newer entry point fails at its first use, loudly
            setattr(lib, name, _missing_symbol(name))
            continue
        fn = getattr(lib, name)       # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _missing_symbol(name):
    def _raise(*a, **k):
        raise DimxError("%s is not exported by %s (DIMX_LIB points to an older build)" % (name, LIB_PATH))
    return _raise


def check(status, what=""):
    if status != 0:
        msg = load().dimx_last_error()
        raise DimxError("%s failed (%d): %s" % (what or "dimx call", status,
                                                 msg.decode() if msg else "?"))


def ptr(t):
    """raw device/host pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "dimx needs contiguous tensors"
    return c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def default_dims():
    d = Dims()
    load().dimx_default_dims(ctypes.byref(d))
    return d


def slm_dims():
    d = default_dims()
    d.variant = 2
    return d


def legacy_dims():
    d = Dims()
    load().dimx_legacy_dims(ctypes.byref(d))
    return d
