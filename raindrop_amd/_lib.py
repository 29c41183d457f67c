"""ctypes binding of libraindrop_hip.so (the C-ABI declared in include/raindrop_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol cannot be
resolved, importing/using the ops raises.  Build it with `python -m raindrop_amd.build`
(or `__graft_entry__.build()`).
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_size_t, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RD_LIB_PATH") or os.path.join(_PKG, "libraindrop_hip.so")   # RD_LIB_PATH: A/B builds (tools/ab_build.sh)


class RdShape(ctypes.Structure):
    """Mirror of `struct rd_shape` (include/raindrop_hip.h)."""
    _fields_ = [(n, c_int32) for n in ("B", "T", "F", "d_ob", "d_pe", "nhead", "nhid", "d_static",
                                       "n_classes", "max_len")]


class RaindropHipError(RuntimeError):
    pass


class RdEncoderPtrs(ctypes.Structure):
    """Mirror of rd_encoder_weights / rd_encoder_grads (12 device pointers)."""
    FIELDS = ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b", "lin2_w",
              "lin2_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b")
    _fields_ = [(n, c_void_p) for n in FIELDS]


_P = c_void_p
_ENC = POINTER(RdEncoderPtrs)
_SHP = POINTER(RdShape)

# name -> (restype, argtypes); every symbol the header declares appears here, and
# tests/test_cabi_symbols.py checks the two lists against each other.
SIGNATURES = {
    "rd_version": (c_int32, []),
    "rd_arch": (c_char_p, []),
    "rd_last_error": (c_char_p, []),
    "rd_set_precision": (c_int32, [c_int32]),
    "rd_get_precision": (c_int32, []),
    "rd_set_seed_cell": (c_int32, [_P]),
    "rd_set_side_stream": (c_int32, [_P]),
    "rd_side_join": (c_int32, [_P]),
    "rd_set_defer_trailing": (c_int32, [c_int32]),
    "rd_flush_trailing": (c_int32, [_P]),
    "rd_drop_trailing": (c_int32, []),
    "rd_seed_cell_advance": (c_int32, [_P, ctypes.c_uint64, _P]),
    "rd_token_plan_bytes": (c_size_t, [_SHP]),
    "rd_token_plan": (c_int32, [_SHP, _P, _P, _P, ctypes.c_uint64, _P]),
    "rd_set_token_plan": (c_int32, [_P]),
    "rd_graph_build": (c_int32, [c_int32, _P, _P, _P, _P, _P, _P]),
    "rd_pe_mask": (c_int32, [_SHP, _P, _P, _P, _P, _P, _P]),
    "rd_edge_softmax": (c_int32, [c_int32, _P, _P, _P, _P]),
    "rd_edge_softmax_list": (c_int32, [c_int32, c_int32, _P, ctypes.c_int64, c_int32, _P, _P, _P, _P]),
    "rd_edge_attention_fwd": (c_int32, [c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, ctypes.c_int64, c_float, ctypes.c_uint64,
                                        _P, _P, _P, _P]),
    "rd_edge_attention_bwd": (c_int32, [c_int32, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, ctypes.c_int64, c_float, ctypes.c_uint64,
                                        _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "rd_edge_softmax_list_dropout": (c_int32, [c_int32, c_int32, _P, ctypes.c_int64, c_int32, _P, c_float, ctypes.c_uint64, _P, _P, _P]),
    "rd_aggregate_fwd": (c_int32, [c_int32, c_int32, _P, _P, _P, _P, _P]),
    "rd_aggregate_bwd": (c_int32, [c_int32, c_int32, _P, _P, _P, _P]),
    "rd_edge_softmax_list_batched": (c_int32, [c_int32, c_int32, c_int32, _P, ctypes.c_int64, ctypes.c_int64, c_int32, _P,
                                               ctypes.c_int64, _P, _P, _P]),
    "rd_edge_gamma_dense": (c_int32, [c_int32, c_int32, _P, ctypes.c_int64, _P, _P, _P]),
    "rd_aggregate_batched_fwd": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P]),
    "rd_aggregate_batched_bwd": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P]),
    "rd_scale_dropout": (c_int32, [ctypes.c_int64, _P, c_float, c_float, ctypes.c_uint64, ctypes.c_uint32, _P, _P]),
    "rd_obs_embed_fwd": (c_int32, [_SHP, _P, _P, c_float, ctypes.c_uint64, _P, _P]),
    "rd_obs_embed_bwd_workspace_bytes": (c_size_t, [_SHP]),
    "rd_obs_embed_bwd": (c_int32, [_SHP, _P, _P, _P, c_float, _P, _P, c_size_t, _P]),
    "rd_rows_to_tokens_fwd": (c_int32, [_SHP, _P, _P, _P, c_int32, _P]),
    "rd_rows_to_tokens_bwd": (c_int32, [_SHP, _P, c_int32, _P, _P, _P]),
    "rd_msgpass_workspace_bytes": (c_size_t, [_SHP]),
    "rd_msgpass_saved_bytes": (c_size_t, [_SHP]),
    "rd_msgpass_fwd": (c_int32, [_SHP] + [_P] * 7 + [c_float, ctypes.c_uint64, _P, c_int32, _P, c_size_t, _P]),
    "rd_sensor_stage_fwd": (c_int32, [_SHP] + [_P] * 10 + [c_float, ctypes.c_uint64, _P, _P, _P, c_size_t, _P]),
    "rd_sensor_stage_fwd_prepared": (c_int32, [_SHP] + [_P] * 10 + [c_float, ctypes.c_uint64, _P, _P, _P, c_size_t, _P]),
    "rd_step_prepare": (c_int32, [_SHP, c_int32, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "rd_step_begin": (c_int32, [_SHP, _P, _P, _P, ctypes.c_uint64, c_int32, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "rd_step_prepare_covers": (c_int32, [_SHP, _P, _P]),
    "rd_msgpass_bwd": (c_int32, [_SHP] + [_P] * 5 + [c_float, _P, c_size_t, _P, _P, c_int32] + [_P] * 5
                       + [_P, c_size_t, _P]),
    "rd_encoder_layer_saved_bytes": (c_size_t, [_SHP]),
    "rd_encoder_layer_workspace_bytes": (c_size_t, [_SHP]),
    "rd_encoder_layer_prepare": (c_int32, [_SHP, _ENC, _P, c_size_t, _P]),
    "rd_encoder_layer_fwd": (c_int32, [_SHP, c_int32, _P, _P, _ENC, c_float, ctypes.c_uint64, _P, _P, c_size_t,
                                        _P, c_size_t, _P]),
    "rd_encoder_layer_bwd": (c_int32, [_SHP, c_int32, _P, _P, _ENC, c_float, ctypes.c_uint64, _P, c_size_t, _P, _P,
                                        _ENC, _P, c_size_t, _P]),
    "rd_attention_fwd": (c_int32, [_SHP, c_int32, _P, _P, c_float, ctypes.c_uint64, _P, _P, _P]),
    "rd_attention_bwd": (c_int32, [_SHP, c_int32, _P, _P, c_float, ctypes.c_uint64, _P, _P, _P, _P, _P, _P]),
    "rd_masked_mean_fwd": (c_int32, [_SHP, c_int32, _P, _P, _P, _P, c_int32, _P]),
    "rd_masked_mean_bwd": (c_int32, [_SHP, c_int32, _P, c_int32, _P, _P, _P, _P]),
    "rd_adam_step": (c_int32, [ctypes.c_int64, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float,
                                ctypes.c_int64, _P]),
    "rd_adam_step_dev": (c_int32, [ctypes.c_int64, _P, _P, _P, _P, c_float, c_float, c_float, _P, _P]),
    "rd_adam_state_advance": (c_int32, [_P, c_float, c_float, _P]),
    "rd_set_adam_state": (c_int32, [_P, c_float, c_float]),
    "rd_linear_fwd": (c_int32, [c_int32, c_int32, c_int32, _P, c_int32, _P, _P, _P, c_int32, c_int32, _P]),
    "rd_linear_fwd_fp32": (c_int32, [c_int32, c_int32, c_int32, _P, c_int32, _P, _P, _P, c_int32, c_int32, _P]),
    "rd_linear_bwd_input": (c_int32, [c_int32, c_int32, c_int32, _P, c_int32, _P, _P, c_int32, _P]),
    "rd_linear_bwd_input_gated": (c_int32, [c_int32, c_int32, c_int32, _P, c_int32, _P, _P, c_int32, _P, c_int32, _P]),
    "rd_softmax_xent": (c_int32, [c_int32, c_int32, _P, _P, _P, _P, _P]),
    "rd_set_rowgemm_rows32": (c_int32, [c_int32]),
    "rd_set_rowgemm_waves16": (c_int32, [c_int32]),
    "rd_head_train_supported": (c_int32, [c_int32, c_int32, c_int32]),
    "rd_head_train_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "rd_head_train": (c_int32, [_SHP, c_int32, c_int32, c_int32, c_int32] + [_P] * 20 + [_P, c_size_t, _P]),
    "rd_head_forward": (c_int32, [_SHP, c_int32, c_int32, c_int32, c_int32] + [_P] * 11 + [_P, c_size_t, _P]),
    "rd_head_backward": (c_int32, [_SHP, c_int32, c_int32, c_int32, c_int32] + [_P] * 18 + [_P, c_size_t, _P]),
    "rd_batch_gather": (c_int32, [c_int32, c_int32, c_int32, c_int32, ctypes.c_int64] + [_P] * 12),
    "rd_graph_beta_kept": (c_int32, [c_int32]),
    "rd_graph_beta_workspace_bytes": (c_size_t, [c_int32] * 5),
    "rd_graph_beta_fwd": (c_int32, [c_int32] * 6 + [_P, _P, _P, _P, ctypes.c_int64, _P, ctypes.c_int64, _P, ctypes.c_int64]
                          + [_P] * 5 + [_P, c_size_t, _P]),
    "rd_graph_beta_bwd": (c_int32, [c_int32] * 6 + [_P, _P, _P, _P, ctypes.c_int64, _P, ctypes.c_int64, _P, ctypes.c_int64]
                          + [_P] * 7 + [_P, c_size_t, _P]),
    "rd_structure_distance": (c_int32, [c_int32, c_int32, _P, _P, _P, _P]),
    "rd_prep_stats_workspace_bytes": (c_size_t, [ctypes.c_int64, c_int32]),
    "rd_prep_stats": (c_int32, [ctypes.c_int64, c_int32, _P, _P, _P, _P, c_size_t, _P]),
    "rd_prep_mask_normalize": (c_int32, [ctypes.c_int64, c_int32, c_int32, _P, _P, _P, _P, c_int32, _P]),
    "rd_prep_static": (c_int32, [ctypes.c_int64, c_int32, _P, _P, _P, _P, _P]),
    "rd_prep_time": (c_int32, [ctypes.c_int64, c_int32, _P, _P, c_int32, _P]),
    "rd_prep_remove_features": (c_int32, [ctypes.c_int64, c_int32, c_int32, _P, _P, c_int32, c_int32, c_int32, _P]),
    "rd_linear_bwd_weight_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "rd_linear_bwd_weight": (c_int32, [c_int32, c_int32, c_int32, _P, c_int32, _P, c_int32, _P, _P,
                                        _P, c_size_t, _P]),
}

_lib = None


def load():
    """Load the library once; raise RaindropHipError (never fall back) if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RaindropHipError(
            "libraindrop_hip.so not found at %s -- build it with `python -m raindrop_amd.build`; "
            "there is no CPU / eager fallback for the Raindrop hot path" % LIB_PATH)
    # PyTorch-ROCm bundles its own libamdhip64; it must be the ONE HIP runtime in the process
    # (streams and device pointers are shared with torch), so torch is imported before the dlopen.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise RaindropHipError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise RaindropHipError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    if lib.rd_version() != 1:
        raise RaindropHipError("ABI version mismatch: library %d, binding 1" % lib.rd_version())
    _lib = lib
    return lib


def call(name, *args):
    """Invoke an int-returning entry point and raise on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.rd_last_error().decode(errors="replace")
        kind = {-1: "RD_EINVAL", -2: "RD_EUNSUPPORTED"}.get(rc, "hipError %d" % rc)
        raise RaindropHipError("%s failed (%s): %s" % (name, kind, msg))


def shape(B, T, F, d_ob, d_pe=16, nhead=2, nhid=0, d_static=0, n_classes=2, max_len=None):
    return RdShape(B, T, F, d_ob, d_pe, nhead, nhid, d_static, n_classes, T if max_len is None else max_len)
