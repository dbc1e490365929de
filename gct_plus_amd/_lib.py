"""ctypes binding of libgctplus_hip.so (C ABI: include/gctplus_hip.h).

The library is mandatory: there is no CPU / eager-PyTorch fallback.  `load()` raises
if the shared object is missing or does not export every declared symbol.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GCT_LIB_PATH") or os.path.join(_HERE, "libgctplus_hip.so")   # override: A/B builds

P, I64, I32, F32, U64, U32 = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_uint64, C.c_uint32

# name -> (restype, [argtypes])   -- mirrors include/gctplus_hip.h one to one
SIGNATURES = {
    "gct_version": (I32, []),
    "gct_last_error": (C.c_char_p, []),
    "gct_wgrad_ws_bytes": (I64, [I64, I64, I64]),
    "gct_rowred_ws_bytes": (I64, [I64, I64]),
    "gct_embed_ws_bytes": (I64, [I32, I32, I32, I32]),
    "gct_norm_fwd": (I32, [P, P, P, P, P, P, I64, I32, F32, P]),
    "gct_norm_bwd": (I32, [P, P, P, P, P, P, P, P, P, P, I64, I32, F32, P, I64, P, F32, U64, U32, P]),
    "gct_embed_pe_fwd": (I32, [P, P, P, P, P, I32, I32, I32, I32, I32, F32, F32, U64, U32, P]),
    "gct_embed_pe_bwd": (I32, [P, P, P, P, P, I32, I32, I32, I32, I32, F32, F32, U64, U32, P]),
    "gct_linear_fwd": (I32, [P, I64, I64, I32, P, P, P, I64, P, P, P, I32, I32, P, P, P, I64,
                             I32, P, P, F32, U64, U32, P]),
    "gct_linear_fwd_ws_bytes": (I64, [I64, I32, I32]),
    "gct_linear_fwd_ws": (I32, [P, I64, I64, I32, P, P, P, I64, P, P, P, I32, I32, P, P, P, I64,
                                I32, P, P, F32, U64, U32, P, I64, P]),
    "gct_linear_fwd_p": (I32, [P, I64, I64, I32, P, P, P, I64, P, I64, P, P, P, I32, I32, P, P, P, I64,
                               I32, P, P, F32, U64, U32, P, I64, P, P]),
    "gct_linear_dgrad_p": (I32, [P, P, P, I64, I64, I32, I32, P, P, P, I64, P, I64, I32, P, I64, I32, P,
                                 F32, U64, U32, P, I64, P, I64, P]),
    "gct_linear_dgrad_ws_bytes": (I64, [I64, I32, I32]),
    "gct_gemm_set_mode": (I32, [I32]),
    "gct_gemm_get_mode": (I32, []),
    "gct_gemm_launch_counts": (I32, [P]),
    "gct_gemm_x6_kernel_launches": (I64, []),
    "gct_split_planes": (I32, [P, I64, P, I64, P]),
    "gct_linear_dgrad": (I32, [P, P, P, I64, I64, I32, I32, P, P, P, I64, I32, P, I64, I32, P,
                               F32, U64, U32, P]),
    "gct_linear_wgrad": (I32, [P, P, P, I64, I64, I32, I32, P, I64, I32, P, P, P, I64, P, P, P,
                               P, P]),
    "gct_dead_rows_nonzero": (I32, [P, I64, I64, I32, P, P, P]),
    "gct_nonzero_row_tiles": (I32, [P, I64, I64, I32, P, P, P, P]),
    "gct_live_rows": (I32, [P, I64, I32, I32, I32, P, I64, I64, P, P, P, P, P, P, P, P, P, P]),
    "gct_gather_quads": (I32, [P, I64, I64, P, I64, I32, P, I64, P]),
    "gct_scatter_quads": (I32, [P, I64, P, I64, I32, P, I64, I64, P]),
    "gct_zero_gap_rows": (I32, [P, I64, I32, P, P, I32, I64, P]),
    "gct_scatter_add_quads": (I32, [P, I64, P, I64, I32, P, I64, I64, P]),
    "gct_linear_wgrad_kt": (I32, [P, P, P, I64, I64, I32, I32, P, I64, I32, P, P, P, I64, P, P, P,
                                  P, P, P, P]),
    "gct_dropout_bwd": (I32, [P, P, I64, I32, F32, U64, U32, P, P]),
    "gct_attn_fwd": (I32, [P, I64, P, I64, P, I64, P, I64, I64, P, I64, P, P, I32, I32, I32, I32,
                           I32, F32, F32, U64, U32, P, P, P, I64, I64, P, P, P]),
    "gct_attn_bwd": (I32, [P, I64, P, I64, P, I64, P, I64, I64, P, P, I64, P, P, I64, P, I64,
                           P, I64, I32, I32, I32, I32, I32, F32, F32, U64, U32, P, P, I32, P, P, P, I64, I64, P, I64, P]),
    "gct_attn_bwd_ws_bytes": (I64, [I32, I32, I32, I32]),
    "gct_key_rows": (I32, [P, I64, I32, I32, P, P, P, P, P, P, P]),
    "gct_attn_mask_pack": (I32, [P, I64, I64, I32, I32, I32, P, P, P]),
    "gct_trg_mask_tokens": (I32, [P, I64, I64, I32, I32, P, P]),
    "gct_reparam_fwd": (I32, [P, P, P, P, P, I64, U64, U32, P]),
    "gct_reparam_bwd": (I32, [P, P, P, P, P, P, P, I64, P]),
    "gct_kld_fwd": (I32, [P, P, P, P, I64, P]),
    "gct_kld_bwd": (I32, [P, P, P, P, P, I64, P]),
    "gct_ce_fwd": (I32, [P, P, P, P, I64, I32, I64, P]),
    "gct_ce_bwd": (I32, [P, P, P, P, I64, I32, I64, P]),
    "gct_attn_decode": (I32, [P, I64, P, P, I64, I64, P, I64, P, I64, I32, I32, I32, I32, F32, P, I32, P, P, I64, P, P]),
    "gct_attn_decode_z": (I32, [P, I64, I32, P, I64, I32, I32, P, I64, I64, I32, P, I64, P, P, I64, I32, I32, I32, I32, F32, P]),
    "gct_decode_embed": (I32, [P, I64, P, I32, P, I32, P, P, I32, I32, F32, P]),
    "gct_decode_advance": (I32, [P, P]),
    "gct_select_token": (I32, [P, I32, P, I64, I32, P, I64, P, P, I32, I32, I64, I64, U64, P, I32, P, P]),
    "gct_smiles_tokenize": (I32, [C.c_char_p, I32, P, P, I32]),
    "gct_smiles_encode_batch": (I32, [P, I32, I32, P, I32, I64, I64, I64, I64, P, I64, P]),
    "gct_adam_step": (I32, [P, P, P, P, I64, F32, F32, F32, F32, I64, F32, P]),
    "gct_adam_step_guarded": (I32, [P, P, P, P, I64, F32, F32, F32, F32, I64, F32, P, P]),
    "gct_copy_rows": (I32, [P, I64, I64, P, I64, I64, I64, I64, I32, I32, P]),
    "gct_small_linear_fwd": (I32, [P, P, P, P, I32, I32, I32, P]),
    "gct_small_linear_bwd": (I32, [P, P, P, P, I32, I32, I32, P]),
    "gct_reduce_slabs": (I32, [P, I32, I64, P, I64, I32, P]),
    "gct_reduce_defer_begin": (I32, []),
    "gct_reduce_defer_flush": (I32, [P]),
    "gct_reduce_defer_end": (I32, [P]),
    "gct_reduce_defer_pending": (I32, []),
    "gct_add": (I32, [P, P, P, I64, P]),
}

ABI_VERSION = 13
_lib = None


class GctError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; hard error if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GctError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C gct_plus_amd/csrc`). gct_plus_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise GctError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.gct_version() != ABI_VERSION:
        raise GctError(f"ABI mismatch: library {lib.gct_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


# ---- the diagnostics library (include/gctplus_diag.h): separate from the operator boundary, optional at run time
DIAG_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgctplus_diag.so")
DIAG_SIGNATURES = {
    "gct_diag_last_error": (C.c_char_p, []),
    "gct_graph_probe": (I32, [I32, I32, I32, P, P, P]),
    "gct_device_facts": (I32, [C.c_char_p, I32]),
    "gct_graph_census": (I32, [P, P]),
    "gct_mfma_clock_probe": (I32, [I32, P, P]),
}
_diag = None


def load_diag():
    """ctypes handle of libgctplus_diag.so (GctError if it is not built: callers treat that as "no diagnosis")."""
    global _diag
    if _diag is not None:
        return _diag
    if not os.path.exists(DIAG_PATH):
        raise GctError(f"{DIAG_PATH} is missing (make -C gct_plus_amd/csrc)")
    lib = C.CDLL(DIAG_PATH)
    for name, (res, args) in DIAG_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _diag = lib
    return lib


def check_diag(rc: int, what: str):
    if rc != 0:
        raise GctError(f"{what} failed (rc={rc}): {load_diag().gct_diag_last_error().decode('utf-8', 'replace')}")


_TRACE = os.environ.get("GCT_TRACE_OPS", "0") != "0"    # debugging aid: name every launch on stderr and drain the device
                                                        # after it, so that a faulting kernel is the last name printed


def check(rc: int, what: str):
    if rc != 0:
        msg = load().gct_last_error().decode("utf-8", "replace")
        raise GctError(f"{what} failed (rc={rc}): {msg}")
    if _TRACE:
        import sys
        import torch
        print(f"[gct] {what}", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
