"""ctypes binding of the C-ABI library (include/echoglad_hip.h).

There is no CPU fallback: if the shared object is missing or does not load,
``load()`` raises and every op that needs it fails loudly."""
from __future__ import annotations

import ctypes as ct
import os
import re
from pathlib import Path
from typing import Dict, List, Optional

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("ECHOGLAD_LIB", _PKG / "lib" / "libechoglad_hip.so"))
HEADER_PATH = _PKG.parent / "include" / "echoglad_hip.h"

EG_OK, EG_ERR_ARG, EG_ERR_UNSUPPORTED, EG_ERR_HIP = 0, -1, -2, -3
ABI_VERSION = 140          # EG_ABI_VERSION of include/echoglad_hip.h that SIGNATURES below was written for

_lib: Optional[ct.CDLL] = None

_p = ct.c_void_p
_i = ct.c_int
_i64 = ct.c_int64

class ClsTrainParams(ct.Structure):
    """eg_cls_train_params of include/echoglad_hip.h (stacked parameters of the 4 classifier heads, train mode)."""
    _fields_ = [(n, _p) for n in ("w1", "b1", "gamma1", "beta1", "w2", "b2", "gamma2", "beta2", "w3", "b3",
                                  "running_mean1", "running_var1", "running_mean2", "running_var2")] + \
               [(n, ct.c_float) for n in ("eps1", "eps2", "momentum1", "momentum2", "p1", "p2")] + \
               [("seed1", ct.c_uint64), ("seed2", ct.c_uint64)]


_f = ct.c_float
_u64 = ct.c_uint64
_pp = ct.POINTER(ClsTrainParams)


class LowerSums(ct.Structure):
    """eg_lower_sums: the layer below the one whose dX launch takes its BatchNorm-backward sums (eg_gcn_layer_bwd_lower)."""
    _fields_ = [("z", _p), ("bn", _p), ("relu", _i), ("dropout_p", _f), ("seed", _u64), ("row_hi", _i64), ("tile_scratch", _p),
                ("sums_out", _p)]


class GivenSums(ct.Structure):
    """eg_given_sums: sums of THIS layer that somebody else has taken already (+ the bilinear backward's later additions)."""
    _fields_ = [("sums", _p), ("frames", _i), ("row_lo", _i64), ("n_valid", _i64), ("taps", _p)]


# name -> (restype, argtypes); must list every symbol the header declares
SIGNATURES: Dict[str, tuple] = {
    "eg_version": (_i, []),
    "eg_last_error": (ct.c_char_p, []),
    "eg_topo_create": (_i, [_i, _i, _i, _i, _i, _i, _i, ct.POINTER(_p)]),
    "eg_csr_create": (_i, [_p, _i64, _i64, _p, ct.POINTER(_p)]),
    "eg_graph_is_symmetric": (_i, [_p]),
    "eg_csr_create_transposed": (_i, [_p, _p, _i64, _p, ct.POINTER(_p)]),
    "eg_graph_destroy": (_i, [_p]),
    "eg_graph_num_nodes": (_i64, [_p]),
    "eg_graph_is_structured": (_i, [_p]),
    "eg_graph_num_tiles": (_i64, [_p]),
    "eg_graph_deg_inv_sqrt": (_i, [_p, _p, _p]),
    "eg_edge_hash": (_i, [_p, _i64, _p, _p]),
    "eg_debug_xcc": (_i, [_p, _i, _p]),
    "eg_dropout_epoch_add": (_i, [ct.c_uint64, _p]),
    "eg_dropout_epoch_set": (_i, [ct.c_uint64, _p]),
    "eg_debug_dropout_epoch": (_i, [ct.POINTER(ct.c_uint64)]),
    "eg_debug_layer_timing_begin": (_i, [_i]),
    "eg_debug_layer_timing_end": (_i, [ct.POINTER(ct.c_float), ct.POINTER(ct.c_int), _i]),
    "eg_debug_phase_cycles": (_i, [_p, ct.POINTER(ct.c_uint64), _i]),
    "eg_gcn_layer_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p]),
    "eg_graph_kidsum_rows": (_i64, [_p]),
    "eg_graph_fused_classifier_ok": (_i, [_p]),
    "eg_gcn_layer_fwd_chain": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p]),
    "eg_gcn_layer_cls_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p]),
    "eg_gcn_layer_fwd_jk": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p]),
    "eg_graph_ps_launches": (ct.c_uint, [_p]),
    "eg_graph_layer_launches": (ct.c_uint, [_p]),
    "eg_gcn_aggregate": (_i, [_p, _i, _p, _p, _p]),
    "eg_linear128_fwd": (_i, [_p, _i64, _p, _p, _p, _p, _i, _i, _p, _p]),
    "eg_classifier_fwd": (_i, [_p, _i, _i64, _i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p]),
    "eg_workspace_bytes": (ct.c_size_t, []),
    "eg_colsum128": (_i, [_p, _i64, _p, _p, _p]),
    "eg_dweight128": (_i, [_p, _p, _i64, _p, _p, _p]),
    "eg_bn_stats": (_i, [_p, _i64, _p, _p, _p, _p]),
    "eg_bn_act_fwd": (_i, [_p, _i64, _p, _p, _p, _i, ct.c_float, ct.c_uint64, _p, _p]),
    "eg_bn_act_fwd_tiles": (_i, [_p, _i, _p, _p, _p, _p, _i, ct.c_float, ct.c_uint64, _p, _p, _p]),
    "eg_bn_act_bwd": (_i, [_p, _p, _i64, _p, _p, _p, _p, _i, ct.c_float, ct.c_uint64, _p, _p, _p, _p, _p]),
    "eg_gcn_layer_train_fwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _f, _f, _i, _f, _u64, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "eg_gcn_layer_bwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _f, _u64, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "eg_classifier_train_workspace_bytes": (ct.c_size_t, []),
    "eg_classifier_train_fwd": (_i, [_p, _i, _i64, _i64, _i64, _pp, _p, _p, _p, _p, _i, _p, _p]),
    "eg_classifier_train_fwd_act": (_i, [_p, _p, _p, _i, _f, _u64, _p, _i, _i64, _i64, _i64, _pp, _p, _p, _p, _p, _i, _p, _i, _p]),
    "eg_classifier_bwd": (_i, [_p, _p, _i, _i64, _i64, _i64, _pp, _p, _p, _p, _p, _p, _p, _p, _p]),
    "eg_classifier_bwd_sums": (_i, [_p, _p, _i, _i64, _i64, _i64, _pp, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _u64, _p, _p, _i, _p]),
    "eg_gcn_layer_bwd_presummed": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _f, _u64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i,
                                        _i64, _i64, _p]),
    "eg_gcn_layer_bwd_lower": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _f, _u64, _i, _p, _p, _p, _p, _p, _p, _p,
                                    ct.POINTER(GivenSums), ct.POINTER(LowerSums), _p]),
    "eg_bilinear4_bwd_rows_sums": (_i, [_p, _i64, _p, _p, _i, _i, _i64, _i64, _i, _p, _p, ct.POINTER(LowerSums), _p, _p]),
    "eg_coord_mlp_fwd": (_i, [_p, _p, _i, _pp, _i, _i, _p, _p, _p, _p, _p, _p]),
    "eg_coord_mlp_bwd": (_i, [_p, _p, _p, _i, _pp, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "eg_coord_mlp_fwd_rows": (_i, [_p, _i64, _p, _p, _i, _pp, _i, _i, _p, _p, _p, _p, _p, _p]),
    "eg_coord_mlp_bwd_rows": (_i, [_p, _p, _p, _i, _pp, _i, _p, _p, _p, _p, _p, _p, _i64, _i, _p, _p, _p]),
    "eg_adam_step": (_i, [_p, _i, _p, ct.c_float, _p, ct.c_float, ct.c_float, ct.c_float, ct.c_float, _i, _p]),
    "eg_coord_update_fwd": (_i, [_p, _i64, _i64, _i64, _p, _i, _pp, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "eg_coord_update_bwd": (_i, [_p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _i, _pp, _i, _p, _p, _p, _p, _p, _p, ct.POINTER(LowerSums), _p, _p, _p, _p]),
    "eg_bilinear4_fwd_rows": (_i, [_p, _p, _i, _i, _i64, _i64, _i, _p, _i64, _p]),
    "eg_bilinear4_bwd_rows": (_i, [_p, _i64, _p, _p, _i, _i, _i64, _i64, _i, _p, _p, _p]),
    "eg_bilinear4_fwd": (_i, [_p, _p, _i, _i, _i64, _i64, _i, _p, _p]),
    "eg_bilinear4_bwd": (_i, [_p, _p, _p, _i, _i, _i64, _i64, _i, _p, _p, _p]),
    "eg_pack_levels": (_i, [ct.POINTER(_p), ct.POINTER(_i), _i, _i, _i64, _i64, _p, _p]),
    "eg_conv1x1_relu_pack_levels": (_i, [ct.POINTER(_p), ct.POINTER(_p), ct.POINTER(_p), ct.POINTER(_i), ct.POINTER(_i), _i, _i,
                                         _i64, _i64, _p, _p]),
    "eg_avg_pool_pyramid_fwd": (_i, [_p, _i64, _i, ct.POINTER(_i), _i, ct.POINTER(_p), _p]),
    "eg_avg_pool_pyramid_bwd": (_i, [ct.POINTER(_p), _p, _i64, _i, ct.POINTER(_i), _i, _p, _p]),
    "eg_unpack_levels": (_i, [_p, ct.POINTER(_p), ct.POINTER(_i), _i, _i, _i64, _i64, _p]),
    "eg_heatmap_workspace_bytes": (ct.c_size_t, [_i, ct.POINTER(_i), _i]),
    "eg_heatmap_expect_fwd": (_i, [_p, _p, _p, _i, _i64, ct.POINTER(_i), ct.POINTER(_i), _i, _p, _p, _p, _p, _p, _p, _p]),
    "eg_heatmap_expect_bwd": (_i, [_p, _p, _p, _p, _i, _i64, ct.POINTER(_i), ct.POINTER(_i), _i, _p, _p]),
    "eg_criteria_workspace_bytes": (ct.c_size_t, [_i, ct.POINTER(_i), _i]),
    "eg_criteria_fwd": (_i, [_p, _p, _p, _i, _i64, ct.POINTER(_i), ct.POINTER(_i), _i, _p, _f, _f, _f, _p, _p, _i64, _f, _p, _p, _p, _p,
                             _p, _p, _p, _p, _p, _p, _p]),
    "eg_criteria_bwd": (_i, [_p, _p, _p, _i, _i64, ct.POINTER(_i), ct.POINTER(_i), _i, _f, _p, _p, _p, _p, _p, _i64, _p, _p, _p, _p,
                             _p, _p, _p]),
    "eg_elm_reduce": (_i, [_p, _p, _p, _p, _i, _i, ct.c_float, _p, _p, _p]),
    "eg_bce_logits_fwd": (_i, [_p, _p, _p, _i64, ct.c_float, _p, _p, _p]),
    "eg_bce_logits_bwd": (_i, [_p, _p, _p, _i64, ct.c_float, _p, _p, _p]),
}


def header_symbols() -> List[str]:
    """Every function name declared in include/echoglad_hip.h."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eg_[a-z0-9_]+)\s*\(", text)))


def load() -> ct.CDLL:
    """Load the library (after torch, so both share one HIP runtime instance)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  — must come first: libamdhip64.so.7 resolves to the copy torch loaded
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -m echoglad_amd.build, or __graft_entry__.build()). There is no CPU fallback.")
    lib = ct.CDLL(str(LIB_PATH), mode=ct.RTLD_GLOBAL if hasattr(ct, "RTLD_GLOBAL") else 0)
    # a library built from other sources than this binding would take the calls below with shifted arguments: refuse it
    try:
        lib.eg_version.restype = _i
        lib.eg_version.argtypes = []
        got = int(lib.eg_version())
    except AttributeError:
        got = None
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} reports eg_version() = {got}, this binding is written for ABI {ABI_VERSION}: rebuild the "
                           "library (python -m echoglad_amd.build --force) or point ECHOGLAD_LIB at a matching build")
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            continue          # checked by check_exports(); ops fail on use
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check_exports() -> List[str]:
    """Names declared in the header that the library does not export."""
    lib = load()
    return [n for n in header_symbols() if not hasattr(lib, n)]


def last_error() -> str:
    msg = load().eg_last_error()
    return msg.decode() if msg else ""


def check(rc: int, what: str) -> None:
    if rc != EG_OK:
        kind = {EG_ERR_ARG: "bad argument", EG_ERR_UNSUPPORTED: "unsupported", EG_ERR_HIP: "HIP error"}.get(rc, str(rc))
        raise RuntimeError(f"{what} failed ({kind}): {last_error()}")
