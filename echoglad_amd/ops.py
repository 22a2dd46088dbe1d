"""Thin Python wrappers over the C-ABI (no compute here; pointers + stream only).

Every function enqueues on ``torch.cuda.current_stream()`` and returns torch
tensors that own the output memory.  Inputs must be CUDA fp32 contiguous
``[rows, 128]``."""
from __future__ import annotations

import ctypes as ct
import os
from typing import Optional

import torch

from . import _lib

C = 128


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ct.c_void_p(t.data_ptr())


def _stream(t: Optional[torch.Tensor] = None):
    """The caller's current stream on the device of ``t`` (default: the current device).  The kernels launch on the
    device that is current for the calling thread, so a tensor on another device is an error, not a silent cross-device
    launch."""
    if t is not None and t.is_cuda and t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: "
                           "wrap the call in torch.cuda.device(tensor.device)")
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_rows(t: torch.Tensor, name: str, rows: Optional[int] = None) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (ROCm) tensor: the HIP path has no CPU fallback")
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != C or not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous float32 [rows, {C}], got {tuple(t.shape)} {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError(f"{name} is on {t.device} but the current device is cuda:{torch.cuda.current_device()} "
                           "(kernels launch on the current device: use torch.cuda.device(tensor.device))")
    if rows is not None and t.shape[0] != rows:
        raise RuntimeError(f"{name} has {t.shape[0]} rows, expected {rows}")


def _check_vec(t: Optional[torch.Tensor], name: str, n: int) -> None:
    if t is None:
        return
    if not t.is_cuda or t.dtype != torch.float32 or t.numel() != n or not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous CUDA float32 tensor with {n} elements")


class Graph:
    """Owns an ``eg_graph`` handle (closed-form topology or generic CSR)."""

    def __init__(self, handle: ct.c_void_p, structured: bool, num_nodes: int, device: torch.device):
        self._h = handle
        self.structured = structured
        self.num_nodes = num_nodes
        self.device = device
        # rows per frame of a child-sum side buffer (chained layers); 0 = not available for this handle
        self.kidsum_rows = int(_lib.load().eg_graph_kidsum_rows(handle)) if structured else 0
        self.num_tiles = int(_lib.load().eg_graph_num_tiles(handle))          # 64-row work tiles per frame
        # a closed-form topology with 'grid-diagonal' levels: stencil in the producer/consumer kernel, per-frame CSR elsewhere
        self.hybrid = False
        self.num_conn = 0           # connection nodes at the head of every frame (rows the heads' node-type filter drops)
        # the handle whose aggregation is A_hat^T (what a backward pass needs): the handle itself unless edge_index is directed
        self.bwd: "Graph" = self

    @classmethod
    def topo(cls, frame_size: int, num_aux_graphs: int, use_main_graph_only: bool = False,
             use_coordinate_graph: bool = False, device=None, use_connection_nodes: bool = False, diag_main: bool = False,
             diag_aux: bool = False) -> "Graph":
        """Implicit-stencil handle of the closed-form topology ('grid' or 'grid-diagonal' levels, with or without coordinate /
        connection nodes: every flag of the reference's builder)."""
        lib = _lib.load()
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = ct.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.eg_topo_create(frame_size, num_aux_graphs, int(use_main_graph_only),
                                          int(use_coordinate_graph), int(use_connection_nodes), int(diag_main), int(diag_aux),
                                          ct.byref(h)), "eg_topo_create")
        g = cls(h, True, int(lib.eg_graph_num_nodes(h)), device)
        g.hybrid = bool(diag_main or ((diag_aux or use_connection_nodes) and not use_main_graph_only))
        g.num_conn = (num_aux_graphs + 1) if (use_connection_nodes and not use_main_graph_only) else 0
        return g

    @classmethod
    def csr(cls, edge_index: torch.Tensor, num_nodes: int) -> "Graph":
        lib = _lib.load()
        if not edge_index.is_cuda or edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise RuntimeError("edge_index must be a CUDA int64 tensor of shape [2, E]")
        ei = edge_index.contiguous()
        h = ct.c_void_p()
        with torch.cuda.device(ei.device):
            _lib.check(lib.eg_csr_create(_ptr(ei), int(num_nodes), int(ei.shape[1]), _stream(), ct.byref(h)),
                       "eg_csr_create")
            g = cls(h, False, int(num_nodes), ei.device)
            if not lib.eg_graph_is_symmetric(h):             # directed graph: the backward aggregates over out-edges
                ht = ct.c_void_p()
                _lib.check(lib.eg_csr_create_transposed(h, _ptr(ei), int(ei.shape[1]), _stream(), ct.byref(ht)),
                           "eg_csr_create_transposed")
                g.bwd = cls(ht, False, int(num_nodes), ei.device)
        return g

    @property
    def fused_classifier_ok(self) -> bool:
        """eg_gcn_layer_cls_fwd (last layer + classifier heads in one kernel) is available for this handle."""
        return bool(_lib.load().eg_graph_fused_classifier_ok(self._h)) if self.structured else False

    @property
    def ps_launches(self) -> int:
        """Launches of the producer/consumer (fused, chained) layer kernel on this handle so far."""
        return int(_lib.load().eg_graph_ps_launches(self._h))

    @property
    def layer_launches(self) -> int:
        """Launches of either fused layer kernel (producer/consumer or symmetric) on this handle so far."""
        return int(_lib.load().eg_graph_layer_launches(self._h))

    def deg_inv_sqrt(self) -> torch.Tensor:
        out = torch.empty(self.num_nodes, dtype=torch.float32, device=self.device)
        _lib.check(_lib.load().eg_graph_deg_inv_sqrt(self._h, _ptr(out), _stream()), "eg_graph_deg_inv_sqrt")
        return out

    def close(self) -> None:
        if self._h is not None and self._h.value:
            _lib.load().eg_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


LAUNCH_KINDS = ("symmetric", "ps_plain", "ps_train_fwd", "ps_dx", "ps_cls")      # EG_LAUNCH_* of include/echoglad_hip.h


def dropout_epoch_add(delta: int = 1) -> None:
    """eg_dropout_epoch_add on the current stream: every mask generated by a later launch hashes with seed + (epoch += delta).
    Capturable -- the first node of engine.GraphedTrainStep's graph."""
    _lib.check(_lib.load().eg_dropout_epoch_add(int(delta) & 0xFFFFFFFFFFFFFFFF, _stream()), "eg_dropout_epoch_add")


def dropout_epoch_set(value: int) -> None:
    _lib.check(_lib.load().eg_dropout_epoch_set(int(value) & 0xFFFFFFFFFFFFFFFF, _stream()), "eg_dropout_epoch_set")


def dropout_epoch() -> int:
    """The device's dropout epoch (synchronises; tests)."""
    out = ct.c_uint64(0)
    _lib.check(_lib.load().eg_debug_dropout_epoch(ct.byref(out)), "eg_debug_dropout_epoch")
    return int(out.value)


class layer_timing:
    """Context manager over eg_debug_layer_timing_begin / _end: HIP events around every fused-layer kernel launch issued inside
    the block (up to `max_launches`), wherever it is issued from (autograd nodes included).  After the block, ``.launches`` is a
    list of (kind, milliseconds).  A measurement tool (process-wide state), used by bench.py's roofline."""

    def __init__(self, max_launches: int = 256):
        self.max = int(max_launches)
        self.launches = []

    def __enter__(self):
        _lib.check(_lib.load().eg_debug_layer_timing_begin(self.max), "eg_debug_layer_timing_begin")
        return self

    def __exit__(self, *exc):
        ms = (ct.c_float * 256)()
        kinds = (ct.c_int * 256)()
        n = _lib.load().eg_debug_layer_timing_end(ms, kinds, 256)
        if n < 0:
            _lib.check(n, "eg_debug_layer_timing_end")
        self.launches = [(LAUNCH_KINDS[kinds[i]], float(ms[i])) for i in range(n)]
        return False

    def mean_ms(self, kind: str) -> Optional[float]:
        v = [m for k, m in self.launches if k == kind]
        return sum(v) / len(v) if v else None


def edge_hash(edge_index: torch.Tensor):
    """(E, order-independent 64-bit digest) of a device edge_index; one host sync."""
    lib = _lib.load()
    ei = edge_index.contiguous()
    out = torch.empty(2, dtype=torch.int64, device=ei.device)
    _lib.check(lib.eg_edge_hash(_ptr(ei), int(ei.shape[1]), _ptr(out), _stream()), "eg_edge_hash")
    e, h = out.cpu().tolist()
    return int(e), int(h) & 0xFFFFFFFFFFFFFFFF


def gcn_layer_fwd(graph: Graph, batch: int, x: torch.Tensor, weight: torch.Tensor,
                  scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
                  residual: Optional[torch.Tensor] = None, relu: bool = False, transpose_w: bool = False,
                  out: Optional[torch.Tensor] = None, kidsum_in: Optional[torch.Tensor] = None,
                  kidsum_out: Optional[torch.Tensor] = None, jk_in: Optional[torch.Tensor] = None,
                  jk_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act((A_hat x) W^T * scale + shift) + residual in one kernel.

    kidsum_in / kidsum_out: child-sum side buffers of a chained stack of layers (see `new_kidsum`,
    include/echoglad_hip.h eg_gcn_layer_fwd_chain).  jk_in / jk_out: running JumpingKnowledge('max') maximum,
    jk_out = max(jk_in, result) (eg_gcn_layer_fwd_jk; the first layer passes x as jk_in)."""
    rows = graph.num_nodes * batch
    _check_rows(x, "x", rows)
    if weight.shape != (C, C) or not weight.is_cuda or weight.dtype != torch.float32 or not weight.is_contiguous():
        raise RuntimeError("weight must be a contiguous CUDA float32 [128, 128] tensor")
    _check_vec(scale, "scale", C)
    _check_vec(shift, "shift", C)
    if residual is not None:
        _check_rows(residual, "residual", rows)
    if out is None:
        out = torch.empty_like(x)
    else:
        _check_rows(out, "out", rows)
    if jk_in is not None or jk_out is not None:
        if jk_in is None or jk_out is None or transpose_w:
            raise RuntimeError("jk_in and jk_out go together (and not with transpose_w)")
        _check_rows(jk_in, "jk_in", rows)
        _check_rows(jk_out, "jk_out", rows)
        for name, t in (("kidsum_in", kidsum_in), ("kidsum_out", kidsum_out)):
            if t is not None:
                _check_rows(t, name, graph.kidsum_rows * batch)
        _lib.check(_lib.load().eg_gcn_layer_fwd_jk(graph._h, batch, _ptr(x), _ptr(weight), _ptr(scale), _ptr(shift), _ptr(residual),
                                                   int(relu), _ptr(out), _ptr(kidsum_in), _ptr(kidsum_out), _ptr(jk_in), _ptr(jk_out),
                                                   _stream()), "eg_gcn_layer_fwd_jk")
        return out
    if kidsum_in is not None or kidsum_out is not None:
        krows = graph.kidsum_rows * batch
        for name, t in (("kidsum_in", kidsum_in), ("kidsum_out", kidsum_out)):
            if t is not None:
                if krows == 0:
                    raise RuntimeError("this graph handle has no child-sum side buffer (kidsum_rows == 0)")
                _check_rows(t, name, krows)
        _lib.check(_lib.load().eg_gcn_layer_fwd_chain(graph._h, batch, _ptr(x), _ptr(weight), _ptr(scale), _ptr(shift),
                                                      _ptr(residual), int(relu), int(transpose_w), _ptr(out),
                                                      _ptr(kidsum_in), _ptr(kidsum_out), _stream()),
                   "eg_gcn_layer_fwd_chain")
        return out
    _lib.check(_lib.load().eg_gcn_layer_fwd(graph._h, batch, _ptr(x), _ptr(weight), _ptr(scale), _ptr(shift),
                                            _ptr(residual), int(relu), int(transpose_w), _ptr(out), _stream()),
               "eg_gcn_layer_fwd")
    return out


def gcn_layer_cls_fwd(graph: Graph, batch: int, x: torch.Tensor, weight: torch.Tensor, scale, shift, residual, relu: bool,
                      packed: dict, sigmoid: bool = False, kidsum_in: Optional[torch.Tensor] = None,
                      jk_in: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Last layer + node-type filter + 4 classifier heads in one kernel -> logits [batch * (num_nodes - num_conn), 4] (the
    connection nodes at the head of every frame have no logits row).
    jk_in: running JumpingKnowledge('max') maximum of the earlier embeddings: the heads then see max(jk_in, layer output)."""
    rows = graph.num_nodes * batch
    _check_rows(x, "x", rows)
    if weight.shape != (C, C) or not weight.is_cuda or weight.dtype != torch.float32 or not weight.is_contiguous():
        raise RuntimeError("weight must be a contiguous CUDA float32 [128, 128] tensor")
    _check_vec(scale, "scale", C)
    _check_vec(shift, "shift", C)
    if residual is not None:
        _check_rows(residual, "residual", rows)
    if kidsum_in is not None:
        _check_rows(kidsum_in, "kidsum_in", graph.kidsum_rows * batch)
    if jk_in is not None:
        _check_rows(jk_in, "jk_in", rows)
    out = torch.empty((graph.num_nodes - graph.num_conn) * batch, 4, dtype=torch.float32, device=x.device)
    p = packed
    _lib.check(_lib.load().eg_gcn_layer_cls_fwd(graph._h, batch, _ptr(x), _ptr(weight), _ptr(scale), _ptr(shift), _ptr(residual),
                                                int(relu), _ptr(kidsum_in), _ptr(jk_in), _ptr(p["w1"]), _ptr(p["s1"]), _ptr(p["t1"]),
                                                _ptr(p["w2"]), _ptr(p["s2"]), _ptr(p["t2"]), _ptr(p["w3"]), _ptr(p["b3"]),
                                                int(sigmoid), _ptr(out), _stream()), "eg_gcn_layer_cls_fwd")
    return out


def new_kidsum(graph: Graph, batch: int) -> Optional[torch.Tensor]:
    """Zero-filled child-sum side buffer [batch * kidsum_rows, 128] for chained layers, or None when the
    topology does not qualify (generic CSR handles, irregular frames)."""
    rows = graph.kidsum_rows
    if rows == 0:
        return None
    return torch.zeros(rows * batch, C, device=graph.device, dtype=torch.float32)


def gcn_aggregate(graph: Graph, batch: int, x: torch.Tensor) -> torch.Tensor:
    """A_hat x (symmetric-normalised adjacency with self loops)."""
    _check_rows(x, "x", graph.num_nodes * batch)
    out = torch.empty_like(x)
    _lib.check(_lib.load().eg_gcn_aggregate(graph._h, batch, _ptr(x), _ptr(out), _stream()), "eg_gcn_aggregate")
    return out


def linear128_fwd(x: torch.Tensor, weight: torch.Tensor, scale=None, shift=None, residual=None, relu: bool = False,
                  transpose_w: bool = False) -> torch.Tensor:
    _check_rows(x, "x")
    _check_vec(scale, "scale", C)
    _check_vec(shift, "shift", C)
    if residual is not None:
        _check_rows(residual, "residual", x.shape[0])
    out = torch.empty_like(x)
    _lib.check(_lib.load().eg_linear128_fwd(_ptr(x), int(x.shape[0]), _ptr(weight.contiguous()), _ptr(scale),
                                            _ptr(shift), _ptr(residual), int(relu), int(transpose_w), _ptr(out),
                                            _stream()), "eg_linear128_fwd")
    return out


def classifier_fwd(h: torch.Tensor, batch: int, n_per_frame: int, row_lo: int, n_valid: int, packed: dict,
                   sigmoid: bool = False) -> torch.Tensor:
    """node-type filter (contiguous row range per frame) + the 4 heads -> [batch*n_valid, 4]."""
    _check_rows(h, "h", batch * n_per_frame)
    out = torch.empty(batch * n_valid, 4, dtype=torch.float32, device=h.device)
    p = packed
    _lib.check(_lib.load().eg_classifier_fwd(_ptr(h), batch, n_per_frame, row_lo, n_valid, _ptr(p["w1"]),
                                             _ptr(p["s1"]), _ptr(p["t1"]), _ptr(p["w2"]), _ptr(p["s2"]),
                                             _ptr(p["t2"]), _ptr(p["w3"]), _ptr(p["b3"]), int(sigmoid), _ptr(out),
                                             _stream()), "eg_classifier_fwd")
    return out


# ---------------------------------------------------------------------------
# training-mode pieces
# ---------------------------------------------------------------------------
_workspaces = {}


def _workspace(device) -> torch.Tensor:
    """One reduction workspace per (device, stream); the C side never allocates."""
    key = (torch.device(device), torch.cuda.current_stream().cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        nbytes = int(_lib.load().eg_workspace_bytes())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def colsum128(x: torch.Tensor) -> torch.Tensor:
    _check_rows(x, "x")
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().eg_colsum128(_ptr(x), int(x.shape[0]), _ptr(_workspace(x.device)), _ptr(out), _stream()),
               "eg_colsum128")
    return out


def dweight128(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """g^T x -> [128(out), 128(in)]"""
    _check_rows(g, "g")
    _check_rows(x, "x", g.shape[0])
    out = torch.empty(C, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().eg_dweight128(_ptr(g), _ptr(x), int(x.shape[0]), _ptr(_workspace(x.device)), _ptr(out),
                                         _stream()), "eg_dweight128")
    return out


def bn_stats(x: torch.Tensor):
    """(mean[128], biased var[128]) over all rows."""
    _check_rows(x, "x")
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    var = torch.empty(C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().eg_bn_stats(_ptr(x), int(x.shape[0]), _ptr(_workspace(x.device)), _ptr(mean), _ptr(var),
                                       _stream()), "eg_bn_stats")
    return mean, var


def bn_act_fwd(z, scale, shift, residual=None, relu=False, dropout_p=0.0, seed=0) -> torch.Tensor:
    _check_rows(z, "z")
    _check_vec(scale, "scale", C)
    _check_vec(shift, "shift", C)
    if residual is not None:
        _check_rows(residual, "residual", z.shape[0])
    out = torch.empty_like(z)
    _lib.check(_lib.load().eg_bn_act_fwd(_ptr(z), int(z.shape[0]), _ptr(scale), _ptr(shift), _ptr(residual), int(relu),
                                         float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(out), _stream()),
               "eg_bn_act_fwd")
    return out


def bn_act_fwd_tiles(graph: Graph, batch: int, z, scale, shift, residual=None, relu=False, dropout_p=0.0, seed=0,
                     kidsum_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bn_act_fwd in the layer kernels' tile order; kidsum_out (optional) receives the child sums of the result."""
    _check_rows(z, "z", graph.num_nodes * batch)
    _check_vec(scale, "scale", C)
    _check_vec(shift, "shift", C)
    if residual is not None:
        _check_rows(residual, "residual", z.shape[0])
    if kidsum_out is not None:
        _check_rows(kidsum_out, "kidsum_out", graph.kidsum_rows * batch)
    out = torch.empty_like(z)
    _lib.check(_lib.load().eg_bn_act_fwd_tiles(graph._h, batch, _ptr(z), _ptr(scale), _ptr(shift), _ptr(residual), int(relu),
                                               float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(out), _ptr(kidsum_out),
                                               _stream()), "eg_bn_act_fwd_tiles")
    return out


def bn_act_bwd(dy, z, mean, invstd, gamma, beta, relu=False, dropout_p=0.0, seed=0):
    """-> (dz, dgamma, dbeta)"""
    _check_rows(dy, "dy")
    _check_rows(z, "z", dy.shape[0])
    for t, n in ((mean, "mean"), (invstd, "invstd"), (gamma, "gamma"), (beta, "beta")):
        _check_vec(t, n, C)
    dz = torch.empty_like(z)
    dgamma = torch.empty(C, dtype=torch.float32, device=z.device)
    dbeta = torch.empty(C, dtype=torch.float32, device=z.device)
    _lib.check(_lib.load().eg_bn_act_bwd(_ptr(dy), _ptr(z), int(z.shape[0]), _ptr(mean), _ptr(invstd), _ptr(gamma),
                                         _ptr(beta), int(relu), float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                         _ptr(_workspace(z.device)), _ptr(dz), _ptr(dgamma), _ptr(dbeta), _stream()),
               "eg_bn_act_bwd")
    return dz, dgamma, dbeta


# ---------------------------------------------------------------------------
# whole train-mode layers (eg_gcn_layer_train_fwd / _bwd, eg_classifier_train_fwd / _bwd)
# ---------------------------------------------------------------------------
def train_chain_supported() -> bool:
    """The train forward hands child sums from layer to layer (kidsum_in / kidsum_out of eg_gcn_layer_train_fwd)."""
    return True


def gcn_layer_train_fwd(graph: Graph, batch: int, x, weight, bias, gamma, beta, running_mean, running_var, momentum,
                        eps: float, relu: bool, dropout_p: float, seed: int, residual: bool, want_agg: bool = True,
                        kidsum_in: Optional[torch.Tensor] = None, kidsum_out: Optional[torch.Tensor] = None,
                        want_out: bool = True):
    """-> (out, z, agg | None, bn [4,128] = mean, invstd, scale, shift).  running_* are updated in place
    (momentum None: no update).  kidsum_in / kidsum_out: child-sum side buffers of a chained train forward (`new_kidsum`)."""
    rows = graph.num_nodes * batch
    _check_rows(x, "x", rows)
    for name, t in (("kidsum_in", kidsum_in), ("kidsum_out", kidsum_out)):
        if t is not None:
            if graph.kidsum_rows == 0:
                raise RuntimeError("this graph handle has no child-sum side buffer (kidsum_rows == 0)")
            _check_rows(t, name, graph.kidsum_rows * batch)
    for t, n in ((bias, "bias"), (gamma, "gamma"), (beta, "beta")):
        _check_vec(t, n, C)
    z = torch.empty_like(x)
    out = torch.empty_like(x) if want_out else None           # None: z, agg and the statistics only (no activation pass)
    agg = torch.empty_like(x) if want_agg else None
    bn = torch.empty(4, C, dtype=torch.float32, device=x.device)
    upd = momentum is not None and running_mean is not None
    _lib.check(_lib.load().eg_gcn_layer_train_fwd(
        graph._h, batch, _ptr(x), _ptr(weight), _ptr(bias), _ptr(gamma), _ptr(beta), _ptr(running_mean) if upd else None,
        _ptr(running_var) if upd else None, float(momentum) if upd else -1.0, float(eps), int(relu), float(dropout_p),
        int(seed) & 0xFFFFFFFFFFFFFFFF, int(residual), _ptr(_workspace(x.device)), _ptr(z), _ptr(agg), _ptr(bn), _ptr(out),
        _ptr(kidsum_in), _ptr(kidsum_out), _stream()), "eg_gcn_layer_train_fwd")
    return out, z, agg, bn


_tile_scratch = {}


def _lower_scratch(graph: Graph, batch: int, device) -> torch.Tensor:
    """[tiles, 2, 128] floats for the per-tile partials of eg_gcn_layer_bwd_lower (one buffer per device, stream and size)."""
    n = graph.num_tiles * batch * 2 * C
    key = (torch.device(device), torch.cuda.current_stream().cuda_stream)
    buf = _tile_scratch.get(key)
    if buf is None or buf.numel() < n:
        buf = _tile_scratch[key] = torch.empty(n, dtype=torch.float32, device=device)
    return buf


def lower_sums_supported(graph_bwd: Graph) -> bool:
    """May the dX launch on this handle also take the BatchNorm-backward sums of the layer below (eg_gcn_layer_bwd_lower)?"""
    return bool(graph_bwd.structured) and os.environ.get("EG_TRAIN_PS", "1") != "0"


def gcn_layer_bwd(graph_bwd: Graph, batch: int, dy, z, agg, weight, gamma, beta, bn, relu: bool, dropout_p: float, seed: int,
                  residual: bool, need_dx: bool, need_dw: bool, dy_sums=None, lower=None):
    """-> (dx | None, dw | None, db | None (zeros), dgamma, dbeta)   [+ lower_sums when ``lower`` is given].
    dy_sums = (sums [256] float64, frames, row_lo, n_valid[, taps]) from classifier_bwd(..., layer=...) or from the dX launch of
    the layer above (``lower``): the BatchNorm-backward sums over those rows of every frame are given, the layer's own sums pass
    only adds the other rows (eg_gcn_layer_bwd_presummed / _lower); taps [frames, 2, 128]: bilinear4_bwd(..., lower=)'s sums.
    lower = (z, bn, relu, dropout_p, seed, row_hi) of the layer BELOW: the dX launch takes its BatchNorm-backward sums over rows
    [0, row_hi) of every frame from the rows it writes (eg_gcn_layer_bwd_lower) -> 6th result, float64 [256]."""
    rows = graph_bwd.num_nodes * batch
    _check_rows(dy, "dy", rows)
    if lower is not None:
        if not (need_dx and residual and lower_sums_supported(graph_bwd)):
            raise RuntimeError("lower sums go with the producer / consumer kernel's dX launch (ops.lower_sums_supported, need_dx, residual)")
        lz, lbn, lrelu, lp, lseed, row_hi = lower
        _check_rows(lz, "lower z", rows)
        if lbn.numel() != 4 * C or not lbn.is_cuda or not lbn.is_contiguous() or lbn.dtype != torch.float32:
            raise RuntimeError("lower bn must be the [4,128] float32 tensor of gcn_layer_train_fwd")
        dz = torch.empty_like(dy)
        dx = torch.empty_like(dy)
        dw = torch.empty(C, C, dtype=torch.float32, device=dy.device) if need_dw else None
        small = torch.empty(3, C, dtype=torch.float32, device=dy.device)
        lsums = torch.empty(2 * C, dtype=torch.float64, device=dy.device)
        ls = _lib.LowerSums(_ptr(lz), _ptr(lbn), int(lrelu), float(lp), int(lseed) & 0xFFFFFFFFFFFFFFFF, int(row_hi),
                            _ptr(_lower_scratch(graph_bwd, batch, dy.device)), _ptr(lsums))
        gs = None
        if dy_sums is not None:
            gs = _given_sums(dy_sums)
        _lib.check(_lib.load().eg_gcn_layer_bwd_lower(
            graph_bwd._h, batch, _ptr(dy), _ptr(z), _ptr(agg), _ptr(weight), _ptr(gamma), _ptr(beta), _ptr(bn), int(relu),
            float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(residual), _ptr(_workspace(dy.device)), _ptr(dz), _ptr(dx), _ptr(dw),
            _ptr(small[0]), _ptr(small[1]), _ptr(small[2]), ct.byref(gs) if gs is not None else None, ct.byref(ls), _stream()),
            "eg_gcn_layer_bwd_lower")
        return dx, dw, small[0], small[1], small[2], lsums
    if dy_sums is not None and len(dy_sums) > 4 and dy_sums[4] is not None:
        # sums with the bilinear backward's later additions: the struct form without a lower layer
        dz = torch.empty_like(dy) if (need_dx or not need_dw) else None
        dx = torch.empty_like(dy) if need_dx else None
        dw = torch.empty(C, C, dtype=torch.float32, device=dy.device) if need_dw else None
        small = torch.empty(3, C, dtype=torch.float32, device=dy.device)
        gs = _given_sums(dy_sums)
        _lib.check(_lib.load().eg_gcn_layer_bwd_lower(
            graph_bwd._h, batch, _ptr(dy), _ptr(z), _ptr(agg), _ptr(weight), _ptr(gamma), _ptr(beta), _ptr(bn), int(relu),
            float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(residual), _ptr(_workspace(dy.device)), _ptr(dz), _ptr(dx), _ptr(dw),
            _ptr(small[0]), _ptr(small[1]), _ptr(small[2]), ct.byref(gs), None, _stream()), "eg_gcn_layer_bwd_lower")
        return dx, dw, small[0], small[1], small[2]
    dz = torch.empty_like(dy) if (need_dx or not need_dw) else None       # dW alone comes out of the fused apply pass
    dx = torch.empty_like(dy) if need_dx else None
    dw = torch.empty(C, C, dtype=torch.float32, device=dy.device) if need_dw else None
    small = torch.empty(3, C, dtype=torch.float32, device=dy.device)           # db, dgamma, dbeta
    common = (graph_bwd._h, batch, _ptr(dy), _ptr(z), _ptr(agg), _ptr(weight), _ptr(gamma), _ptr(beta), _ptr(bn), int(relu),
              float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(residual), _ptr(_workspace(dy.device)), _ptr(dz), _ptr(dx),
              _ptr(dw), _ptr(small[0]), _ptr(small[1]), _ptr(small[2]))
    if dy_sums is None:
        _lib.check(_lib.load().eg_gcn_layer_bwd(*common, _stream()), "eg_gcn_layer_bwd")
    else:
        sums, frames, row_lo, n_valid = dy_sums[:4]
        if sums.dtype != torch.float64 or sums.numel() != 2 * C or not sums.is_cuda or not sums.is_contiguous():
            raise RuntimeError("dy_sums must be a contiguous CUDA float64 tensor of 256 elements")
        _lib.check(_lib.load().eg_gcn_layer_bwd_presummed(*common, _ptr(sums), int(frames), int(row_lo), int(n_valid), _stream()),
                   "eg_gcn_layer_bwd_presummed")
    return dx, dw, small[0], small[1], small[2]


def _given_sums(dy_sums) -> "_lib.GivenSums":
    sums, frames, row_lo, n_valid = dy_sums[:4]
    taps = dy_sums[4] if len(dy_sums) > 4 else None
    if sums.dtype != torch.float64 or sums.numel() != 2 * C or not sums.is_cuda or not sums.is_contiguous():
        raise RuntimeError("dy_sums must be a contiguous CUDA float64 tensor of 256 elements")
    if taps is not None and (taps.dtype != torch.float32 or taps.numel() != int(frames) * 2 * C or not taps.is_cuda or not taps.is_contiguous()):
        raise RuntimeError("taps must be a contiguous CUDA float32 tensor [frames, 2, 128]")
    gs = _lib.GivenSums(_ptr(sums), int(frames), int(row_lo), int(n_valid), _ptr(taps))
    gs._keep = (sums, taps)
    return gs


CLS_GRADS_FLOATS = 19076
_cls_workspaces = {}


def _cls_workspace(device) -> torch.Tensor:
    key = (torch.device(device), torch.cuda.current_stream().cuda_stream)
    ws = _cls_workspaces.get(key)
    if ws is None:
        ws = torch.empty(int(_lib.load().eg_classifier_train_workspace_bytes()), dtype=torch.uint8, device=device)
        _cls_workspaces[key] = ws
    return ws


def _cls_params(P: dict) -> "_lib.ClsTrainParams":
    s = _lib.ClsTrainParams()
    for k in ("w1", "b1", "gamma1", "beta1", "w2", "b2", "gamma2", "beta2", "w3", "b3", "running_mean1", "running_var1",
              "running_mean2", "running_var2"):
        t = P.get(k)
        if t is not None and (not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous()):
            raise RuntimeError(f"classifier parameter {k} must be a contiguous CUDA float32 tensor")
        setattr(s, k, None if t is None else t.data_ptr())
    for k in ("eps1", "eps2", "p1", "p2"):
        setattr(s, k, float(P[k]))
    s.momentum1 = -1.0 if P.get("momentum1") is None else float(P["momentum1"])
    s.momentum2 = -1.0 if P.get("momentum2") is None else float(P["momentum2"])
    s.seed1, s.seed2 = int(P["seed1"]) & 0xFFFFFFFFFFFFFFFF, int(P["seed2"]) & 0xFFFFFFFFFFFFFFFF
    return s


def classifier_train_fwd(h, batch: int, n_per_frame: int, row_lo: int, n_valid: int, P: dict, sigmoid: bool):
    """-> (logits [batch*n_valid,4], z1, z2, bn [768])"""
    _check_rows(h, "h", batch * n_per_frame)
    rows = batch * n_valid
    dev = h.device
    z1 = torch.empty(rows, C, dtype=torch.float32, device=dev)
    z2 = torch.empty(rows, 64, dtype=torch.float32, device=dev)
    bn = torch.empty(4 * C + 4 * 64, dtype=torch.float32, device=dev)
    logits = torch.empty(rows, 4, dtype=torch.float32, device=dev)
    s = _cls_params(P)
    _lib.check(_lib.load().eg_classifier_train_fwd(_ptr(h), batch, n_per_frame, row_lo, n_valid, ct.byref(s),
                                                   _ptr(_cls_workspace(dev)), _ptr(z1), _ptr(z2), _ptr(bn), int(sigmoid),
                                                   _ptr(logits), _stream()), "eg_classifier_train_fwd")
    return logits, z1, z2, bn


def classifier_recompute_h_supported(batch: int, n_per_frame: int, n_valid: int) -> bool:
    """May the heads' backward rebuild the layer output it needs from z and the residual (classifier_bwd(recompute=), so that
    classifier_train_fwd_act(h_sparse=True) never writes it)?  Where the layer's sums come out of the heads' backward, arrays < 2 GB."""
    lim = (1 << 31) - (1 << 20)
    return (classifier_layer_sums_supported(batch, n_per_frame, n_valid) and batch * n_valid * C * 4 < lim and
            batch * n_per_frame * C * 4 < lim)


def classifier_train_fwd_act(z, layer_bn, residual, relu: bool, dropout_p: float, seed: int, batch: int, n_per_frame: int,
                             row_lo: int, n_valid: int, P: dict, sigmoid: bool, h_sparse: bool = False):
    """The heads' train forward with the last GNN layer's activation pass folded in (eg_classifier_train_fwd_act): z, layer_bn =
    what gcn_layer_train_fwd(..., want_out=False) returned, residual = that layer's input rows or None.
    -> (h [batch*n_per_frame,128], logits [batch*n_valid,4], z1, z2, bn [768]).  h_sparse: only the rows of h OUTSIDE the heads'
    filter are written (the rest of the tensor is uninitialised memory: the backward must take classifier_bwd(recompute=))."""
    _check_rows(z, "z", batch * n_per_frame)
    if residual is not None:
        _check_rows(residual, "residual", batch * n_per_frame)
    rows = batch * n_valid
    dev = z.device
    h = torch.empty_like(z)
    z1 = torch.empty(rows, C, dtype=torch.float32, device=dev)
    z2 = torch.empty(rows, 64, dtype=torch.float32, device=dev)
    bn = torch.empty(4 * C + 4 * 64, dtype=torch.float32, device=dev)
    logits = torch.empty(rows, 4, dtype=torch.float32, device=dev)
    s = _cls_params(P)
    _lib.check(_lib.load().eg_classifier_train_fwd_act(
        _ptr(z), _ptr(layer_bn), _ptr(residual), int(relu), float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(h), batch,
        n_per_frame, row_lo, n_valid, ct.byref(s), _ptr(_cls_workspace(dev)), _ptr(z1), _ptr(z2), _ptr(bn), int(sigmoid),
        _ptr(logits), int(bool(h_sparse)), _stream()), "eg_classifier_train_fwd_act")
    return h, logits, z1, z2, bn


def classifier_layer_sums_supported(batch: int, n_per_frame: int, n_valid: int) -> bool:
    """Does eg_classifier_bwd_sums cover this shape (the fused first-layers kernel: n_valid >= 64, < 2^32 elements)?"""
    return n_valid >= 64 and batch * n_per_frame * C < (1 << 32)


def classifier_bwd(dlogits, h, batch: int, n_per_frame: int, row_lo: int, n_valid: int, P: dict, z1, z2, bn, need_dh: bool,
                   layer=None, recompute=False):
    """-> (dh | None [batch*n_per_frame,128], grads [19076] packed as in include/echoglad_hip.h)
    layer = (z, bn, gamma, beta, relu, dropout_p, seed) of the GNN layer whose output h is: also returns that layer's
    BatchNorm-backward sums over the heads' rows, -> (dh, grads, sums [256] float64 | None) (eg_classifier_bwd_sums; None where
    the entry point does not cover the shape -- see classifier_layer_sums_supported -- and the plain backward ran instead).
    recompute = (residual rows | None,): h was written sparsely (classifier_train_fwd_act(h_sparse=True)); the kernel rebuilds the
    rows it needs from the layer's z and residual (needs ``layer``; classifier_recompute_h_supported)."""
    rows = batch * n_valid
    dev = h.device
    _check_logits(dlogits, "dlogits", rows)
    dh1 = torch.empty(rows, C, dtype=torch.float32, device=dev)
    dh = torch.empty_like(h) if need_dh else None
    grads = torch.empty(CLS_GRADS_FLOATS, dtype=torch.float32, device=dev)
    s = _cls_params(P)
    common = (_ptr(dlogits), _ptr(h), batch, n_per_frame, row_lo, n_valid, ct.byref(s), _ptr(z1), _ptr(z2), _ptr(bn),
              _ptr(_cls_workspace(dev)), _ptr(dh1), _ptr(dh), _ptr(grads))
    if layer is None:
        _lib.check(_lib.load().eg_classifier_bwd(*common, _stream()), "eg_classifier_bwd")
        return dh, grads
    lz, lbn, lgamma, lbeta, relu, p, seed = layer
    _check_rows(lz, "layer z", batch * n_per_frame)
    _check_vec(lgamma, "layer gamma", C)
    _check_vec(lbeta, "layer beta", C)
    sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
    rec = recompute is not False and recompute is not None
    res = recompute[0] if rec else None
    if res is not None:
        _check_rows(res, "layer residual", batch * n_per_frame)
    rc = _lib.load().eg_classifier_bwd_sums(*common, _ptr(lz), _ptr(lbn), _ptr(lgamma), _ptr(lbeta), int(relu), float(p),
                                            int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(sums), _ptr(res), int(rec), _stream())
    if rc == _lib.EG_ERR_UNSUPPORTED and rec:
        raise RuntimeError("classifier_bwd(recompute=): " + _lib.last_error())      # (h does not exist: there is no plain backward to fall back to)
    if rc == _lib.EG_ERR_UNSUPPORTED:           # nothing was launched (include/echoglad_hip.h): the plain backward, no sums
        _lib.check(_lib.load().eg_classifier_bwd(*common, _stream()), "eg_classifier_bwd")
        return dh, grads, None
    _lib.check(rc, "eg_classifier_bwd_sums")
    return dh, grads, sums


# ---------------------------------------------------------------------------
# coordinate-graph landmark update (models.py:438-453)
# ---------------------------------------------------------------------------
COORD_MLP_GRADS_FLOATS = 5042


def _frame_rows_ptr(h, batch: int, n_per_frame: int, row0: int):
    """(pointer to row `row0` of frame 0, floats between two frames) of a [batch * n_per_frame, 128] node array: where the 4
    coordinate rows of every frame live (eg_*_rows entry points)."""
    _check_rows(h, "h", batch * n_per_frame)
    if row0 < 0 or row0 + 4 > n_per_frame:
        raise RuntimeError("the 4 rows must lie inside a frame")
    return ct.c_void_p(h.data_ptr() + row0 * C * 4), n_per_frame * C


def coord_mlp_fwd(lm, coords, batch: int, P: dict, train: bool, frame: int, want_backward: bool, in_rows=None):
    """lm [4*batch,128], coords [4*batch,2] -> (new_coords [4*batch,2], saved = (z1, z2, bn, pre) | None).
    P: the _cls_params dictionary with the 136-32-16-2 head's tensors.
    in_rows = (h, n_per_frame, row0): the landmark rows are rows row0 .. row0 + 3 of every frame of h (no gathered copy); `lm`
    is then an OUTPUT -- the packed copy of those rows the backward needs -- or None."""
    rows = 4 * batch
    if lm is not None and (not lm.is_cuda or lm.dtype != torch.float32 or not lm.is_contiguous() or tuple(lm.shape) != (rows, C)):
        raise RuntimeError(f"lm must be a contiguous CUDA float32 [{rows}, {C}] tensor")
    if lm is None and in_rows is None:
        raise RuntimeError("lm or in_rows")
    _check_coords(coords, batch, 4)
    for k, shape in (("w1", (32, C + 8)), ("w2", (16, 32)), ("w3", (2, 16))):
        if tuple(P[k].shape) != shape:
            raise RuntimeError(f"coordinate MLP {k} must be {shape}, got {tuple(P[k].shape)}")
    dev = coords.device
    z1 = torch.empty(rows, 32, dtype=torch.float32, device=dev)
    z2 = torch.empty(rows, 16, dtype=torch.float32, device=dev)
    bn = torch.empty(96, dtype=torch.float32, device=dev)
    pre = torch.empty(rows, 2, dtype=torch.float32, device=dev) if want_backward else None
    new = torch.empty(rows, 2, dtype=torch.float32, device=dev)
    s = _cls_params(P)
    if in_rows is not None:
        ptr, stride = _frame_rows_ptr(in_rows[0], batch, in_rows[1], in_rows[2])
        _lib.check(_lib.load().eg_coord_mlp_fwd_rows(ptr, stride, _ptr(lm), _ptr(coords), batch, ct.byref(s), int(train), frame,
                                                     _ptr(z1), _ptr(z2), _ptr(bn), _ptr(pre), _ptr(new), _stream()), "eg_coord_mlp_fwd_rows")
    else:
        _lib.check(_lib.load().eg_coord_mlp_fwd(_ptr(lm), _ptr(coords), batch, ct.byref(s), int(train), frame, _ptr(z1), _ptr(z2),
                                                _ptr(bn), _ptr(pre), _ptr(new), _stream()), "eg_coord_mlp_fwd")
    return new, ((z1, z2, bn, pre) if want_backward else None)


def coord_mlp_bwd(dnew, lm, coords, batch: int, P: dict, frame: int, saved, need_dlm: bool, need_dcoords: bool, out_rows=None,
                  accumulate: bool = False):
    """-> (dlm | None, dcoords | None, grads [5042] packed as in include/echoglad_hip.h)
    out_rows = (dh, n_per_frame, row0): dlm is written (accumulate: added) into rows row0 .. row0 + 3 of every frame of dh
    instead of being returned."""
    rows = 4 * batch
    z1, z2, bn, pre = saved
    dev = lm.device
    if not dnew.is_cuda or dnew.dtype != torch.float32 or not dnew.is_contiguous() or dnew.numel() != rows * 2:
        raise RuntimeError(f"dnew must be contiguous CUDA float32 with {rows * 2} elements")
    scratch = torch.empty(rows, 56, dtype=torch.float32, device=dev)
    dlm = torch.empty(rows, C, dtype=torch.float32, device=dev) if (need_dlm and out_rows is None) else None
    dcoords = torch.empty(rows, 2, dtype=torch.float32, device=dev) if need_dcoords else None
    grads = torch.empty(COORD_MLP_GRADS_FLOATS, dtype=torch.float32, device=dev)
    s = _cls_params(P)
    if out_rows is not None:
        ptr, stride = _frame_rows_ptr(out_rows[0], batch, out_rows[1], out_rows[2])
        _lib.check(_lib.load().eg_coord_mlp_bwd_rows(_ptr(dnew), _ptr(lm), _ptr(coords), batch, ct.byref(s), frame, _ptr(z1), _ptr(z2),
                                                     _ptr(bn), _ptr(pre), _ptr(scratch), ptr, stride, int(accumulate), _ptr(dcoords),
                                                     _ptr(grads), _stream()), "eg_coord_mlp_bwd_rows")
        return None, dcoords, grads
    _lib.check(_lib.load().eg_coord_mlp_bwd(_ptr(dnew), _ptr(lm), _ptr(coords), batch, ct.byref(s), frame, _ptr(z1), _ptr(z2),
                                            _ptr(bn), _ptr(pre), _ptr(scratch), _ptr(dlm), _ptr(dcoords), _ptr(grads),
                                            _stream()), "eg_coord_mlp_bwd")
    return dlm, dcoords, grads


def coord_update_fwd(h, coords, batch: int, n_per_frame: int, coord_base: int, main_base: int, P: dict, train: bool, frame: int,
                     want_backward: bool, resample: bool = True):
    """The coordinate update of one GNN layer on the node array IN PLACE (eg_coord_update_fwd: models.py:438-473): the landmark MLP on
    the 4 coordinate rows of every frame of h, then (resample) those rows overwritten with the main grid sampled at the new positions.
    One launch up to batch 16.  -> ((new_coords, new_coords_again) [4*batch,2] each: the same coordinates in two tensors -- one to hand
    out, one to keep for the backward, without a copy launch --, lm = packed copy of the rows the MLP read, saved = (z1, z2, bn, pre) | None)"""
    rows = 4 * batch
    _check_rows(h, "h", batch * n_per_frame)
    _check_coords(coords, batch, 4)
    for k, shape in (("w1", (32, C + 8)), ("w2", (16, 32)), ("w3", (2, 16))):
        if tuple(P[k].shape) != shape:
            raise RuntimeError(f"coordinate MLP {k} must be {shape}, got {tuple(P[k].shape)}")
    dev = h.device
    lm = torch.empty(rows, C, dtype=torch.float32, device=dev)
    z1 = torch.empty(rows, 32, dtype=torch.float32, device=dev)
    z2 = torch.empty(rows, 16, dtype=torch.float32, device=dev)
    bn = torch.empty(96, dtype=torch.float32, device=dev)
    pre = torch.empty(rows, 2, dtype=torch.float32, device=dev) if want_backward else None
    new = (torch.empty(rows, 2, dtype=torch.float32, device=dev), torch.empty(rows, 2, dtype=torch.float32, device=dev))
    s = _cls_params(P)
    _lib.check(_lib.load().eg_coord_update_fwd(_ptr(h), n_per_frame, coord_base, main_base, _ptr(coords), batch, ct.byref(s), int(train),
                                               frame, int(bool(resample)), _ptr(lm), _ptr(z1), _ptr(z2), _ptr(bn), _ptr(pre), _ptr(new[0]),
                                               _ptr(new[1]), _stream()), "eg_coord_update_fwd")
    return new, lm, ((z1, z2, bn, pre) if want_backward else None)


def coord_update_bwd(dx, dnew, h, new, lm, coords, batch: int, n_per_frame: int, coord_base: int, main_base: int, P: dict, frame: int,
                     saved, need_dcoords: bool, lower=None):
    """Backward of coord_update_fwd(resample=True) on the gradient array dx IN PLACE (eg_coord_update_bwd): dx is the gradient w.r.t. the
    tensor after the update and leaves as the gradient w.r.t. the tensor before it.  dnew: d new_coords from downstream or None;
    h: the tensor the samples were taken from (after the update); lower as in bilinear4_bwd.
    -> (dcoords | None, grads [5042], taps [batch,2,128] | None).  One launch up to batch 16."""
    rows = 4 * batch
    z1, z2, bn, pre = saved
    _check_rows(dx, "dx", batch * n_per_frame)
    _check_rows(h, "h", batch * n_per_frame)
    if dnew is not None and (not dnew.is_cuda or dnew.dtype != torch.float32 or not dnew.is_contiguous() or dnew.numel() != rows * 2):
        raise RuntimeError(f"dnew must be contiguous CUDA float32 with {rows * 2} elements")
    dev = dx.device
    scratch = torch.empty(rows, 56, dtype=torch.float32, device=dev)
    dbil = torch.empty(rows, 2, dtype=torch.float32, device=dev)
    dcoords = torch.empty(rows, 2, dtype=torch.float32, device=dev) if need_dcoords else None
    grads = torch.empty(COORD_MLP_GRADS_FLOATS, dtype=torch.float32, device=dev)
    taps, ls = None, None
    if lower is not None:
        lz, lbn, lrelu, lp, lseed = lower[:5]
        _check_rows(lz, "lower z", batch * n_per_frame)
        taps = torch.empty(batch, 2, C, dtype=torch.float32, device=dev)
        ls = ct.byref(_lib.LowerSums(_ptr(lz), _ptr(lbn), int(lrelu), float(lp), int(lseed) & 0xFFFFFFFFFFFFFFFF, 0, None, None))
    s = _cls_params(P)
    _lib.check(_lib.load().eg_coord_update_bwd(_ptr(dx), n_per_frame, coord_base, main_base, _ptr(h), _ptr(new), _ptr(dnew), _ptr(lm),
                                               _ptr(coords), batch, ct.byref(s), frame, _ptr(z1), _ptr(z2), _ptr(bn), _ptr(pre),
                                               _ptr(scratch), _ptr(dbil), ls, _ptr(taps), _ptr(dcoords), _ptr(grads), _stream()),
               "eg_coord_update_bwd")
    return dcoords, grads, taps


# ---------------------------------------------------------------------------
# coordinate-graph resampling
# ---------------------------------------------------------------------------
def _check_coords(coords, batch, points):
    if not coords.is_cuda or coords.dtype != torch.float32 or not coords.is_contiguous() or \
            coords.numel() != batch * points * 2:
        raise RuntimeError(f"coords must be contiguous CUDA float32 with {batch * points * 2} elements")


def bilinear4_fwd(h, coords, batch, n_per_frame, main_base, frame, points=4, out_rows=None) -> torch.Tensor:
    """out_rows = (dst, n_per_frame, row0): the samples are written into rows row0 .. of every frame of dst (None is returned)."""
    _check_rows(h, "h", batch * n_per_frame)
    _check_coords(coords, batch, points)
    if out_rows is not None:
        ptr, stride = _frame_rows_ptr(out_rows[0], batch, out_rows[1], out_rows[2])
        _lib.check(_lib.load().eg_bilinear4_fwd_rows(_ptr(h), _ptr(coords), batch, points, n_per_frame, main_base, frame, ptr, stride,
                                                     _stream()), "eg_bilinear4_fwd_rows")
        return None
    out = torch.empty(batch * points, C, dtype=torch.float32, device=h.device)
    _lib.check(_lib.load().eg_bilinear4_fwd(_ptr(h), _ptr(coords), batch, points, n_per_frame, main_base, frame,
                                            _ptr(out), _stream()), "eg_bilinear4_fwd")
    return out


def bilinear4_bwd(dout, h, coords, batch, n_per_frame, main_base, frame, dh=None, want_dcoords=True, points=4, dout_rows=None,
                  lower=None):
    """dout_rows = (src, n_per_frame, row0): the samples' gradient is read from rows row0 .. of every frame of src (dout is None).
    lower = (z, bn, relu, dropout_p, seed, ...) of the layer whose dy ``dh`` is, when that layer's BatchNorm-backward sums were taken
    before this call (gcn_layer_bwd(..., lower=)): -> (dcoords, taps [batch, 2, 128]), the sums of what is added here."""
    _check_rows(h, "h", batch * n_per_frame)
    _check_coords(coords, batch, points)
    dcoords = torch.empty(batch * points, 2, dtype=torch.float32, device=h.device) if want_dcoords else None
    if lower is not None:
        if dout_rows is None or dh is None:
            raise RuntimeError("tap sums go with the in-place form (dout_rows, dh)")
        lz, lbn, lrelu, lp, lseed = lower[:5]
        _check_rows(lz, "lower z", batch * n_per_frame)
        taps = torch.empty(batch, 2, C, dtype=torch.float32, device=h.device)
        ls = _lib.LowerSums(_ptr(lz), _ptr(lbn), int(lrelu), float(lp), int(lseed) & 0xFFFFFFFFFFFFFFFF, 0, None, None)
        ptr, stride = _frame_rows_ptr(dout_rows[0], batch, dout_rows[1], dout_rows[2])
        _lib.check(_lib.load().eg_bilinear4_bwd_rows_sums(ptr, stride, _ptr(h), _ptr(coords), batch, points, n_per_frame, main_base, frame,
                                                          _ptr(dh), _ptr(dcoords), ct.byref(ls), _ptr(taps), _stream()),
                   "eg_bilinear4_bwd_rows_sums")
        return dcoords, taps
    if dout_rows is not None:
        ptr, stride = _frame_rows_ptr(dout_rows[0], batch, dout_rows[1], dout_rows[2])
        _lib.check(_lib.load().eg_bilinear4_bwd_rows(ptr, stride, _ptr(h), _ptr(coords), batch, points, n_per_frame, main_base, frame,
                                                     _ptr(dh), _ptr(dcoords), _stream()), "eg_bilinear4_bwd_rows")
        return dcoords
    dout = dout.contiguous()
    _lib.check(_lib.load().eg_bilinear4_bwd(_ptr(dout), _ptr(h), _ptr(coords), batch, points, n_per_frame, main_base,
                                            frame, _ptr(dh), _ptr(dcoords), _stream()), "eg_bilinear4_bwd")
    return dcoords


class _Bilinear4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, coords, batch, n_per_frame, main_base, frame):
        coords = coords.contiguous()
        ctx.save_for_backward(h, coords)
        ctx.dims = (batch, n_per_frame, main_base, frame)
        return bilinear4_fwd(h, coords, batch, n_per_frame, main_base, frame)

    @staticmethod
    def backward(ctx, dout):
        h, coords = ctx.saved_tensors
        batch, n_per_frame, main_base, frame = ctx.dims
        dh = torch.zeros_like(h) if ctx.needs_input_grad[0] else None
        dcoords = bilinear4_bwd(dout, h, coords, batch, n_per_frame, main_base, frame, dh=dh,
                                want_dcoords=ctx.needs_input_grad[1])
        if dcoords is not None:
            dcoords = dcoords.view_as(coords)
        return dh, dcoords, None, None, None, None


def bilinear4(h, coords, batch, n_per_frame, main_base, frame) -> torch.Tensor:
    """[batch*4, 128] features at the landmark coordinates (differentiable wrt h and coords)."""
    c = coords.reshape(batch * 4, 2)
    if torch.is_grad_enabled() and (h.requires_grad or c.requires_grad):
        return _Bilinear4Fn.apply(h, c, batch, n_per_frame, main_base, frame)
    return bilinear4_fwd(h, c.contiguous(), batch, n_per_frame, main_base, frame)


def scatter_coord_rows(h, new_feats, batch, n_per_frame, coord_base) -> torch.Tensor:
    """h[type==1 rows] = new_feats (models.py:473).  The coordinate rows are the last 4 rows of every frame, so
    this is a strided slice assignment; under autograd a copy keeps the saved forward values intact."""
    need_grad = torch.is_grad_enabled() and (h.requires_grad or new_feats.requires_grad)
    tgt = h.clone() if need_grad else h
    tgt.view(batch, n_per_frame, C)[:, coord_base:coord_base + 4, :] = new_feats.view(batch, 4, C)
    return tgt


# ---------------------------------------------------------------------------
# losses on the logits / landmark decode (heatmap.hip)
# ---------------------------------------------------------------------------
def _check_logits(t: torch.Tensor, name: str, rows: Optional[int] = None) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (ROCm) tensor: the HIP path has no CPU fallback")
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != 4 or not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous float32 [rows, 4], got {tuple(t.shape)} {t.dtype}")
    if rows is not None and t.shape[0] != rows:
        raise RuntimeError(f"{name} has {t.shape[0]} rows, expected {rows}")


def _level_arrays(levels):
    n = len(levels)
    if not 1 <= n <= 16:
        raise RuntimeError("1..16 levels supported")
    start = (ct.c_int * n)(*[int(s) for s, _ in levels])
    side = (ct.c_int * n)(*[int(p) for _, p in levels])
    return start, side, n


_hm_workspaces = {}


def _hm_workspace(device, nbytes: int) -> torch.Tensor:
    key = (torch.device(device), torch.cuda.current_stream().cuda_stream)
    ws = _hm_workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
        _hm_workspaces[key] = ws
    return ws


def heatmap_expect_fwd(logits: torch.Tensor, batch: int, levels, labels: Optional[torch.Tensor] = None,
                       valid: Optional[torch.Tensor] = None, want_argmax: bool = False):
    """Per (frame, level, channel): softmax-expected (h, w), (max, sum-exp), first arg max, label (h, w), mean(valid).

    levels: [(first row inside a frame's rows, side)]; logits [batch * n_rows, 4].
    Returns dict(expect [B,L,4,2], stats [B,L,4,2], argmax [B,L,4] | None, gt [B,L,4,2] | None, vmean [B,L,4] | None)."""
    if logits.shape[0] % batch:
        raise RuntimeError("logit rows are not a multiple of the batch size")
    n_rows = logits.shape[0] // batch
    _check_logits(logits, "logits")
    for name, t in (("labels", labels), ("valid", valid)):
        if t is not None:
            _check_logits(t, name, logits.shape[0])
    start, side, n = _level_arrays(levels)
    lib = _lib.load()
    ws = _hm_workspace(logits.device, int(lib.eg_heatmap_workspace_bytes(batch, side, n)))
    dev = logits.device
    expect = torch.empty(batch, n, 4, 2, dtype=torch.float32, device=dev)
    stats = torch.empty(batch, n, 4, 2, dtype=torch.float32, device=dev)
    argmax = torch.empty(batch, n, 4, dtype=torch.int64, device=dev) if want_argmax else None
    gt = torch.empty(batch, n, 4, 2, dtype=torch.float32, device=dev) if labels is not None else None
    vmean = torch.empty(batch, n, 4, dtype=torch.float32, device=dev) if valid is not None else None
    _lib.check(lib.eg_heatmap_expect_fwd(_ptr(logits), _ptr(labels), _ptr(valid), batch, n_rows, start, side, n, _ptr(ws),
                                         _ptr(expect), _ptr(stats), _ptr(argmax), _ptr(gt), _ptr(vmean), _stream()),
               "eg_heatmap_expect_fwd")
    return {"expect": expect, "stats": stats, "argmax": argmax, "gt": gt, "vmean": vmean}


def heatmap_expect_bwd(logits, expect, stats, d_expect, batch: int, levels) -> torch.Tensor:
    n_rows = logits.shape[0] // batch
    start, side, n = _level_arrays(levels)
    d_logits = torch.empty_like(logits)
    _lib.check(_lib.load().eg_heatmap_expect_bwd(_ptr(logits), _ptr(expect), _ptr(stats), _ptr(d_expect.contiguous()), batch,
                                                 n_rows, start, side, n, _ptr(d_logits), _stream()), "eg_heatmap_expect_bwd")
    return d_logits


class _HeatmapExpectFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, batch, levels, labels, valid):
        r = heatmap_expect_fwd(logits, batch, levels, labels, valid)
        ctx.save_for_backward(logits, r["expect"], r["stats"])
        ctx.batch, ctx.levels = batch, levels
        gt = r["gt"] if r["gt"] is not None else logits.new_zeros(0)
        vm = r["vmean"] if r["vmean"] is not None else logits.new_zeros(0)
        ctx.mark_non_differentiable(gt, vm)
        return r["expect"], gt, vm

    @staticmethod
    def backward(ctx, d_expect, _dgt, _dvm):
        logits, expect, stats = ctx.saved_tensors
        return heatmap_expect_bwd(logits, expect, stats, d_expect, ctx.batch, ctx.levels), None, None, None, None


def heatmap_expect(logits, batch: int, levels, labels=None, valid=None):
    """(expect, gt, vmean) with autograd through `expect` (d/d logits)."""
    return _HeatmapExpectFn.apply(logits, batch, tuple(levels), labels, valid)


def elm_reduce(expect, gt, vmean, inv_side, weight: float):
    """ExpectedLandmarkMSE's combination of the per-(frame, level, channel) expectations and its gradient in one launch
    (eg_elm_reduce) -> (loss [1], d loss / d expect [B, L, 4, 2])."""
    B, L = int(expect.shape[0]), int(expect.shape[1])
    for name, t, shape in (("expect", expect, (B, L, 4, 2)), ("gt", gt, (B, L, 4, 2)), ("vmean", vmean, (B, L, 4)), ("inv_side", inv_side, None)):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or (shape is not None and tuple(t.shape) != shape):
            raise RuntimeError(f"{name} must be a contiguous CUDA float32 tensor" + (f" of shape {shape}" if shape else ""))
    if inv_side.numel() != L:
        raise RuntimeError("inv_side must hold one value per level")
    loss = torch.empty(1, dtype=torch.float32, device=expect.device)
    d = torch.empty_like(expect)
    _lib.check(_lib.load().eg_elm_reduce(_ptr(expect), _ptr(gt), _ptr(vmean), _ptr(inv_side), B, L, ct.c_float(weight), _ptr(loss),
                                         _ptr(d), _stream()), "eg_elm_reduce")
    return loss, d


def bce_logits_fwd(logits, labels, valid, ones_weight: float) -> torch.Tensor:
    """[sum(w * bce * valid), sum(valid), ratio] as a float32 device tensor (no host sync)."""
    for name, t in (("logits", logits), ("labels", labels)):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"{name} must be a contiguous CUDA float32 tensor")
    if labels.numel() != logits.numel() or (valid is not None and valid.numel() != logits.numel()):
        raise RuntimeError("logits, labels and valid must have the same number of elements")
    # the kernel reads 16 bytes per lane: a contiguous view at an odd element offset (flat[1:], a slice of a packed buffer) is copied
    logits, labels, valid = (t if t is None or t.data_ptr() % 16 == 0 else t.clone() for t in (logits, labels, valid))
    lib = _lib.load()
    one = (ct.c_int * 1)(1)
    ws = _hm_workspace(logits.device, int(lib.eg_heatmap_workspace_bytes(1, one, 1)))
    out = torch.empty(3, dtype=torch.float32, device=logits.device)
    _lib.check(lib.eg_bce_logits_fwd(_ptr(logits), _ptr(labels), _ptr(valid), logits.numel(), ct.c_float(ones_weight),
                                     _ptr(ws), _ptr(out), _stream()), "eg_bce_logits_fwd")
    return out


class _BCELogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, valid, ones_weight):
        out = bce_logits_fwd(logits, labels, valid, ones_weight)
        ctx.save_for_backward(logits, labels, valid if valid is not None else logits.new_zeros(0), out)
        ctx.ones_weight, ctx.has_valid = ones_weight, valid is not None
        return out[2]

    @staticmethod
    def backward(ctx, g):
        logits, labels, valid, out = ctx.saved_tensors
        scale = (g / out[1]).reshape(1).to(torch.float32).contiguous()
        dx = torch.empty_like(logits)
        _lib.check(_lib.load().eg_bce_logits_bwd(_ptr(logits), _ptr(labels), _ptr(valid) if ctx.has_valid else None,
                                                 logits.numel(), ct.c_float(ctx.ones_weight), _ptr(scale), _ptr(dx), _stream()),
                   "eg_bce_logits_bwd")
        return dx, None, None, None


def bce_logits(logits, labels, valid=None, ones_weight: float = 1.0) -> torch.Tensor:
    """sum(w * bce_with_logits(x, y) * valid) / sum(valid), w = ones_weight where y == 1 (autograd wrt logits)."""
    return _BCELogitsFn.apply(logits, labels, valid, float(ones_weight))


class _CriteriaFn(torch.autograd.Function):
    """WeightedBCEWithLogitsLoss + ExpectedLandmarkMSE (+ MSE on the landmark coordinates) of one training step as ONE autograd
    node over eg_criteria_fwd / eg_criteria_bwd: (logits [B*n,4], coord_pred [R,2] | None) -> (total, bce, elm, coord | None),
    every output a 0-d tensor that can be backpropagated on its own or summed (engine.py:582-600, :271)."""

    @staticmethod
    def forward(ctx, logits, coord_pred, labels, valid, coord_y, batch, levels, inv_side, ones_weight, w_bce, w_elm, w_coord):
        dev = logits.device
        n_rows = logits.shape[0] // batch
        start, side, n = _level_arrays(levels)
        lib = _lib.load()
        ws = _hm_workspace(dev, int(lib.eg_criteria_workspace_bytes(batch, side, n)))
        expect = torch.empty(batch, n, 4, 2, dtype=torch.float32, device=dev)
        stats = torch.empty_like(expect)
        d_expect = torch.empty_like(expect)
        has_coord = coord_pred is not None
        cp = coord_pred.contiguous() if has_coord else None
        cy = coord_y.to(torch.float32).contiguous() if has_coord else None
        d_coord = torch.empty_like(cp) if has_coord else None
        bce_scale = torch.empty(1, dtype=torch.float32, device=dev)
        total, vb, ve = (torch.empty((), dtype=torch.float32, device=dev) for _ in range(3))
        vc = torch.empty((), dtype=torch.float32, device=dev) if has_coord else None
        _lib.check(lib.eg_criteria_fwd(_ptr(logits), _ptr(labels), _ptr(valid), batch, n_rows, start, side, n, _ptr(inv_side),
                                       ct.c_float(ones_weight), ct.c_float(w_bce), ct.c_float(w_elm), _ptr(cp), _ptr(cy),
                                       cp.numel() if has_coord else 0, ct.c_float(w_coord), _ptr(ws), _ptr(expect), _ptr(stats),
                                       _ptr(d_expect), _ptr(d_coord), _ptr(bce_scale), _ptr(total), _ptr(vb), _ptr(ve), _ptr(vc),
                                       _stream()), "eg_criteria_fwd")
        ctx.meta = (batch, levels, ones_weight, has_coord)
        ctx.save_for_backward(logits, labels, valid, expect, stats, d_expect, bce_scale, d_coord if has_coord else logits.new_zeros(0))
        ctx.set_materialize_grads(False)
        return total, vb, ve, vc

    @staticmethod
    def backward(ctx, g_total, g_bce, g_elm, g_coord):
        logits, labels, valid, expect, stats, d_expect, bce_scale, d_coord = ctx.saved_tensors
        batch, levels, ones_weight, has_coord = ctx.meta
        start, side, n = _level_arrays(levels)
        gs = [None if g is None else g.to(torch.float32).reshape(1).contiguous() for g in (g_total, g_bce, g_elm, g_coord)]
        d_logits = torch.empty_like(logits)
        want_coord = has_coord and ctx.needs_input_grad[1]
        d_coord_out = torch.empty_like(d_coord) if want_coord else None
        _lib.check(_lib.load().eg_criteria_bwd(_ptr(logits), _ptr(labels), _ptr(valid), batch, logits.shape[0] // batch, start, side, n,
                                               ct.c_float(ones_weight), _ptr(expect), _ptr(stats), _ptr(d_expect), _ptr(bce_scale),
                                               _ptr(d_coord) if want_coord else None, d_coord.numel() if want_coord else 0,
                                               _ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), _ptr(gs[3]), _ptr(d_logits), _ptr(d_coord_out),
                                               _stream()), "eg_criteria_bwd")
        return (d_logits, d_coord_out) + (None,) * 10


def landmark_criteria(logits, labels, valid, batch: int, levels, inv_side, ones_weight: float, w_bce: float, w_elm: float,
                      coord_pred=None, coord_y=None, w_coord: float = 1.0):
    """-> (total, bce, elm, coord | None): the step's criteria as one autograd node (5 launches forward + backward)."""
    for name, t in (("logits", logits), ("labels", labels), ("valid", valid)):
        _check_logits(t, name, logits.shape[0])
    if any(t.data_ptr() % 16 for t in (logits, labels, valid)):
        logits, labels, valid = (t if t.data_ptr() % 16 == 0 else t.clone() for t in (logits, labels, valid))
    return _CriteriaFn.apply(logits, coord_pred, labels, valid, coord_y, int(batch), levels, inv_side, float(ones_weight), float(w_bce),
                             float(w_elm), float(w_coord))


# ---------------------------------------------------------------------------
# average-pool pyramid (pool.hip) + node-feature packing (pack.hip)
# ---------------------------------------------------------------------------
def pyramid_supported(x: torch.Tensor, sides) -> bool:
    """eg_avg_pool_pyramid_* cover square float32 CUDA planes up to 512 x 512 and strictly ascending sides <= the frame."""
    sides = list(sides)
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[2] <= 512 and
            1 <= len(sides) <= 16 and all(1 <= a < b for a, b in zip(sides, sides[1:])) and sides[-1] <= x.shape[2] and sides[0] >= 1)


def _pool_fwd(x: torch.Tensor, sides):
    B, Cn, Fr, _ = x.shape
    maps = [torch.empty(B, Cn, p, p, dtype=torch.float32, device=x.device) for p in sides]
    n = len(sides)
    _lib.check(_lib.load().eg_avg_pool_pyramid_fwd(_ptr(x), B * Cn, Fr, (ct.c_int * n)(*sides), n,
                                                   (ct.c_void_p * n)(*[m.data_ptr() for m in maps]), _stream()), "eg_avg_pool_pyramid_fwd")
    return maps


def _pool_bwd(grads, frame_grad, sides, shape, device):
    B, Cn, Fr, _ = shape
    n = len(sides)
    dx = torch.empty(shape, dtype=torch.float32, device=device)
    ptrs = (ct.c_void_p * n)(*[(g.data_ptr() if g is not None else None) for g in grads])
    _lib.check(_lib.load().eg_avg_pool_pyramid_bwd(ptrs, _ptr(frame_grad), B * Cn, Fr, (ct.c_int * n)(*sides), n, _ptr(dx), _stream()),
               "eg_avg_pool_pyramid_bwd")
    return dx


class _AvgPoolPyramidFn(torch.autograd.Function):
    """[B, C, F, F] -> tuple of F.adaptive_avg_pool2d(x, p) for every side p, one launch each way."""

    @staticmethod
    def forward(ctx, x, *sides):
        x = x.contiguous()
        ctx.meta = (tuple(sides), tuple(x.shape), x.device)
        return tuple(_pool_fwd(x, list(sides)))

    @staticmethod
    def backward(ctx, *grads):
        sides, shape, device = ctx.meta
        gs = [g.contiguous() if g is not None else None for g in grads]
        return (_pool_bwd(gs, None, list(sides), shape, device),) + (None,) * len(sides)


def avg_pool_pyramid(x: torch.Tensor, sides):
    """[F.adaptive_avg_pool2d(x, (p, p)) for p in sides] (strictly ascending) in one launch; differentiable w.r.t. x."""
    if not pyramid_supported(x, sides):
        raise RuntimeError("avg_pool_pyramid: square float32 CUDA planes up to 512 x 512, strictly ascending sides <= the frame")
    return list(_AvgPoolPyramidFn.apply(x, *[int(p) for p in sides]))


class _PyramidPackFn(torch.autograd.Function):
    """create_node_pixels of the base model (models.py:511-523) as ONE autograd node: pooled pyramid of the frame embedding +
    the frame itself -> node-major rows.  Forward: eg_avg_pool_pyramid_fwd + eg_pack_levels; backward: eg_unpack_levels +
    eg_avg_pool_pyramid_bwd (the frame's own rows are added there: no second [B,128,F,F] gradient for autograd to sum)."""

    @staticmethod
    def forward(ctx, x, batch, n_rows, row_offset, out, *sides):
        x = x.contiguous()
        sides = list(sides)
        maps = _pool_fwd(x, sides) + [x]
        used = sum(int(m.shape[2]) ** 2 for m in maps)
        if out is not None:
            nodes = _pack_out(out, [], batch, n_rows, row_offset, used)
        else:
            alloc = torch.empty if (row_offset == 0 and used == n_rows) else torch.zeros
            nodes = alloc(batch * n_rows, C, dtype=torch.float32, device=x.device)
        _pack_call("eg_pack_levels", maps, nodes, batch, n_rows, row_offset)
        ctx.meta = (batch, n_rows, row_offset, sides, tuple(x.shape))
        if out is not None:
            ctx.mark_dirty(out)
        return nodes

    @staticmethod
    def backward(ctx, d_nodes):
        batch, n_rows, row_offset, sides, shape = ctx.meta
        grads = [torch.empty(shape[0], shape[1], p, p, dtype=torch.float32, device=d_nodes.device) for p in sides]
        g_frame = torch.empty(shape, dtype=torch.float32, device=d_nodes.device)
        _pack_call("eg_unpack_levels", grads + [g_frame], d_nodes.contiguous(), batch, n_rows, row_offset)
        return (_pool_bwd(grads, g_frame, sides, shape, d_nodes.device),) + (None,) * (4 + len(sides))


def pyramid_pack(x: torch.Tensor, sides, batch: int, n_rows: int, row_offset: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Pooled pyramid (sides ascending) of x [batch,128,F,F] + x itself, packed node-major [batch * n_rows, 128] like
    ``pack_levels([adaptive_avg_pool2d(x, p) ...] + [x])``.  Differentiable w.r.t. x (not with ``out=``: written in place)."""
    if out is not None and torch.is_grad_enabled() and x.requires_grad:
        raise RuntimeError("pyramid_pack into out= is not differentiable: call it under torch.no_grad() or without out=")
    return _PyramidPackFn.apply(x, int(batch), int(n_rows), int(row_offset), out, *[int(p) for p in sides])


# ---------------------------------------------------------------------------
# node-feature packing (pack.hip)
# ---------------------------------------------------------------------------
def _pack_call(fn_name, maps, nodes, batch, n_rows, row_offset):
    n = len(maps)
    if not 1 <= n <= 16:
        raise RuntimeError("1..16 level maps supported")
    sides = []
    for m in maps:
        if not m.is_cuda or m.dtype != torch.float32 or m.dim() != 4 or m.shape[0] != batch or m.shape[1] != C or \
                m.shape[2] != m.shape[3] or not m.is_contiguous():
            raise RuntimeError(f"level maps must be contiguous CUDA float32 [batch, {C}, side, side], got {tuple(m.shape)}")
        sides.append(int(m.shape[2]))
    ptrs = (ct.c_void_p * n)(*[m.data_ptr() for m in maps])
    side = (ct.c_int * n)(*sides)
    lib = _lib.load()
    if fn_name == "eg_pack_levels":
        rc = lib.eg_pack_levels(ptrs, side, n, batch, n_rows, row_offset, _ptr(nodes), _stream())
    else:
        rc = lib.eg_unpack_levels(_ptr(nodes), ptrs, side, n, batch, n_rows, row_offset, _stream())
    _lib.check(rc, fn_name)


class _PackLevelsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, batch, n_rows, row_offset, *maps):
        maps = [m.contiguous() for m in maps]
        used = sum(int(m.shape[2]) ** 2 for m in maps)
        alloc = torch.empty if (row_offset == 0 and used == n_rows) else torch.zeros
        nodes = alloc(batch * n_rows, C, dtype=torch.float32, device=maps[0].device)
        _pack_call("eg_pack_levels", maps, nodes, batch, n_rows, row_offset)
        ctx.meta = (batch, n_rows, row_offset, [tuple(m.shape) for m in maps])
        return nodes

    @staticmethod
    def backward(ctx, d_nodes):
        batch, n_rows, row_offset, shapes = ctx.meta
        grads = [torch.empty(s, dtype=torch.float32, device=d_nodes.device) for s in shapes]
        _pack_call("eg_unpack_levels", grads, d_nodes.contiguous(), batch, n_rows, row_offset)
        return (None, None, None, *grads)


def _pack_out(out: torch.Tensor, inputs, batch: int, n_rows: int, row_offset: int, used: int) -> torch.Tensor:
    """``out=`` of the packing calls: the caller's [batch * n_rows, 128] buffer is written in place (a static node-feature
    buffer that a captured HIP graph reads: nn.HierarchicalPatchModel.enable_hip_graph).  No autograd through it."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in inputs):
        raise RuntimeError("pack into out= is not differentiable: call it under torch.no_grad() or without out=")
    _check_rows(out, "out", batch * n_rows)
    if row_offset > 0 or used < n_rows:              # rows no level covers (connection / coordinate nodes) read as zero, as without out=
        v = out.view(batch, n_rows, C)
        if row_offset > 0:
            v[:, :row_offset].zero_()
        if row_offset + used < n_rows:
            v[:, row_offset + used:].zero_()
    return out


def pack_levels(maps, batch: int, n_rows: int, row_offset: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """NCHW level maps [batch,128,p,p] (coarse to fine) -> node-major [batch * n_rows, 128]; level l lands at rows
    row_offset + sum_{k<l} p_k^2 of every frame.  Differentiable w.r.t. the maps (not with ``out=``: written in place)."""
    if out is not None:
        maps = [m.contiguous() for m in maps]
        _pack_out(out, maps, int(batch), int(n_rows), int(row_offset), sum(int(m.shape[2]) ** 2 for m in maps))
        _pack_call("eg_pack_levels", maps, out, int(batch), int(n_rows), int(row_offset))
        return out
    return _PackLevelsFn.apply(int(batch), int(n_rows), int(row_offset), *maps)


def _conv_pack_call(feats, weights, biases, nodes, batch, n_rows, row_offset):
    n = len(feats)
    if not 1 <= n <= 16 or len(weights) != n or len(biases) != n:
        raise RuntimeError("1..16 levels, one weight and one bias (or None) per level")
    sides, chans = [], []
    for f, w, b in zip(feats, weights, biases):
        if not f.is_cuda or f.dtype != torch.float32 or f.dim() != 4 or f.shape[0] != batch or f.shape[2] != f.shape[3] or \
                not f.is_contiguous():
            raise RuntimeError(f"level features must be contiguous CUDA float32 [batch, C_l, side, side], got {tuple(f.shape)}")
        cin = int(f.shape[1])
        if not w.is_cuda or w.dtype != torch.float32 or not w.is_contiguous() or w.numel() != C * cin or w.shape[0] != C:
            raise RuntimeError(f"level weight must be a contiguous CUDA float32 [{C}, {cin}(, 1, 1)] tensor, got {tuple(w.shape)}")
        if b is not None:
            _check_vec(b, "level bias", C)
        sides.append(int(f.shape[2]))
        chans.append(cin)
    fp = (ct.c_void_p * n)(*[f.data_ptr() for f in feats])
    wp = (ct.c_void_p * n)(*[w.data_ptr() for w in weights])
    bp = (ct.c_void_p * n)(*[(b.data_ptr() if b is not None else None) for b in biases])
    _lib.check(_lib.load().eg_conv1x1_relu_pack_levels(fp, wp, bp, (ct.c_int * n)(*chans), (ct.c_int * n)(*sides), n, batch, n_rows,
                                                       row_offset, _ptr(nodes), _stream()), "eg_conv1x1_relu_pack_levels")


class _ConvReluPackFn(torch.autograd.Function):
    """relu(conv1x1(features[l])) of every level, packed node-major, in one launch.  The backward unpacks the node gradient to
    NCHW (eg_unpack_levels) and lets torch differentiate the recomputed relu(conv2d) of each level (the small levels carry the
    wide channel counts; the frame-sized one has 4 input channels)."""

    @staticmethod
    def forward(ctx, batch, n_rows, row_offset, n_levels, *tensors):
        feats = [t.contiguous() for t in tensors[:n_levels]]
        weights = [t.contiguous() for t in tensors[n_levels:2 * n_levels]]
        biases = list(tensors[2 * n_levels:3 * n_levels])
        used = sum(int(f.shape[2]) ** 2 for f in feats)
        alloc = torch.empty if (row_offset == 0 and used == n_rows) else torch.zeros
        nodes = alloc(batch * n_rows, C, dtype=torch.float32, device=feats[0].device)
        _conv_pack_call(feats, weights, [b.contiguous() if b is not None else None for b in biases], nodes, batch, n_rows, row_offset)
        ctx.meta = (batch, n_rows, row_offset, n_levels)
        ctx.has_bias = [b is not None for b in biases]
        ctx.save_for_backward(*feats, *weights, *[b for b in biases if b is not None])
        return nodes

    @staticmethod
    def backward(ctx, d_nodes):
        batch, n_rows, row_offset, n = ctx.meta
        saved = ctx.saved_tensors
        feats, weights = saved[:n], saved[n:2 * n]
        bl = list(saved[2 * n:])
        biases = [bl.pop(0) if hb else None for hb in ctx.has_bias]
        g_maps = [torch.empty(batch, C, f.shape[2], f.shape[3], dtype=torch.float32, device=d_nodes.device) for f in feats]
        _pack_call("eg_unpack_levels", g_maps, d_nodes.contiguous(), batch, n_rows, row_offset)
        gf, gw, gb = [], [], []
        for l in range(n):
            need = (ctx.needs_input_grad[4 + l], ctx.needs_input_grad[4 + n + l], biases[l] is not None and ctx.needs_input_grad[4 + 2 * n + l])
            if not any(need):
                gf.append(None); gw.append(None); gb.append(None)
                continue
            with torch.enable_grad():
                f = feats[l].detach().requires_grad_(need[0])
                w = weights[l].detach().requires_grad_(need[1])
                b = biases[l].detach().requires_grad_(need[2]) if biases[l] is not None else None
                y = torch.relu(torch.nn.functional.conv2d(f, w.view(C, -1, 1, 1), b))
                ins = [t for t, k in ((f, need[0]), (w, need[1]), (b, need[2])) if k]
                outs = list(torch.autograd.grad(y, ins, g_maps[l]))
            gf.append(outs.pop(0) if need[0] else None)
            gw.append(outs.pop(0).view_as(weights[l]) if need[1] else None)
            gb.append(outs.pop(0) if need[2] else None)
        return (None, None, None, None, *gf, *gw, *gb)


def conv1x1_relu_pack_levels(feats, weights, biases, batch: int, n_rows: int, row_offset: int = 0,
                             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """relu(Conv2d(C_l, 128, 1)(feats[l])) for every level (coarse to fine), written node-major [batch * n_rows, 128] like
    `pack_levels` (models.py:707-710 + :726-756 in one launch).  Differentiable w.r.t. features, weights and biases (not with
    ``out=``: written in place)."""
    n = len(feats)
    if out is not None:
        feats = [f.contiguous() for f in feats]
        weights = [w.detach().contiguous() for w in weights]
        biases = [b.detach().contiguous() if b is not None else None for b in biases]
        _pack_out(out, list(feats), int(batch), int(n_rows), int(row_offset), sum(int(f.shape[2]) ** 2 for f in feats))
        _conv_pack_call(feats, weights, biases, out, int(batch), int(n_rows), int(row_offset))
        return out
    return _ConvReluPackFn.apply(int(batch), int(n_rows), int(row_offset), n, *feats, *weights, *biases)
