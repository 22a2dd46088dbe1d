"""Thin Python wrappers over the C-ABI (no compute here; pointers + stream only).

Every function enqueues on ``torch.cuda.current_stream()`` and returns torch
tensors that own the output memory.  Inputs must be CUDA fp32 contiguous
``[rows, 128]``."""
from __future__ import annotations

import ctypes as ct
from typing import Optional

import torch

from . import _lib

C = 128


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ct.c_void_p(t.data_ptr())


def _stream():
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_rows(t: torch.Tensor, name: str, rows: Optional[int] = None) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (ROCm) tensor: the HIP path has no CPU fallback")
    if t.dtype != torch.float32 or t.dim() != 2 or t.shape[1] != C or not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous float32 [rows, {C}], got {tuple(t.shape)} {t.dtype}")
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

    @classmethod
    def topo(cls, frame_size: int, num_aux_graphs: int, use_main_graph_only: bool = False,
             use_coordinate_graph: bool = False, device=None) -> "Graph":
        lib = _lib.load()
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        h = ct.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.eg_topo_create(frame_size, num_aux_graphs, int(use_main_graph_only),
                                          int(use_coordinate_graph), ct.byref(h)), "eg_topo_create")
        return cls(h, True, int(lib.eg_graph_num_nodes(h)), device)

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
        return cls(h, False, int(num_nodes), ei.device)

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
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act((A_hat x) W^T * scale + shift) + residual in one kernel."""
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
    _lib.check(_lib.load().eg_gcn_layer_fwd(graph._h, batch, _ptr(x), _ptr(weight), _ptr(scale), _ptr(shift),
                                            _ptr(residual), int(relu), int(transpose_w), _ptr(out), _stream()),
               "eg_gcn_layer_fwd")
    return out


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
