"""Host-side mirror of the reference's operator surface for the GNN hot path.

Same names, constructor arguments, call shapes and state_dict keys as the
reference (src/core/models.py:262-553 and the torch_geometric classes it
imports at :5), so a loop shaped like src/engine.py:240-262 drives the HIP
kernels unchanged and ``miccai2023.pth``-style checkpoints load ``strict=True``:

    gnn_layers.{i}.module_0.lin.weight / .bias        (GCNConv)
    gnn_layers.{i}.module_1.{weight,bias,running_*}   (BatchNorm1d)
    node_classifiers.{c}.{0,1,4,5,8}.*
    node_coordinate_mlp.{i}.{0,1,4,5,8}.*

Compute goes through the C-ABI library only (echoglad_amd/ops.py); torch is
used for parameter storage, autograd bookkeeping, streams and the tiny
per-landmark coordinate MLP (4 rows per frame).  There is no CPU fallback."""
from __future__ import annotations

import os
import sys
import weakref
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .topology import TopologySpec, candidate_specs, commutative_edge_hash, get_topology

C = ops.C


# ---------------------------------------------------------------------------
# graph resolution: incoming PyG edge_index -> implicit topology or CSR handle
# ---------------------------------------------------------------------------
_TOPO_GRAPHS: Dict[tuple, ops.Graph] = {}          # (structured spec fields, device) -> handle, shared by every resolver
_EXPECTED_HASH: Dict[tuple, Tuple[int, int]] = {}  # (spec, batch) -> (E_dir, digest) of the closed form


def _topo_graph(spec: TopologySpec, device) -> ops.Graph:
    device = torch.device(device)
    key = (spec.frame_size, 0 if spec.use_main_graph_only else spec.num_aux_graphs, spec.use_main_graph_only,
           spec.use_coordinate_graph and not spec.use_main_graph_only, device)
    g = _TOPO_GRAPHS.get(key)
    if g is None:
        g = ops.Graph.topo(spec.frame_size, spec.num_aux_graphs, spec.use_main_graph_only, spec.use_coordinate_graph,
                           device=device)
        _TOPO_GRAPHS[key] = g
    return g


def _expected_hash(spec: TopologySpec, batch: int) -> Tuple[int, int]:
    key = (spec, batch)
    if key not in _EXPECTED_HASH:
        if len(_EXPECTED_HASH) > 64:
            _EXPECTED_HASH.clear()
        _EXPECTED_HASH[key] = commutative_edge_hash(get_topology(spec).batched_edge_index(batch))
    return _EXPECTED_HASH[key]


class GraphResolver:
    """Maps an incoming ``edge_index`` to a kernel graph handle.

    Two cache levels.  (1) identity: the same live tensor object at the same ``_version`` resolves without touching
    the device (a weak reference is kept, so a freed-and-reallocated tensor at the same address can never hit).
    (2) content: anything else is digested on the device (``eg_edge_hash``: edge count + order-independent 64-bit sum,
    one 16-byte read-back) and looked up by ``(device, rows, E, digest)``; only an unseen digest builds a handle.

    A handle is the implicit-stencil topology when the digest equals the closed form's — of the model's own static
    topology (``spec``), or, for a stand-alone ``GCNConv`` (``spec=None``: it is constructed without any graph
    information, models.py:330-331), of whichever structured closed form has these node and edge counts
    (``topology.candidate_specs``) — and a CSR built from the edge_index otherwise."""

    MAX_HANDLES = 16

    def __init__(self, spec: Optional[TopologySpec] = None):
        self.spec = spec
        self._ident: Dict[int, tuple] = {}
        self._by_digest: "OrderedDict[tuple, Tuple[ops.Graph, int]]" = OrderedDict()

    def topo_graph(self, device) -> ops.Graph:
        return _topo_graph(self.spec, device)

    def _candidates(self, num_rows: int, n_edges: int):
        if self.spec is not None:
            topo = get_topology(self.spec)
            if topo.is_structured() and num_rows % topo.num_nodes == 0:
                batch = num_rows // topo.num_nodes
                if n_edges == batch * 2 * topo.num_undirected_edges:
                    return [(self.spec, batch)]
            return []
        return candidate_specs(num_rows, n_edges)

    def resolve(self, edge_index: torch.Tensor, num_rows: int) -> Tuple[ops.Graph, int]:
        ent = self._ident.get(id(edge_index))
        if ent is not None and ent[0]() is edge_index and ent[1] == edge_index._version and ent[2] == num_rows:
            return ent[3]
        n_edges, digest = ops.edge_hash(edge_index)
        key = (edge_index.device, num_rows, n_edges, digest)
        result = self._by_digest.get(key)
        if result is None:
            for spec, batch in self._candidates(num_rows, n_edges):
                if _expected_hash(spec, batch) == (n_edges, digest):
                    result = (_topo_graph(spec, edge_index.device), batch)
                    break
            if result is None:
                result = (ops.Graph.csr(edge_index, num_rows), 1)
            while len(self._by_digest) >= self.MAX_HANDLES:
                self._by_digest.popitem(last=False)       # (a handle still referenced elsewhere, e.g. by a captured HIP graph, lives on)
            self._by_digest[key] = result
        else:
            self._by_digest.move_to_end(key)
        if len(self._ident) > 64:
            self._ident = {k: v for k, v in self._ident.items() if v[0]() is not None}
            if len(self._ident) > 64:
                self._ident.clear()
        self._ident[id(edge_index)] = (weakref.ref(edge_index), edge_index._version, num_rows, result)
        return result


_SHARED_RESOLVER = GraphResolver(None)      # every stand-alone GCNConv: the layers of a stack see the same edge_index


# ---------------------------------------------------------------------------
# autograd functions over the C-ABI
# ---------------------------------------------------------------------------
class _GCNConvFn(torch.autograd.Function):
    """y = A_hat x W^T + b.  Backward: dx = (A_hat^T dy) W, dW = (A_hat^T dy)^T x, db = sum dy
    (graph.bwd is the graph itself whenever A_hat is symmetric, i.e. for every undirected edge_index)."""

    @staticmethod
    def forward(ctx, x, weight, bias, graph, batch):
        ctx.graph, ctx.batch = graph, batch
        ctx.save_for_backward(x, weight)
        return ops.gcn_layer_fwd(graph, batch, x.contiguous(), weight.contiguous(), None,
                                 bias.contiguous() if bias is not None else None, None, False)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.gcn_layer_fwd(ctx.graph.bwd, ctx.batch, dy, weight.contiguous(), None, None, None, False,
                                   transpose_w=True)
        if ctx.needs_input_grad[1]:
            g = ops.gcn_aggregate(ctx.graph.bwd, ctx.batch, dy)
            dw = ops.dweight128(g, x.contiguous())
        if ctx.needs_input_grad[2]:
            db = ops.colsum128(dy)
        return dx, dw, db, None, None


# ---- who owns an incoming gradient buffer? -------------------------------------------------------------------------------
# _LayerTrainFn.backward and _CoordScatterFn.backward work IN PLACE on the gradient they receive (4 coordinate rows per frame
# are patched; nothing of [B*N,128] size is cloned, filled or added).  Autograd does not promise that the buffer is theirs
# alone: Add hands ONE tensor to both inputs (as an expanded view), a tensor hook may keep the very object, a later consumer
# may read it.  The exclusive case has a fixed signature in a given torch build -- (python references, TensorImpl use count,
# storage use count, no view base) -- which is measured once on a toy function of the same arity; anything else is cloned.
def _grad_signature(t: torch.Tensor) -> tuple:
    return (sys.getrefcount(t), t._use_count(), torch._C._storage_Use_Count(t.untyped_storage()._cdata))


_OWN_SIG: Dict[int, Optional[tuple]] = {}


def _probe_entry(dy: torch.Tensor, n_out: int, seen: list) -> None:
    seen.append(_grad_signature(dy))


def _calibrate_ownership(n_out: int) -> Optional[tuple]:
    seen = []

    class _Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return tuple(x * float(k + 2) for k in range(n_out)) if n_out > 1 else x * 2.0

        @staticmethod
        def backward(ctx, dy, *rest):
            _probe_entry(dy, n_out, seen)              # same call depth as _own_or_clone below (the frames hold references too)
            return dy

    try:
        with torch.enable_grad():
            x = torch.zeros(2, 2, requires_grad=True)
            y = _Probe.apply(x)
            first = y[0] if n_out > 1 else y
            loss = (first * 3.0).sum()                # a fresh, exclusively owned gradient reaches the probe
            for other in (y[1:] if n_out > 1 else ()):
                loss = loss + (other * 5.0).sum()
            loss.backward()
        return seen[0]
    except Exception:                               # an unknown torch build: never assume ownership
        return None


def _own_or_clone(dy: torch.Tensor, n_out: int) -> torch.Tensor:
    """``dy`` itself when this backward provably holds the only references to it, else a contiguous copy."""
    if n_out not in _OWN_SIG:
        _OWN_SIG[n_out] = _calibrate_ownership(n_out)
    base = _OWN_SIG[n_out]
    if base is None or dy._base is not None or not dy.is_contiguous():
        return dy.contiguous() if (dy._base is None and not dy.is_contiguous()) else dy.clone(memory_format=torch.contiguous_format)
    sig = _grad_signature(dy)
    if sig[0] > base[0] or sig[1] > base[1] or sig[2] > base[2]:
        return dy.clone(memory_format=torch.contiguous_format)
    return dy


def _bn_step(bn: nn.BatchNorm1d):
    """What nn.BatchNorm1d.forward decides before calling F.batch_norm: (use batch statistics?, update factor | None).
    Counts the batch in ``num_batches_tracked``; ``momentum=None`` is the cumulative moving average."""
    use_batch = bn.training or bn.running_mean is None
    factor = None
    if bn.training and bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            bn.num_batches_tracked += 1
        factor = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else float(bn.momentum)
    return use_batch, factor


class _LayerTrainFn(torch.autograd.Function):
    """One whole train-mode layer as a single autograd node over the two C-ABI composites
    (eg_gcn_layer_train_fwd / eg_gcn_layer_bwd; models.py:328-335, :431-435):
        z = A_hat x W^T + b;  out = relu|id(dropout(BN_batch(z))) + x.
    Kept for the backward: z, the aggregated input A_hat x (so that dW = dz^T (A_hat x) needs no second aggregation) and the
    batch statistics — not x.  Returns (out, lm): lm = out's coordinate-node rows [4B,128] (``coord_rows`` = (B, n, first
    row), else an empty tensor), i.e. the gather of models.py:447 done here so that its gradient comes back into THIS node
    and is added into 4 rows per frame, instead of autograd summing two dense [B*N,128] gradients."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, graph, batch, relu, p, momentum, eps, seed,
                residual, coord_rows):
        x = x.contiguous()
        need_w = weight.requires_grad
        out, z, agg, bn = ops.gcn_layer_train_fwd(graph, batch, x, weight.contiguous(), bias.contiguous(), gamma.contiguous(),
                                                  beta.contiguous(), running_mean, running_var, momentum, eps, relu, p, seed,
                                                  residual, want_agg=need_w)
        ctx.save_for_backward(z, agg if agg is not None else z.new_zeros(0), weight.detach().contiguous(),
                              gamma.detach().contiguous(), beta.detach().contiguous(), bn)
        ctx.cfg = (graph, batch, relu, p, seed, residual, coord_rows, need_w)
        if coord_rows is not None:
            B, n, lo = coord_rows
            lm = out.view(B, n, C)[:, lo:lo + 4, :].reshape(B * 4, C).clone()     # a copy: out is overwritten in place later
        else:
            lm = out.new_zeros(0)
        return out, lm

    @staticmethod
    def backward(ctx, dy, dlm):
        z, agg, weight, gamma, beta, bn = ctx.saved_tensors
        graph, batch, relu, p, seed, residual, coord_rows, had_agg = ctx.cfg
        if coord_rows is not None and dlm is not None:
            dy = _own_or_clone(dy, 2)                                     # in place only on a buffer that is provably ours
            B, n, lo = coord_rows
            dy.view(B, n, C)[:, lo:lo + 4, :] += dlm.view(B, 4, C)
        else:
            dy = dy.contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and had_agg, ctx.needs_input_grad[2]
        dx, dw, db, dgamma, dbeta = ops.gcn_layer_bwd(graph.bwd, batch, dy, z, agg if had_agg else None, weight, gamma, beta, bn,
                                                      relu, p, seed, residual, need_x, need_w)
        return (dx, dw, db if need_b else None, dgamma, dbeta) + (None,) * 11


class _CoordScatterFn(torch.autograd.Function):
    """models.py:455-473 in place: sample the layer output's main grid at the updated landmark coordinates
    (bilinear_interpolation as a 4-tap gather) and overwrite the frame's 4 coordinate-node rows with the samples.
    Backward works in place on the incoming gradient buffer as well: the coordinate rows' gradient becomes the samples'
    gradient, those rows are zeroed (their old values were overwritten) and the 16 taps per frame are accumulated into the
    main-grid rows; d/d coords comes from the same kernel."""

    @staticmethod
    def forward(ctx, h, coords, batch, n, main_base, frame, coord_base):
        coords = coords.reshape(batch * 4, 2).contiguous()
        new = ops.bilinear4_fwd(h, coords, batch, n, main_base, frame)
        h.view(batch, n, C)[:, coord_base:coord_base + 4, :] = new.view(batch, 4, C)
        ctx.mark_dirty(h)
        ctx.save_for_backward(h, coords)
        ctx.dims = (batch, n, main_base, frame, coord_base)
        return h

    @staticmethod
    def backward(ctx, dh):
        h, coords = ctx.saved_tensors
        batch, n, main_base, frame, coord_base = ctx.dims
        dh = _own_or_clone(dh, 1)                                          # in place only on a buffer that is provably ours
        rows = dh.view(batch, n, C)[:, coord_base:coord_base + 4, :]
        dnew = rows.reshape(batch * 4, C).clone()
        rows.zero_()
        dcoords = ops.bilinear4_bwd(dnew, h, coords, batch, n, main_base, frame, dh=dh, want_dcoords=ctx.needs_input_grad[1])
        return dh, (dcoords.view(batch, 4, 2) if dcoords is not None else None), None, None, None, None, None


_HEAD_PARAM_IDX = ((0, "weight"), (0, "bias"), (1, "weight"), (1, "bias"), (4, "weight"), (4, "bias"), (5, "weight"),
                   (5, "bias"), (8, "weight"), (8, "bias"))


class _ClassifierTrainFn(torch.autograd.Function):
    """models.py:363-377, :485-490 in train mode over eg_classifier_train_fwd / eg_classifier_bwd: node-type filter + the
    four heads as one stacked network (first layers one [128 -> 128] product, 4 x BatchNorm1d(32) == BatchNorm1d(128) on the
    stacked output; second layers block-diagonal [128 -> 64]; third a 16-wide dot).  ``params`` = for each head its 10
    parameters in _HEAD_PARAM_IDX order; gradients go back to each of them."""

    @staticmethod
    def forward(ctx, h, batch, n, row_lo, n_valid, sigmoid, cfg, *params):
        heads = [params[10 * k:10 * k + 10] for k in range(4)]
        cat = lambda j: torch.cat([hd[j].reshape(-1) if hd[j].dim() == 1 else hd[j] for hd in heads], dim=0).contiguous()
        P = dict(cfg)
        P.update(w1=cat(0), b1=cat(1), gamma1=cat(2), beta1=cat(3), w2=torch.stack([hd[4] for hd in heads]).contiguous(),
                 b2=cat(5), gamma2=cat(6), beta2=cat(7), w3=cat(8), b3=cat(9))
        h = h.contiguous()
        logits, z1, z2, bn = ops.classifier_train_fwd(h, batch, n, row_lo, n_valid, P, sigmoid)
        ctx.P = {k: v for k, v in P.items() if not k.startswith("running")}
        ctx.dims = (batch, n, row_lo, n_valid, sigmoid)
        ctx.save_for_backward(h, z1, z2, bn, logits if sigmoid else logits.new_zeros(0))
        return logits

    @staticmethod
    def backward(ctx, dy):
        h, z1, z2, bn, y = ctx.saved_tensors
        batch, n, row_lo, n_valid, sigmoid = ctx.dims
        dl = dy.contiguous()
        if sigmoid:
            dl = dl * y * (1.0 - y)
        dh, g = ops.classifier_bwd(dl, h, batch, n, row_lo, n_valid, ctx.P, z1, z2, bn, ctx.needs_input_grad[0])
        dw1, db1, dg1, dbe1 = g[:16384].view(4, 32, C), g[16384:16512].view(4, 32), g[16512:16640].view(4, 32), g[16640:16768].view(4, 32)
        o = 16768
        dw2 = g[o:o + 2048].view(4, 16, 32)
        db2, dg2, dbe2 = (g[o + 2048 + 64 * k:o + 2048 + 64 * (k + 1)].view(4, 16) for k in range(3))
        dw3 = g[o + 2240:o + 2304].view(4, 1, 16)
        db3 = g[o + 2304:o + 2308].view(4, 1)
        grads = []
        for k in range(4):
            grads += [dw1[k], db1[k], dg1[k], dbe1[k], dw2[k], db2[k], dg2[k], dbe2[k], dw3[k], db3[k]]
        return (dh, None, None, None, None, None, None) + tuple(grads)


_MLP_NAMES = ("w1", "b1", "gamma1", "beta1", "w2", "b2", "gamma2", "beta2", "w3", "b3")


class _CoordMlpFn(torch.autograd.Function):
    """models.py:441-453 in train mode as one autograd node over eg_coord_mlp_fwd / eg_coord_mlp_bwd:
    (landmark rows [4B,128], coords [B,4,2]) -> clamp(coords + node_coordinate_mlp(cat(lm, pairwise offsets)), 0, frame-1).
    ``params`` = the head's 10 parameters in _HEAD_PARAM_IDX order."""

    @staticmethod
    def forward(ctx, lm, coords, batch, frame, cfg, *params):
        P = dict(cfg)
        P.update({k: p.detach().contiguous() for k, p in zip(_MLP_NAMES, params)})
        lm = lm.contiguous()
        flat = coords.reshape(batch * 4, 2).contiguous()
        new, saved = ops.coord_mlp_fwd(lm, flat, batch, P, True, frame, True)
        ctx.P = {k: v for k, v in P.items() if not k.startswith("running")}
        ctx.dims = (batch, frame)
        ctx.save_for_backward(lm, flat, *saved)
        return new.view(batch, 4, 2)

    @staticmethod
    def backward(ctx, dnew):
        lm, flat, z1, z2, bn, pre = ctx.saved_tensors
        batch, frame = ctx.dims
        dlm, dc, g = ops.coord_mlp_bwd(dnew.contiguous().view(batch * 4, 2), lm, flat, batch, ctx.P, frame, (z1, z2, bn, pre),
                                       ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        sizes = (32 * 136, 32, 32, 32, 16 * 32, 16, 16, 16, 2 * 16, 2)
        parts = torch.split(g, sizes)
        grads = (parts[0].view(32, 136), parts[1], parts[2], parts[3], parts[4].view(16, 32), parts[5], parts[6], parts[7],
                 parts[8].view(2, 16), parts[9])
        return (dlm, dc.view(batch, 4, 2) if dc is not None else None, None, None, None) + grads


# ---------------------------------------------------------------------------
# torch_geometric-compatible modules
# ---------------------------------------------------------------------------
class _GlorotLinear(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        a = (6.0 / (in_channels + out_channels)) ** 0.5
        nn.init.uniform_(self.weight, -a, a)


class GCNConv(nn.Module):
    """Counterpart of ``torch_geometric.nn.GCNConv(in_channels, out_channels)`` as the
    reference constructs it (src/core/models.py:330-331: defaults improved=False,
    cached=False, add_self_loops=True, normalize=True, bias=True).
    ``forward(x, edge_index) -> x``.  Only 128 -> 128 is built (default.yml:13-14)."""

    def __init__(self, in_channels: int, out_channels: int, **kwargs):
        super().__init__()
        if in_channels != C or out_channels != C:
            raise NotImplementedError(f"the HIP GCNConv is built for {C}->{C} channels, got {in_channels}->{out_channels}")
        for k, default in (("improved", False), ("cached", False), ("add_self_loops", True), ("normalize", True),
                           ("bias", True)):
            if kwargs.get(k, default) != default:
                raise NotImplementedError(f"GCNConv({k}={kwargs[k]!r}) is not supported")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = _GlorotLinear(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        graph, batch = _SHARED_RESOLVER.resolve(edge_index, x.shape[0])
        return self.forward_graph(x, graph, batch)

    def forward_graph(self, x, graph: ops.Graph, batch: int) -> torch.Tensor:
        return _GCNConvFn.apply(x, self.lin.weight, self.bias, graph, batch)


class Sequential(nn.Module):
    """Counterpart of ``torch_geometric.nn.Sequential('x, edge_index', [(conv, 'x, edge_index -> x'), m, ...])``
    (src/core/models.py:329-335): children are registered as ``module_{i}``."""

    def __init__(self, input_args: str, modules: Sequence):
        super().__init__()
        self._takes_graph: List[bool] = []
        for i, m in enumerate(modules):
            takes = False
            if isinstance(m, (tuple, list)):
                m, desc = m
                takes = "edge_index" in desc.split("->")[0]
            self.add_module(f"module_{i}", m)
            self._takes_graph.append(takes)

    def __len__(self):
        return len(self._takes_graph)

    def __getitem__(self, i):
        return getattr(self, f"module_{i}")

    def forward(self, x, edge_index):
        for i, takes in enumerate(self._takes_graph):
            m = getattr(self, f"module_{i}")
            x = m(x, edge_index) if takes else m(x)
        return x

    def forward_graph(self, x, graph: ops.Graph, batch: int):
        for i, takes in enumerate(self._takes_graph):
            m = getattr(self, f"module_{i}")
            x = m.forward_graph(x, graph, batch) if takes else m(x)
        return x


class JumpingKnowledge(nn.Module):
    def __init__(self, mode: str):
        super().__init__()
        if mode not in ("max",):
            raise NotImplementedError("only gnn_jk_mode in ('last', 'max') is supported "
                                      "('cat' cannot work in the reference either: models.py:365)")
        self.mode = mode

    def forward(self, xs):
        return torch.stack(xs, dim=-1).max(dim=-1)[0]


# ---------------------------------------------------------------------------
# parameter folding for the inference kernels
# ---------------------------------------------------------------------------
def _fold_bn(bn: nn.BatchNorm1d, lin_bias: Optional[torch.Tensor]):
    """eval-mode BN(z + b) == z * scale + shift."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    if lin_bias is not None:
        shift = shift + lin_bias * scale
    return scale.contiguous(), shift.contiguous()


def _versions(module: nn.Module) -> tuple:
    return tuple(t._version for t in list(module.parameters()) + list(module.buffers())) + \
           tuple(t.data_ptr() for t in module.parameters())


def _mlp_head(in_f, hid, out_f, drop_p, last):
    return nn.Sequential(nn.Linear(in_f, hid), nn.BatchNorm1d(hid), nn.ReLU(inplace=True), nn.Dropout(p=drop_p),
                         nn.Linear(hid, hid // 2), nn.BatchNorm1d(hid // 2), nn.ReLU(inplace=True),
                         nn.Dropout(p=drop_p), nn.Linear(hid // 2, out_f), last)


class HierarchicalPatchModel(nn.Module):
    """Counterpart of the reference ``HierarchicalPatchModel`` (src/core/models.py:262-553).

    ``forward(data_batch=None, x=, node_coords=, edge_index=, node_type=, batch_idx=)``
    -> ``(logits [B*N_valid, n_out] (squeezed), node_coords [4B,2] | None)`` exactly as
    engine.py:248-255 calls it.  ``forward_nodes`` enters at the node features
    ``[B*N, 128]`` — the interval the throughput metric is defined on."""

    def __init__(self, frame_size: int = 32, gnn_dropout_p: float = 0.0, classifier_dropout_p: float = 0.0,
                 node_embedding_dim: int = 128, node_hidden_dim: int = 64, num_output_channels: int = 4,
                 num_gnn_layers: int = 3, num_aux_graphs: int = 4, gnn_jk_mode: str = "last",
                 classifier_hidden_dim: int = 16, residual: bool = True, use_coordinate_graph: bool = False,
                 output_activation: str = "sigmoid", use_connection_nodes=False, use_main_graph_only=False):
        super().__init__()
        if gnn_jk_mode not in ("last", "max", "cat"):
            raise ValueError("Only last, max or cat jumping knowledge mode is supported.")
        if node_embedding_dim != C or node_hidden_dim != C:
            raise NotImplementedError(f"the HIP path is built for node_embedding_dim = node_hidden_dim = {C}")
        self.gnn_layers = nn.ModuleList()
        self.node_coordinate_mlp = nn.ModuleList()
        for i in range(num_gnn_layers):
            self.gnn_layers.append(Sequential("x, edge_index", [
                (GCNConv(in_channels=node_embedding_dim if i == 0 else node_hidden_dim,
                         out_channels=node_hidden_dim), "x, edge_index -> x"),
                nn.BatchNorm1d(node_hidden_dim),
                nn.Dropout(p=gnn_dropout_p),
                nn.Identity() if i == num_gnn_layers - 1 else nn.ReLU(inplace=True)]))
            if use_coordinate_graph:
                self.node_coordinate_mlp.append(
                    _mlp_head(node_hidden_dim + 8, classifier_hidden_dim, 2, classifier_dropout_p, nn.Identity()))
        self.output_activation = output_activation
        if output_activation == "sigmoid":
            make_last = nn.Sigmoid
        elif output_activation == "logit":
            make_last = nn.Identity
        else:
            raise ValueError(f"invalid output_activation:{output_activation}")
        self.node_classifiers = nn.ModuleList(
            [_mlp_head(node_hidden_dim, classifier_hidden_dim, 1, classifier_dropout_p, make_last())
             for _ in range(num_output_channels)])
        self.jk = JumpingKnowledge(gnn_jk_mode) if gnn_jk_mode != "last" else None
        self.frame_size = frame_size
        self.residual = residual
        self.num_gnn_layers = num_gnn_layers
        self.node_embedding_dim = node_embedding_dim
        self.num_aux_graphs = num_aux_graphs
        self.use_coordinate_graph = use_coordinate_graph
        self.use_connection_nodes = use_connection_nodes
        self.use_main_graph_only = use_main_graph_only
        self.classifier_hidden_dim = classifier_hidden_dim
        self.num_output_channels = num_output_channels
        # static topology implied by the constructor arguments (datasets.py:1441-1584); the graph
        # *type* ('grid' vs 'grid-diagonal') is dataset config, so it is verified per edge_index.
        self.topology_spec = TopologySpec(frame_size=frame_size, num_aux_graphs=num_aux_graphs,
                                          use_main_graph_only=bool(use_main_graph_only),
                                          use_coordinate_graph=bool(use_coordinate_graph),
                                          use_connection_nodes=bool(use_connection_nodes))
        self._resolver = GraphResolver(self.topology_spec)
        self._fold_cache: Dict[str, tuple] = {}
        self._hip_graphs: Dict[tuple, tuple] = {}
        self.use_hip_graph = False
        # eval path: layer i leaves the child sums of its output in a side buffer for layer i+1
        # (eg_gcn_layer_fwd_chain); EG_CHAIN=0 runs every layer on its own
        self.chain_layers = os.environ.get("EG_CHAIN", "1") != "0"
        # ... and the last layer runs the classifier heads on its output tile inside the kernel (EG_FUSE_CLS=0: separate)
        self.fuse_classifier = os.environ.get("EG_FUSE_CLS", "1") != "0"
        self._kidsum: Dict[tuple, tuple] = {}
        # optional callable (layer index, layer output incl. residual and coordinate rows) -> None, called by forward_nodes
        self.layer_output_hook = None

    def enable_hip_graph(self, flag: bool = True) -> "HierarchicalPatchModel":
        """Inference only: capture the kernel sequence of ``forward_nodes`` (3 fused layers + classifier
        + queue resets) into a HIP graph the first time a given input buffer is seen and replay it on
        later calls with the same buffers (same data_ptr / shape / weights).  The returned logits
        tensor is owned by the graph and overwritten by the next replay."""
        self.use_hip_graph = bool(flag)
        self._hip_graphs.clear()
        return self

    # ---- static row ranges (replace the reference's node_type host syncs, models.py:447,456,473,485)
    def _row_ranges(self):
        topo = get_topology(self.topology_spec)
        return topo.num_nodes, topo.n_conn, topo.num_valid_nodes, topo.main.base, topo.coord_base

    # ---- folded inference parameters, cached on parameter versions -------------------------
    def _folded_layers(self):
        key = tuple(_versions(l) for l in self.gnn_layers)
        hit = self._fold_cache.get("layers")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                vals = []
                for l in self.gnn_layers:
                    conv, bn = l.module_0, l.module_1
                    scale, shift = _fold_bn(bn, conv.bias)
                    vals.append((conv.lin.weight.detach().contiguous(), scale, shift))
            hit = (key, vals)
            self._fold_cache["layers"] = hit
        return hit[1]

    def _packed_classifier(self):
        key = tuple(_versions(c) for c in self.node_classifiers)
        hit = self._fold_cache.get("cls")
        if hit is None or hit[0] != key:
            if self.num_output_channels != 4 or self.classifier_hidden_dim != 32:
                raise NotImplementedError("the fused classifier kernel is built for 4 heads of 128-32-16-1")
            with torch.no_grad():
                w1 = torch.cat([c[0].weight for c in self.node_classifiers], dim=0)           # [128,128]
                st1 = [_fold_bn(c[1], c[0].bias) for c in self.node_classifiers]
                w2 = torch.stack([c[4].weight for c in self.node_classifiers], dim=0)          # [4,16,32]
                st2 = [_fold_bn(c[5], c[4].bias) for c in self.node_classifiers]
                w3 = torch.cat([c[8].weight for c in self.node_classifiers], dim=0)            # [4,16]
                b3 = torch.cat([c[8].bias for c in self.node_classifiers], dim=0)              # [4]
                packed = {"w1": w1.contiguous(), "s1": torch.cat([s for s, _ in st1]).contiguous(),
                          "t1": torch.cat([t for _, t in st1]).contiguous(), "w2": w2.contiguous(),
                          "s2": torch.cat([s for s, _ in st2]).contiguous(),
                          "t2": torch.cat([t for _, t in st2]).contiguous(), "w3": w3.contiguous(),
                          "b3": b3.contiguous()}
            hit = (key, packed)
            self._fold_cache["cls"] = hit
        return hit[1]

    # ---- one GNN layer in train mode: one autograd node over eg_gcn_layer_train_fwd / eg_gcn_layer_bwd ------------------
    def _layer_train(self, i: int, x_in: torch.Tensor, graph: ops.Graph, gb: int, coord_rows=None):
        """-> (h, lm): lm = the coordinate-node rows of h (models.py:447) when ``coord_rows`` is given, else None."""
        layer = self.gnn_layers[i]
        conv, bn, drop = layer.module_0, layer.module_1, layer.module_2
        p = float(drop.p)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0      # host RNG: reproducible under torch.manual_seed
        relu = i < self.num_gnn_layers - 1
        if not (bn.training and bn.affine and drop.training):
            # a frozen (eval-mode) BatchNorm / Dropout inside a training model: GCNConv kernel + the torch modules
            h = layer.forward_graph(x_in, graph, gb)
            return (h + x_in if self.residual else h), None
        _, momentum = _bn_step(bn)
        h, lm = _LayerTrainFn.apply(x_in, conv.lin.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                    graph, gb, relu, p, momentum, bn.eps, seed, bool(self.residual), coord_rows)
        return h, (lm if coord_rows is not None else None)

    # ---- coordinate-graph update (models.py:438-473) ---------------------------------------
    def _coordinate_update(self, i: int, h: torch.Tensor, node_coords: torch.Tensor, batch: int, lm=None):
        n, _, _, main_base, coord_base = self._row_ranges()
        fs = self.frame_size
        # lm given: h is the fresh output of _LayerTrainFn, which nothing else has saved for its backward -- only then may
        # the resampled rows overwrite it in place (the output of a torch module, e.g. the ReLU of the frozen-BatchNorm
        # fallback, is saved by that module and must not be written to)
        fresh_layer_output = lm is not None
        # pairwise (other - self) offsets per frame, flattened to 8 numbers per landmark (:441-444)
        if lm is None:
            lm = h.view(batch, n, C)[:, coord_base:, :].reshape(batch * 4, C).clone()
        new_coords = self._coord_mlp_kernel(self.node_coordinate_mlp[i], lm, node_coords, batch, fs)
        if new_coords is None:
            # a mix of frozen and training sub-modules, or eval mode with gradients: the torch modules, op by op
            shape_feats = (node_coords.unsqueeze(1) - node_coords.unsqueeze(2)).reshape(batch * 4, 8)
            delta = self.node_coordinate_mlp[i](torch.cat((lm, shape_feats), dim=1))
            new_coords = torch.clamp(node_coords + delta.view(batch, 4, 2), min=0, max=fs - 1)
        node_coords = new_coords
        if fresh_layer_output and torch.is_grad_enabled() and (h.requires_grad or node_coords.requires_grad) and h.grad_fn is not None:
            # train step: in place on the layer output, in place on its gradient (no [B*N,128] copies, fills or adds)
            h = _CoordScatterFn.apply(h, node_coords, batch, n, main_base, fs, coord_base)
        else:
            new_feats = ops.bilinear4(h, node_coords, batch, n, main_base, fs)            # [4B, 128]
            h = ops.scatter_coord_rows(h, new_feats, batch, n, coord_base)
        return h, node_coords

    def _coord_mlp_kernel(self, mlp: nn.Sequential, lm: torch.Tensor, node_coords: torch.Tensor, batch: int, frame: int):
        """models.py:441-453 on eg_coord_mlp_fwd / _bwd (one launch each way) -> new coords [B,4,2], or None when the
        module states are not ones the kernel implements."""
        if self.classifier_hidden_dim != 32 or self.node_embedding_dim != C or node_coords.shape[-1] != 2 or node_coords.dtype != torch.float32 or \
                os.environ.get("EG_COORD_MLP_KERNEL", "1") == "0":
            return None
        bn1, bn2, d1, d2 = mlp[1], mlp[5], mlp[3], mlp[7]
        if not (bn1.affine and bn2.affine):
            return None
        params = [getattr(mlp[j], name) for j, name in _HEAD_PARAM_IDX]
        cfg = dict(eps1=bn1.eps, eps2=bn2.eps, p1=float(d1.p), p2=float(d2.p), seed1=0, seed2=0,
                   running_mean1=bn1.running_mean, running_var1=bn1.running_var, running_mean2=bn2.running_mean,
                   running_var2=bn2.running_var)
        if bn1.training and bn2.training and d1.training and d2.training:
            if cfg["p1"] > 0 or cfg["p2"] > 0:
                cfg["seed1"], cfg["seed2"] = torch.randint(0, 2 ** 62, (2,)).tolist()     # host RNG, like the layers
            _, cfg["momentum1"] = _bn_step(bn1)
            _, cfg["momentum2"] = _bn_step(bn2)
            return _CoordMlpFn.apply(lm, node_coords, batch, frame, cfg, *params)
        frozen = not (bn1.training or bn2.training or d1.training or d2.training)
        needs_grad = torch.is_grad_enabled() and (lm.requires_grad or node_coords.requires_grad or
                                                  any(p.requires_grad for p in params))
        if frozen and not needs_grad and bn1.running_mean is not None and bn2.running_mean is not None:
            P = dict(cfg, momentum1=None, momentum2=None)
            P.update({k: p.detach().contiguous() for k, p in zip(_MLP_NAMES, params)})
            new, _ = ops.coord_mlp_fwd(lm.contiguous(), node_coords.reshape(batch * 4, 2).contiguous(), batch, P, False, frame,
                                       False)
            return new.view(batch, 4, 2)
        return None

    # ---- the hot path ------------------------------------------------------------------------
    def forward_nodes(self, node_feats: torch.Tensor, edge_index: torch.Tensor, batch: Optional[int] = None,
                      node_coords: Optional[torch.Tensor] = None):
        """node_feats [B*N,128] -> (logits [B*N_valid, n_out], node_coords | None)."""
        graph, gb = self._resolver.resolve(edge_index, node_feats.shape[0])
        n, n_conn, n_valid, _, _ = self._row_ranges()
        if node_feats.shape[0] % n != 0:
            raise RuntimeError(f"{node_feats.shape[0]} node rows is not a multiple of the {n} nodes per frame")
        B = node_feats.shape[0] // n
        if batch is not None and batch != B:
            raise RuntimeError(f"batch_idx implies {batch} frames but the node rows imply {B}")
        if self.use_coordinate_graph:
            node_coords = node_coords.reshape(B, 4, -1)
        else:
            node_coords = None
        fused = (not self.training) and (not torch.is_grad_enabled() or not node_feats.requires_grad)
        fused = fused and self.layer_output_hook is None and not any(
            p.requires_grad and torch.is_grad_enabled() for p in self.parameters())
        # JumpingKnowledge('max') stays on the fused path as a running maximum written by the layer kernels
        # (eg_gcn_layer_fwd_jk); where those do not cover the handle (CSR graphs, coordinate / connection nodes) the
        # layers run one by one and torch takes the maximum, as before
        jk_fused = (fused and self.jk is not None and graph.fused_classifier_ok and not self.use_coordinate_graph
                    and os.environ.get("EG_JK_FUSED", "1") != "0")
        fused = fused and (self.jk is None or jk_fused)
        if fused and self.use_hip_graph and not self.use_coordinate_graph and not torch.cuda.is_current_stream_capturing():
            return self._forward_nodes_graphed(node_feats, edge_index, B), None
        hidden = [node_feats.contiguous()]
        kid = (None, None)
        if fused:
            folded = self._folded_layers()
            # chained layers: each layer leaves the child sums of its output behind for the next one
            if self.chain_layers and not self.use_coordinate_graph and graph.kidsum_rows > 0 and self.num_gnn_layers > 1:
                kid = self._kidsum_buffers(graph, gb)
        fuse_cls = (fused and self.fuse_classifier and graph.fused_classifier_ok and not self.use_coordinate_graph
                    and (kid[0] is not None or graph.kidsum_rows == 0) and n_conn == 0 and n_valid == n
                    and self.num_output_channels == 4 and self.classifier_hidden_dim == 32)
        jkb = self._jk_buffers(graph, gb, node_feats) if jk_fused else None
        for i in range(self.num_gnn_layers):
            x_in = hidden[i]
            if fused:
                w, scale, shift = folded[i]
                last = i == self.num_gnn_layers - 1
                jk_prev = None if not jk_fused else (x_in if i == 0 else jkb[(i + 1) & 1])     # max over node features, h_1 .. h_i
                if last and fuse_cls:
                    # the last layer hands its output tile to the classifier heads inside the kernel
                    out = ops.gcn_layer_cls_fwd(graph, gb, x_in, w, scale, shift, x_in if self.residual else None, False,
                                                self._packed_classifier(), sigmoid=(self.output_activation == "sigmoid"),
                                                kidsum_in=kid[(i + 1) & 1] if i > 0 else None, jk_in=jk_prev)
                    return out.squeeze(1), None
                h = ops.gcn_layer_fwd(graph, gb, x_in, w, scale, shift, x_in if self.residual else None,
                                      relu=not last, kidsum_in=kid[(i + 1) & 1] if i > 0 else None,
                                      kidsum_out=None if last else kid[i & 1], jk_in=jk_prev,
                                      jk_out=jkb[i & 1] if jk_fused else None)
            elif self.training:
                _, _, _, _, coord_base = self._row_ranges()
                h, lm = self._layer_train(i, x_in, graph, gb, (B, n, coord_base) if self.use_coordinate_graph else None)
            else:
                h = self.gnn_layers[i].forward_graph(x_in, graph, gb)
                if self.residual and h.shape[1] == x_in.shape[1]:
                    h = h + x_in
            if self.use_coordinate_graph:
                h, node_coords = self._coordinate_update(i, h, node_coords, B, lm if (self.training and not fused) else None)
            if self.layer_output_hook is not None:
                self.layer_output_hook(i, h)              # e.g. h.retain_grad() / h.register_hook(...) in a test
            hidden.append(h)
        if jk_fused:
            h = jkb[(self.num_gnn_layers - 1) & 1]
        else:
            h = self.jk(hidden) if self.jk is not None else hidden[-1]
        if fused:
            out = ops.classifier_fwd(h, B, n, n_conn, n_valid, self._packed_classifier(),
                                     sigmoid=(self.output_activation == "sigmoid"))
        else:
            if self.training and self._stacked_heads_ok():
                out = self._classifier_train(h, B, n, n_conn, n_valid)
            else:
                hv = h.view(B, n, C)[:, n_conn:n_conn + n_valid, :].reshape(B * n_valid, C)
                out = torch.cat([clf(hv) for clf in self.node_classifiers], dim=1)
        if self.use_coordinate_graph:
            node_coords = node_coords.reshape(B * 4, -1)
        return out.squeeze(1), node_coords

    # ---- the 4 classifier heads in train mode as ONE stacked network ----------------------------------------
    def _stacked_heads_ok(self) -> bool:
        plain_bn = all(m.training and m.affine and m.track_running_stats and m.momentum is not None and
                       m.momentum == self.node_classifiers[0][1].momentum and m.eps == self.node_classifiers[0][1].eps
                       for hd in self.node_classifiers for m in (hd[1], hd[5]))
        drops_on = all(m.training for hd in self.node_classifiers for m in (hd[3], hd[7]))
        return (self.num_output_channels == 4 and self.classifier_hidden_dim == 32 and self.node_embedding_dim == C
                and plain_bn and drops_on and os.environ.get("EG_STACKED_HEADS", "1") != "0")

    def _classifier_train(self, h: torch.Tensor, B: int, n: int, row_lo: int, n_valid: int) -> torch.Tensor:
        """models.py:363-377, :485-490 in train mode on the HIP kernels (_ClassifierTrainFn): the node-type filter is a row
        range, the four heads run as one stacked network.  Running statistics: the kernels update stacked copies, which are
        written back to the 8 BatchNorm modules with two multi-tensor copies."""
        heads = list(self.node_classifiers)
        bn1, bn2 = [hd[1] for hd in heads], [hd[5] for hd in heads]
        p1, p2 = float(heads[0][3].p), float(heads[0][7].p)
        seeds = torch.randint(0, 2 ** 62, (2,)).tolist() if (p1 > 0 or p2 > 0) else [0, 0]      # host RNG, like the layers
        with torch.no_grad():
            rm1, rv1 = torch.cat([b.running_mean for b in bn1]), torch.cat([b.running_var for b in bn1])
            rm2, rv2 = torch.cat([b.running_mean for b in bn2]), torch.cat([b.running_var for b in bn2])
        cfg = dict(running_mean1=rm1, running_var1=rv1, running_mean2=rm2, running_var2=rv2, eps1=bn1[0].eps, eps2=bn2[0].eps,
                   momentum1=bn1[0].momentum, momentum2=bn2[0].momentum, p1=p1, p2=p2, seed1=seeds[0], seed2=seeds[1])
        params = [getattr(hd[j], name) for hd in heads for j, name in _HEAD_PARAM_IDX]
        out = _ClassifierTrainFn.apply(h, B, n, row_lo, n_valid, self.output_activation == "sigmoid", cfg, *params)
        with torch.no_grad():
            torch._foreach_copy_([b.running_mean for b in bn1] + [b.running_var for b in bn1] +
                                 [b.running_mean for b in bn2] + [b.running_var for b in bn2],
                                 list(rm1.split(32)) + list(rv1.split(32)) + list(rm2.split(16)) + list(rv2.split(16)))
            torch._foreach_add_([b.num_batches_tracked for b in bn1 + bn2], 1)
        return out

    def _kidsum_buffers(self, graph, gb):
        key = (id(graph), gb)
        hit = self._kidsum.get(key)
        if hit is None or hit[0] is not graph:
            if len(self._kidsum) > 4:
                self._kidsum.clear()                  # (captured HIP graphs keep their own references, see below)
            hit = (graph, ops.new_kidsum(graph, gb), ops.new_kidsum(graph, gb))
            self._kidsum[key] = hit
        return hit[1], hit[2]

    def _jk_buffers(self, graph, gb, like):
        key = ("jk", id(graph), gb, tuple(like.shape))
        hit = self._kidsum.get(key)
        if hit is None or hit[0] is not graph:
            hit = (graph, torch.empty_like(like), torch.empty_like(like))
            self._kidsum[key] = hit
        return hit[1], hit[2]

    def _forward_nodes_graphed(self, node_feats, edge_index, B):
        key = (id(node_feats), node_feats.data_ptr(), tuple(node_feats.shape), id(edge_index), edge_index._version, B,
               tuple(_versions(l) for l in self.gnn_layers), tuple(_versions(c) for c in self.node_classifiers))
        hit = self._hip_graphs.get(key)
        if hit is not None and (hit[2] is not node_feats or hit[3] is not edge_index):
            hit = None
        if hit is None:
            was = self.use_hip_graph
            self.use_hip_graph = False
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                     # warm-up outside capture (allocations, caches)
                    self.forward_nodes(node_feats, edge_index, B)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    out, _ = self.forward_nodes(node_feats, edge_index, B)
            finally:
                self.use_hip_graph = was
            if len(self._hip_graphs) > 8:
                self._hip_graphs.clear()
            # everything the captured kernels point at stays alive with the entry: the input tensors, the graph handle,
            # the child-sum side buffers and the folded / packed parameters (their caches may evict independently)
            graph, gb = self._resolver.resolve(edge_index, node_feats.shape[0])
            keep = (graph, self._kidsum.get((id(graph), gb)), self._kidsum.get(("jk", id(graph), gb, tuple(node_feats.shape))),
                    self._fold_cache.get("layers"), self._fold_cache.get("cls"))
            hit = (g, out, node_feats, edge_index, keep)
            self._hip_graphs[key] = hit
        hit[0].replay()
        return hit[1]

    # ---- avg-pool node features (models.py:498-537): the step in front of the hot path ---------
    def create_node_pixels(self, echo_frames: torch.Tensor, num_samples_per_batch: int, node_coords=None):
        """models.py:498-537: average-pooled pyramid of the frame embedding + the frame itself, node-major."""
        B = int(num_samples_per_batch)
        maps = []
        if not self.use_main_graph_only:
            maps = [F.adaptive_avg_pool2d(echo_frames, output_size=(2 ** g, 2 ** g)) for g in range(1, self.num_aux_graphs + 1)]
        maps.append(echo_frames)
        conn = None
        if self.use_connection_nodes and not self.use_main_graph_only:
            conn = echo_frames.mean(dim=(2, 3)).unsqueeze(1).expand(B, self.num_aux_graphs + 1, C)
        return self.pack_node_features(maps, B, node_coords, conn)

    def pack_node_features(self, level_maps, num_samples_per_batch: int, node_coords=None, connection_embed=None):
        """The tail every create_node_pixels variant of the reference shares (models.py:511-537, :603-636, :726-756):
        NCHW level maps (coarse to fine, the last one is the frame-sized map) -> [B*N, 128] in the GNN's node order,
        in one packing launch (eg_pack_levels) instead of a per-sample permute / cat loop.  The UNet / CNN variants
        pass their own per-level feature maps and connection-node embeddings [B, naux+1, 128]."""
        B = int(num_samples_per_batch)
        n, n_conn, _, main_base, coord_base = self._row_ranges()
        fs = self.frame_size
        feats = ops.pack_levels([m.float() for m in level_maps], B, n, n_conn)
        if n_conn:
            feats = feats.clone() if feats.requires_grad else feats
            feats.view(B, n, C)[:, :n_conn, :] = connection_embed
        if self.use_coordinate_graph and not self.use_main_graph_only:
            new = ops.bilinear4(feats, node_coords.reshape(B, 4, 2).contiguous(), B, n, main_base, fs)
            feats = ops.scatter_coord_rows(feats, new, B, n, coord_base)
        return feats

    def pack_node_features_linear(self, features, linears, num_samples_per_batch: int, node_coords=None, connection_embed=None):
        """The UNet variant's whole tail (models.py:707-756): ``F.relu(self.linears[i](features[i]))`` for every level (1x1
        convolutions to 128 channels) AND the node-major packing in one launch (eg_conv1x1_relu_pack_levels); the 128-channel
        NCHW maps are never formed.  ``features``: decoder maps coarse to fine [B, C_l, p_l, p_l] (the last one frame-sized),
        ``linears``: the matching ``nn.Conv2d(C_l, 128, kernel_size=1)`` modules.  Connection-node embeddings
        [B, naux+1, 128] (means of the activated maps) come from the caller, as in ``pack_node_features``."""
        B = int(num_samples_per_batch)
        n, n_conn, _, main_base, coord_base = self._row_ranges()
        fs = self.frame_size
        feats = ops.conv1x1_relu_pack_levels([f.float() for f in features], [m.weight for m in linears],
                                             [m.bias for m in linears], B, n, n_conn)
        if n_conn:
            feats = feats.clone() if feats.requires_grad else feats
            feats.view(B, n, C)[:, :n_conn, :] = connection_embed
        if self.use_coordinate_graph and not self.use_main_graph_only:
            new = ops.bilinear4(feats, node_coords.reshape(B, 4, 2).contiguous(), B, n, main_base, fs)
            feats = ops.scatter_coord_rows(feats, new, B, n, coord_base)
        return feats

    def forward(self, data_batch=None, x=None, node_coords=None, edge_index=None, node_type=None, batch_idx=None):
        if data_batch is not None:
            x, edge_index, batch_idx, node_type = data_batch.x, data_batch.edge_index, data_batch.batch, \
                data_batch.node_type
            if self.use_coordinate_graph:
                node_coords = data_batch.node_coords
        # the reference reads B = batch_idx[-1] + 1 from the device (models.py:420); the frame
        # count is implied by the static topology, so no host sync is needed here.
        B = x.shape[0]
        nc = node_coords.reshape(B, 4, -1) if self.use_coordinate_graph else None
        node_feats = self.create_node_pixels(x, B, nc)
        return self.forward_nodes(node_feats, edge_index, B, node_coords)
