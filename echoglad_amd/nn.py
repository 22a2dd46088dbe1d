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
from .topology import TopologySpec, candidate_specs, commutative_edge_hash, get_topology, graph_type_variants

C = ops.C


# ---------------------------------------------------------------------------
# graph resolution: incoming PyG edge_index -> implicit topology or CSR handle
# ---------------------------------------------------------------------------
_TOPO_GRAPHS: Dict[tuple, ops.Graph] = {}          # (structured spec fields, device) -> handle, shared by every resolver
_EXPECTED_HASH: Dict[tuple, Tuple[int, int]] = {}  # (spec, batch) -> (E_dir, digest) of the closed form


def _topo_graph(spec: TopologySpec, device) -> ops.Graph:
    device = torch.device(device)
    diag_main = spec.main_graph_type == "grid-diagonal"
    diag_aux = spec.aux_graph_type == "grid-diagonal" and not spec.use_main_graph_only
    conn = spec.use_connection_nodes and not spec.use_main_graph_only
    key = (spec.frame_size, 0 if spec.use_main_graph_only else spec.num_aux_graphs, spec.use_main_graph_only,
           spec.use_coordinate_graph and not spec.use_main_graph_only, conn, diag_main, diag_aux, device)
    g = _TOPO_GRAPHS.get(key)
    if g is None:
        g = ops.Graph.topo(spec.frame_size, spec.num_aux_graphs, spec.use_main_graph_only, spec.use_coordinate_graph,
                           device=device, use_connection_nodes=conn, diag_main=diag_main, diag_aux=diag_aux)
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
            # the model's own static topology, with whichever graph types ('grid' / 'grid-diagonal' per level kind: dataset
            # configuration, not a constructor argument of the model) give this edge count
            out = []
            for spec in graph_type_variants(self.spec):
                topo = get_topology(spec)
                if topo.is_structured() and num_rows % topo.num_nodes == 0:
                    batch = num_rows // topo.num_nodes
                    if n_edges == batch * 2 * topo.num_undirected_edges:
                        out.append((spec, batch))
            return out
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


class Routes:
    """Switches of the fused routes whose FALLBACK must exist anyway -- module states and shapes the fused kernels do not cover
    (frozen sub-modules, hooks, JumpingKnowledge on irregular handles, ...) take it by themselves -- so that a test can run the
    fallback on the inputs of the fused route and compare.  Python attributes, deliberately NOT environment variables (rounds 3 - 5
    had one EG_* knob per route: ~20 untimed routes a user could land on by accident; the run-time variables that are left are
    listed in include/echoglad_hip.h)."""
    layer_sums_in_heads = True      # the last layer's BatchNorm-backward sums inside the heads' backward
    coord_mlp_kernel = True         # node_coordinate_mlp on eg_coord_mlp_* (off: the torch modules)
    coord_fused = True              # coordinate update folded into the consuming node (off: autograd nodes of its own)
    act_in_heads = True             # the last layer's activation pass inside the heads' first kernel
    train_chain = True              # child sums handed from layer to layer in the train forward
    jk_fused = True                 # JumpingKnowledge('max') as a running maximum inside the layer kernels
    stacked_heads = True            # the four heads as one stacked network in train mode
    heads_recompute_h = True        # the last layer's output is never written in full: the heads' backward rebuilds its tile (off: h is kept)
    heads_state_in_place = True     # the 4 heads' parameters and running statistics LIVE in the stacked arrays the kernels take (off: copied per step)
    chain_layers = True             # eval: child sums handed from layer to layer (default of HierarchicalPatchModel.chain_layers)
    fuse_classifier = True          # eval: the heads inside the last layer's kernel (default of HierarchicalPatchModel.fuse_classifier)


ROUTES = Routes()


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


# ---- BatchNorm-backward sums handed DOWN the stack (round 6) --------------------------------------------------------------------
# dx of layer i + 1 is dy of layer i: the dX launch takes layer i's sums where the rows leave it (ops.gcn_layer_bwd(lower=)), and
# the node of layer i picks them up here instead of running its own sums pass over dy and z.  Keyed by what the tensor IS -- address,
# shape, version -- never by a Python identity that autograd may not preserve: an entry is written right before the producing node
# returns dx and popped by the first backward that receives a tensor with that address AND version (autograd accumulating another
# gradient into dx bumps the version: the sums would be stale, the entry is ignored and the layer takes its own).  The model clears
# the table at every train-mode forward, so an entry never outlives the backward pass of its own step.
_SUMS_DOWN: Dict[tuple, tuple] = {}


def _sums_key(t: torch.Tensor) -> tuple:
    return (t.data_ptr(), tuple(t.shape), t.device, t._version)


def _hand_down(dx: torch.Tensor, sums, frames: int, row_hi: int, taps=None) -> None:
    if len(_SUMS_DOWN) > 8:
        _SUMS_DOWN.clear()
    _SUMS_DOWN[_sums_key(dx)] = (sums, frames, 0, row_hi, taps)


def _handed_down(dy: torch.Tensor):
    return _SUMS_DOWN.pop(_sums_key(dy), None) if _SUMS_DOWN else None


def _lower_of(box, dims_row_hi):
    """(z, bn, relu, p, seed, row_hi) of the layer below from the box its forward filled, or None."""
    if not box:
        return None
    z, bn, relu, p, seed = box
    return z, bn, relu, p, seed, dims_row_hi


def _bn_step(bn: nn.BatchNorm1d, pending: Optional[list] = None):
    """What nn.BatchNorm1d.forward decides before calling F.batch_norm: (use batch statistics?, update factor | None).
    Counts the batch in ``num_batches_tracked``; ``momentum=None`` is the cumulative moving average.  ``pending``: a list that
    collects the counters instead (the caller bumps them all with one multi-tensor add: a launch per BatchNorm otherwise)."""
    use_batch = bn.training or bn.running_mean is None
    factor = None
    if bn.training and bn.track_running_stats and bn.running_mean is not None:
        if pending is not None and bn.momentum is not None:
            pending.append(bn.num_batches_tracked)
        else:
            with torch.no_grad():
                bn.num_batches_tracked += 1
        factor = 1.0 / float(bn.num_batches_tracked) if bn.momentum is None else float(bn.momentum)
    return use_batch, factor


class _LayerTrainFn(torch.autograd.Function):
    """One whole train-mode layer as a single autograd node over the two C-ABI composites
    (eg_gcn_layer_train_fwd / eg_gcn_layer_bwd; models.py:328-335, :431-435):
        z = A_hat x W^T + b;  out = relu|id(dropout(BN_batch(z))) + x.
    Kept for the backward: z, the aggregated input A_hat x (so that dW = dz^T (A_hat x) needs no second aggregation) and the
    batch statistics -- not x.  ``kid`` = (kidsum_in | None, kidsum_out | None): child-sum side buffers of a chained train
    forward (the layer that produces x leaves the child sums of x behind, eg_gcn_layer_train_fwd)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, graph, batch, relu, p, momentum, eps, seed,
                residual, kid=(None, None), down=None):
        """down = (box this layer fills for the layer above | None, box of the layer below | None, row_hi): the hand-down of the
        BatchNorm-backward sums (_SUMS_DOWN)."""
        x = x.contiguous()
        need_w = weight.requires_grad
        out, z, agg, bn = ops.gcn_layer_train_fwd(graph, batch, x, weight.contiguous(), bias.contiguous(), gamma.contiguous(),
                                                  beta.contiguous(), running_mean, running_var, momentum, eps, relu, p, seed,
                                                  residual, want_agg=need_w, kidsum_in=kid[0], kidsum_out=kid[1])
        mine, below, row_hi = down if down is not None else (None, None, 0)
        if mine is not None:
            mine[:] = [z, bn, relu, p, seed]
        lower = _lower_of(below, row_hi)
        ctx.lower = None if lower is None else lower[2:]
        ctx.save_for_backward(z, agg if agg is not None else z.new_zeros(0), weight.detach().contiguous(),
                              gamma.detach().contiguous(), beta.detach().contiguous(), bn, *(lower[:2] if lower is not None else ()))
        ctx.cfg = (graph, batch, relu, p, seed, residual, need_w)
        return out

    @staticmethod
    def backward(ctx, dy):
        z, agg, weight, gamma, beta, bn, *lz = ctx.saved_tensors
        graph, batch, relu, p, seed, residual, had_agg = ctx.cfg
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and had_agg, ctx.needs_input_grad[2]
        given = _handed_down(dy)
        dy = dy.contiguous()
        if ctx.lower is not None and need_x and residual:
            dx, dw, db, dgamma, dbeta, lsums = ops.gcn_layer_bwd(graph.bwd, batch, dy, z, agg if had_agg else None, weight, gamma,
                                                                 beta, bn, relu, p, seed, residual, True, need_w, dy_sums=given,
                                                                 lower=(lz[0], lz[1]) + ctx.lower)
            _hand_down(dx, lsums, batch, ctx.lower[3])
        else:
            dx, dw, db, dgamma, dbeta = ops.gcn_layer_bwd(graph.bwd, batch, dy, z, agg if had_agg else None, weight, gamma,
                                                          beta, bn, relu, p, seed, residual, need_x, need_w, dy_sums=given)
        return (dx, dw, db if need_b else None, dgamma, dbeta) + (None,) * 12


# ---- the coordinate-graph update (models.py:438-473) as part of the node that CONSUMES the layer output -------------------------
# The update of layer i reads the layer's output h (its 4 coordinate rows per frame -> node_coordinate_mlp -> new landmark
# positions), samples h's main grid at the new positions and overwrites the coordinate rows with the samples.  Under autograd
# that is a scatter into a [B*N,128] tensor and, backwards, a patch of 4 + up to 16 rows per frame of a [B*N,128] gradient.  Done
# as nodes of their own those patches either cost dense copies / adds per layer (autograd sums two [B*N,128] gradients), or have
# to happen in place on a gradient buffer that autograd may have handed to somebody else as well.  Neither: the update is
# folded into the NEXT node (layer i + 1, or the classifier heads after the last layer), whose backward allocates the gradient
# it returns -- every in-place row patch happens on a buffer that node created itself, and the forward overwrites rows of a
# tensor that is the fresh output of the node before it (nothing else holds it; forward_nodes takes this route only when no
# hook could have seen it).
_MLP_NAMES = ("w1", "b1", "gamma1", "beta1", "w2", "b2", "gamma2", "beta2", "w3", "b3")


def _coord_update_fwd(h, coords_prev, dims, mlp_cfg, mlp_params, want_backward=True, sample=True):
    """h [B*N,128] (coordinate rows overwritten IN PLACE) -> (new coords [4B,2], state for _coord_update_bwd + the new coordinates once
    more, in a tensor of their own: `new` is saved for the backward, the other one is what the node hands out).
    sample=False: the coordinate rows are NOT resampled (after the last layer nobody reads them -- the heads drop the coordinate
    rows, models.py:485 -- and with h written sparsely the main grid the samples would come from does not exist)."""
    B, n, main_base, frame, coord_base = dims
    P = dict(mlp_cfg)
    P.update({k: p.detach().contiguous() for k, p in zip(_MLP_NAMES, mlp_params)})
    flat = coords_prev.reshape(B * 4, 2).contiguous()
    # the MLP reads the coordinate rows where they live (and leaves the packed copy the backward needs: the rows change below),
    # the samples are written straight into them: no gather / scatter launches around the two kernels
    (new, new_out), lm, saved = ops.coord_update_fwd(h, flat, B, n, coord_base, main_base, P, True, frame, want_backward, resample=sample)
    return new, (lm, flat, saved, {k: v for k, v in P.items() if not k.startswith("running")}, new_out)


def _coord_update_bwd(dx, dcoords_new, h, new, lm, flat, saved, P, dims, need_dprev, sampled_rows_used=True, lower=None):
    """dx: gradient w.r.t. the tensor AFTER the overwrite, a buffer the caller has just allocated; turned IN PLACE into the
    gradient w.r.t. the tensor BEFORE it.  -> (dcoords_prev [4B,2] | None, packed MLP gradients, taps | None).
    lower: dx is the dy of a layer whose BatchNorm-backward sums were taken before this call (ops.gcn_layer_bwd(lower=)): the sums
    of what the 16 taps per frame add are returned as taps [B,2,128]."""
    B, n, main_base, frame, coord_base = dims
    if sampled_rows_used:
        # the sampled rows' gradient is read where it lies (the coordinate rows of dx), 16 taps per frame go into dx's main-grid rows,
        # d lm is written into the coordinate rows (their old values were overwritten in the forward): one launch up to batch 16
        dprev, g, taps = ops.coord_update_bwd(dx, None if dcoords_new is None else dcoords_new.contiguous().view(B * 4, 2), h, new, lm, flat, B, n,
                                              coord_base, main_base, P, frame, saved, need_dprev, lower=lower)
        return dprev, g, taps
    total = dcoords_new
    if total is None:
        total = torch.zeros(B * 4, 2, dtype=torch.float32, device=dx.device)
    # d lm is ADDED to the coordinate rows: they fed the MLP and nothing else, and their samples were not used
    _, dprev, g = ops.coord_mlp_bwd(total.contiguous().view(B * 4, 2), lm, flat, B, P, frame, saved, True, need_dprev,
                                    out_rows=(dx, n, coord_base), accumulate=True)
    taps = None
    return dprev, g, taps


def _mlp_grads(g):
    sizes = (32 * 136, 32, 32, 32, 16 * 32, 16, 16, 16, 2 * 16, 2)
    parts = torch.split(g, sizes)
    return (parts[0].view(32, 136), parts[1], parts[2], parts[3], parts[4].view(16, 32), parts[5], parts[6], parts[7],
            parts[8].view(2, 16), parts[9])


class _CoordLayerTrainFn(torch.autograd.Function):
    """Coordinate update of layer i - 1 (on this node's input) + train-mode layer i:
        (h_prev [B*N,128], coords_prev [B,4,2]) -> (out, coords [B,4,2]).
    ``cfg`` = (graph, batch, relu, p, momentum, eps, seed, residual, dims, mlp_cfg, kid, down); params = the 10 tensors of
    node_coordinate_mlp[i - 1] in _HEAD_PARAM_IDX order.  ``down`` as in _LayerTrainFn."""

    @staticmethod
    def forward(ctx, h_prev, coords_prev, weight, bias, gamma, beta, running_mean, running_var, cfg, *mlp_params):
        graph, batch, relu, p, momentum, eps, seed, residual, dims, mlp_cfg, kid, down = cfg
        new, (lm, flat, saved, P, new_out) = _coord_update_fwd(h_prev, coords_prev, dims, mlp_cfg, mlp_params)
        need_w = weight.requires_grad
        out, z, agg, bn = ops.gcn_layer_train_fwd(graph, batch, h_prev, weight.contiguous(), bias.contiguous(), gamma.contiguous(),
                                                  beta.contiguous(), running_mean, running_var, momentum, eps, relu, p, seed,
                                                  residual, want_agg=need_w, kidsum_in=kid[0], kidsum_out=kid[1])
        mine, below, row_hi = down if down is not None else (None, None, 0)
        if mine is not None:
            mine[:] = [z, bn, relu, p, seed]
        lower = _lower_of(below, row_hi)
        ctx.lower = None if lower is None else lower[2:]
        ctx.n_lower = 0 if lower is None else 2
        ctx.save_for_backward(z, agg if agg is not None else z.new_zeros(0), weight.detach().contiguous(),
                              gamma.detach().contiguous(), beta.detach().contiguous(), bn, h_prev, new, lm, flat,
                              *(lower[:2] if lower is not None else ()), *saved)
        ctx.cfg = (graph, batch, relu, p, seed, residual, need_w, dims, P)
        return out, new_out.view(dims[0], 4, 2)          # (a tensor of its own: `new` is saved for the backward; dims[0] = frames; `batch` = copies of the handle's graph: 1 for a CSR of the whole batch)

    @staticmethod
    def backward(ctx, dy, dcoords):
        z, agg, weight, gamma, beta, bn, h_prev, new, lm, flat, *saved = ctx.saved_tensors
        lz, saved = saved[:ctx.n_lower], saved[ctx.n_lower:]
        graph, batch, relu, p, seed, residual, had_agg, dims, P = ctx.cfg
        need_w, need_b = ctx.needs_input_grad[2] and had_agg, ctx.needs_input_grad[3]
        given = _handed_down(dy)
        lower = (lz[0], lz[1]) + ctx.lower if (ctx.lower is not None and residual) else None
        res = ops.gcn_layer_bwd(graph.bwd, batch, dy.contiguous(), z, agg if had_agg else None, weight, gamma,
                                beta, bn, relu, p, seed, residual, True, need_w, dy_sums=given, lower=lower)
        dx, dw, db, dgamma, dbeta = res[:5]
        B = dims[0]
        dprev, g, taps = _coord_update_bwd(dx, None if dcoords is None else dcoords.reshape(B * 4, 2), h_prev, new, lm, flat,
                                           tuple(saved), P, dims, ctx.needs_input_grad[1], lower=lower)
        if lower is not None:
            _hand_down(dx, res[5], batch, lower[5], taps)
        return (dx, None if dprev is None else dprev.view(B, 4, 2), dw, db if need_b else None, dgamma, dbeta, None, None,
                None) + _mlp_grads(g)


_HEAD_PARAM_IDX = ((0, "weight"), (0, "bias"), (1, "weight"), (1, "bias"), (4, "weight"), (4, "bias"), (5, "weight"),
                   (5, "bias"), (8, "weight"), (8, "bias"))


def _seq_params(seq: nn.Sequential):
    """The 10 parameters of a Linear-BN-ReLU-Drop-Linear-BN-ReLU-Drop-Linear head in _HEAD_PARAM_IDX order (plain dict lookups:
    this runs for 7 heads several times per training step, and at batch 1 the step is bound by the host)."""
    m = seq._modules
    return [m[str(j)]._parameters[name] for j, name in _HEAD_PARAM_IDX]


class _ClassifierTrainFn(torch.autograd.Function):
    """models.py:363-377, :485-490 in train mode over eg_classifier_train_fwd / eg_classifier_bwd: node-type filter + the
    four heads as one stacked network (first layers one [128 -> 128] product, 4 x BatchNorm1d(32) == BatchNorm1d(128) on the
    stacked output; second layers block-diagonal [128 -> 64]; third a 16-wide dot).  ``params`` = for each head its 10
    parameters in _HEAD_PARAM_IDX order; gradients go back to each of them."""

    @staticmethod
    def forward(ctx, h, batch, n, row_lo, n_valid, sigmoid, cfg, *params):
        P = _stack_head_params(params, cfg)
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
        return (dh, None, None, None, None, None, None) + _unstack_head_grads(g)


# per-head sizes of the 10 parameters in _HEAD_PARAM_IDX order, and where the stacked arrays start in one flat buffer (the layout
# of eg_classifier_bwd's packed gradients, include/echoglad_hip.h)
_HEAD_SIZES = (32 * C, 32, 32, 32, 16 * 32, 16, 16, 16, 16, 1)
_HEAD_NAMES = ("w1", "b1", "gamma1", "beta1", "w2", "b2", "gamma2", "beta2", "w3", "b3")
_HEAD_SHAPES = ((4 * 32, C), (128,), (128,), (128,), (4, 16, 32), (64,), (64,), (64,), (64,), (4,))


def _head_param_offsets():
    """Element offset of parameter (head k, array j) = params[10 * k + j] in the flat buffer of _stack_head_params."""
    offs, start = [0] * 40, 0
    for j, size in enumerate(_HEAD_SIZES):
        for k in range(4):
            offs[10 * k + j] = start + k * size
        start += 4 * size
    return offs


def _views_of(bank: torch.Tensor, tensors, offsets) -> bool:
    """Is tensors[i] the contiguous float32 slice of ``bank`` that starts at element offsets[i]?"""
    base, dev = bank.data_ptr(), bank.device
    for t, o in zip(tensors, offsets):
        if t.data_ptr() != base + 4 * o or t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
            return False
    return True


def _move_into(bank: torch.Tensor, tensors, offsets, assign) -> None:
    """Copies tensors[i] into bank[offsets[i] : + numel] and re-points it there (assign(i, view)): once, not per step."""
    with torch.no_grad():
        for i, (t, o) in enumerate(zip(tensors, offsets)):
            view = bank[o:o + t.numel()].view(t.shape)
            view.copy_(t)
            assign(i, view)


def _stack_head_params(params, cfg):
    """The 4 x 10 head parameters as the stacked arrays the kernels take: ONE flat buffer filled by one multi-tensor copy (a
    torch.cat / stack per array was 10 launches per step), the arrays are views of it."""
    bank = cfg.get("_param_bank")
    if bank is not None and _views_of(bank, params, _head_param_offsets()):
        P, start = dict(cfg), 0                   # the parameters ARE the stacked arrays (HierarchicalPatchModel._heads_in_place): nothing to copy
        for size, name, shape in zip(_HEAD_SIZES, _HEAD_NAMES, _HEAD_SHAPES):
            P[name] = bank[start:start + 4 * size].view(shape)
            start += 4 * size
        return P
    flat = torch.empty(4 * sum(_HEAD_SIZES), dtype=torch.float32, device=params[0].device)
    dst, src, start = [], [], 0
    P = dict(cfg)
    for j, (size, name, shape) in enumerate(zip(_HEAD_SIZES, _HEAD_NAMES, _HEAD_SHAPES)):
        P[name] = flat[start:start + 4 * size].view(shape)
        for k in range(4):
            dst.append(flat[start + k * size:start + (k + 1) * size])
            src.append(params[10 * k + j].detach().reshape(-1))
        start += 4 * size
    torch._foreach_copy_(dst, src)
    return P


def _unstack_head_grads(g):
    dw1, db1, dg1, dbe1 = g[:16384].view(4, 32, C), g[16384:16512].view(4, 32), g[16512:16640].view(4, 32), g[16640:16768].view(4, 32)
    o = 16768
    dw2 = g[o:o + 2048].view(4, 16, 32)
    db2, dg2, dbe2 = (g[o + 2048 + 64 * k:o + 2048 + 64 * (k + 1)].view(4, 16) for k in range(3))
    dw3 = g[o + 2240:o + 2304].view(4, 1, 16)
    db3 = g[o + 2304:o + 2308].view(4, 1)
    grads = []
    for k in range(4):
        grads += [dw1[k], db1[k], dg1[k], dbe1[k], dw2[k], db2[k], dg2[k], dbe2[k], dw3[k], db3[k]]
    return tuple(grads)


class _CoordClassifierTrainFn(torch.autograd.Function):
    """Coordinate update of the LAST layer (on this node's input) + node-type filter + the four heads in train mode:
        (h [B*N,128], coords_prev [B,4,2]) -> (logits, coords [B,4,2]).
    The heads drop the coordinate rows (models.py:485), so the rows the update samples into h feed nothing: backwards only the
    MLP's gradient enters the coordinate rows (no bilinear backward).  params = 10 MLP tensors + 4 x 10 head tensors."""

    @staticmethod
    def forward(ctx, h, coords_prev, dims5, sigmoid, cls_cfg, coord_dims, mlp_cfg, *params):
        batch, n, row_lo, n_valid = dims5
        new, (lm, flat, saved, Pm, new_out) = _coord_update_fwd(h, coords_prev, coord_dims, mlp_cfg, params[:10])
        P = _stack_head_params(params[10:], cls_cfg)
        logits, z1, z2, bn = ops.classifier_train_fwd(h, batch, n, row_lo, n_valid, P, sigmoid)
        ctx.P = {k: v for k, v in P.items() if not k.startswith("running")}
        ctx.cfg = (batch, n, row_lo, n_valid, sigmoid, coord_dims, Pm)
        ctx.save_for_backward(h, z1, z2, bn, logits if sigmoid else logits.new_zeros(0), new, lm, flat, *saved)
        return logits, new_out.view(batch, 4, 2)

    @staticmethod
    def backward(ctx, dy, dcoords):
        h, z1, z2, bn, y, new, lm, flat, *saved = ctx.saved_tensors
        batch, n, row_lo, n_valid, sigmoid, coord_dims, Pm = ctx.cfg
        if dy is None:
            dy = torch.zeros(batch * n_valid, 4, dtype=torch.float32, device=h.device)
        dl = dy.contiguous()
        if sigmoid:
            dl = dl * y * (1.0 - y)
        dh, g = ops.classifier_bwd(dl, h, batch, n, row_lo, n_valid, ctx.P, z1, z2, bn, True)
        dprev, gm, _ = _coord_update_bwd(dh, None if dcoords is None else dcoords.reshape(batch * 4, 2), h, new, lm, flat, tuple(saved),
                                         Pm, coord_dims, ctx.needs_input_grad[1], sampled_rows_used=False)
        return (dh, None if dprev is None else dprev.view(batch, 4, 2), None, None, None, None, None) + _mlp_grads(gm) + \
            _unstack_head_grads(g)


class _LastLayerHeadsTrainFn(torch.autograd.Function):
    """The LAST train-mode layer and the classifier heads as one node, so that the layer's activation pass runs inside the heads'
    first kernel (eg_classifier_train_fwd_act: h is written once and never read back for the heads' first product):
        (h_prev [B*N,128], coords_prev [B,4,2] | None) -> (logits, coords [B,4,2] | None).
    With the coordinate graph the node also carries the update in front of the layer (of layer L - 2, on h_prev; absent when
    L = 1) and the one behind it (of layer L - 1, on h), exactly as _CoordLayerTrainFn / _CoordClassifierTrainFn do.
    cfg = (graph, batch, relu, p, momentum, eps, seed, residual, kid_in, dims5, sigmoid, cls_cfg, coord_dims | None,
           mlp_prev_cfg | None, mlp_cfg | None[, down]); params = [10 tensors of the MLP in front] + [10 of the MLP behind] + 40 head
    tensors.  ``down`` = (None, box of the layer below | None, row_hi) as in _LayerTrainFn."""

    @staticmethod
    def forward(ctx, h_prev, coords_prev, weight, bias, gamma, beta, running_mean, running_var, cfg, *params):
        down = cfg[15] if len(cfg) > 15 else None
        (graph, batch, relu, p, momentum, eps, seed, residual, kid_in, dims5, sigmoid, cls_cfg, cdims, mlp_prev_cfg, mlp_cfg) = cfg[:15]
        has_coord, has_prev = mlp_cfg is not None, mlp_prev_cfg is not None
        k0 = 10 if has_prev else 0
        k1 = k0 + (10 if has_coord else 0)
        h_prev = h_prev.contiguous()
        coords_mid, st_prev = coords_prev, None
        if has_prev:
            coords_mid, st_prev = _coord_update_fwd(h_prev, coords_prev, cdims, mlp_prev_cfg, params[:10])
        need_w = weight.requires_grad
        _, z, agg, bn = ops.gcn_layer_train_fwd(graph, batch, h_prev, weight.contiguous(), bias.contiguous(), gamma.contiguous(),
                                                beta.contiguous(), running_mean, running_var, momentum, eps, relu, p, seed,
                                                residual, want_agg=need_w, kidsum_in=kid_in, want_out=False)
        B, n, row_lo, n_valid = dims5
        P = _stack_head_params(params[k1:], cls_cfg)
        # h = act(z) + h_prev feeds the heads' first product (inside this kernel) and, backwards, dW1 = dz1^T h: where the backward takes
        # the layer's sums in the heads' kernel it holds z anyway and rebuilds its h tile from z and h_prev, so h is never written in
        # full (1.18 GB per step at batch 32) -- only the coordinate rows the landmark MLP reads are
        sparse = bool(ROUTES.heads_recompute_h and ROUTES.layer_sums_in_heads and ops.classifier_recompute_h_supported(B, n, n_valid))
        h, logits, z1, z2, cbn = ops.classifier_train_fwd_act(z, bn, h_prev if residual else None, relu, p, seed, B, n, row_lo,
                                                              n_valid, P, sigmoid, h_sparse=sparse)
        ctx.h_sparse = sparse
        new, st = None, None
        if has_coord:
            new, st = _coord_update_fwd(h, coords_mid, cdims, mlp_cfg, params[k0:k1], sample=not sparse)
        ctx.P = {k: v for k, v in P.items() if not k.startswith("running")}
        ctx.cfg = (graph, batch, relu, p, seed, residual, need_w, dims5, sigmoid, cdims, has_coord, has_prev,
                   st[3] if has_coord else None, st_prev[3] if has_prev else None)
        saved = [z, agg if agg is not None else z.new_zeros(0), weight.detach().contiguous(), gamma.detach().contiguous(),
                 beta.detach().contiguous(), bn, h_prev, h, z1, z2, cbn, logits if sigmoid else logits.new_zeros(0)]
        if has_coord:
            saved += [new, st[0], st[1], *st[2]]
        if has_prev:
            saved += [coords_mid, st_prev[0], st_prev[1], *st_prev[2]]
        ctx.n_saved2 = len(st[2]) if has_coord else 0
        lower = _lower_of(down[1], down[2]) if down is not None else None
        ctx.lower = None if lower is None else lower[2:]
        if lower is not None:
            saved += [lower[0], lower[1]]                 # (at the END: z and bn of the layer below)
        ctx.save_for_backward(*saved)
        return logits, (st[4].view(B, 4, 2) if has_coord else None)

    @staticmethod
    def backward(ctx, dy, dcoords):
        (graph, batch, relu, p, seed, residual, had_agg, dims5, sigmoid, cdims, has_coord, has_prev, Pm, Pm_prev) = ctx.cfg
        z, agg, weight, gamma, beta, bn, h_prev, h, z1, z2, cbn, y, *rest = ctx.saved_tensors
        lower = None
        if ctx.lower is not None:
            lower, rest = (rest[-2], rest[-1]) + ctx.lower, rest[:-2]
        B, n, row_lo, n_valid = dims5
        if dy is None:
            dy = torch.zeros(B * n_valid, 4, dtype=torch.float32, device=h.device)
        dl = dy.contiguous()
        if sigmoid:
            dl = dl * y * (1.0 - y)
        # the layer's BatchNorm-backward sums over the heads' rows are taken where dh leaves the heads' backward (no separate
        # sums pass over dh and z for the layer afterwards: only the rows the filter drops are added there)
        presum = None
        if ctx.h_sparse or (ops.classifier_layer_sums_supported(B, n, n_valid) and ROUTES.layer_sums_in_heads):
            dh, g, sums = ops.classifier_bwd(dl, h, B, n, row_lo, n_valid, ctx.P, z1, z2, cbn, True,
                                             layer=(z, bn, gamma, beta, relu, p, seed),
                                             recompute=((h_prev if residual else None),) if ctx.h_sparse else False)
            presum = None if sums is None else (sums, B, row_lo, n_valid)
        else:
            dh, g = ops.classifier_bwd(dl, h, B, n, row_lo, n_valid, ctx.P, z1, z2, cbn, True)     # (a buffer of this node)
        dmid, gm, gm_prev = None, None, None
        if has_coord:
            new, lm, flat = rest[0], rest[1], rest[2]
            saved2 = tuple(rest[3:3 + ctx.n_saved2])
            rest = rest[3 + ctx.n_saved2:]
            dmid, gm, _ = _coord_update_bwd(dh, None if dcoords is None else dcoords.reshape(B * 4, 2), h, new, lm, flat, saved2, Pm,
                                            cdims, has_prev or ctx.needs_input_grad[1], sampled_rows_used=False)
        need_x = has_prev or ctx.needs_input_grad[0]
        need_w, need_b = ctx.needs_input_grad[2] and had_agg, ctx.needs_input_grad[3]
        if not (need_x and residual):
            lower = None
        res = ops.gcn_layer_bwd(graph.bwd, batch, dh, z, agg if had_agg else None, weight, gamma, beta, bn,
                                relu, p, seed, residual, need_x, need_w, dy_sums=presum, lower=lower)
        dx, dw, db, dgamma, dbeta = res[:5]
        dprev = dmid
        taps = None
        if has_prev:
            new1, lm1, flat1 = rest[0], rest[1], rest[2]
            dprev, gm_prev, taps = _coord_update_bwd(dx, dmid, h_prev, new1, lm1, flat1, tuple(rest[3:]), Pm_prev, cdims,
                                                     ctx.needs_input_grad[1], lower=lower)
        if lower is not None:
            _hand_down(dx, res[5], batch, lower[5], taps)
        out = (dx if ctx.needs_input_grad[0] else None, None if dprev is None else dprev.view(B, 4, 2), dw,
               db if need_b else None, dgamma, dbeta, None, None, None)
        if has_prev:
            out += _mlp_grads(gm_prev)
        if has_coord:
            out += _mlp_grads(gm)
        return out + _unstack_head_grads(g)


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
        return (dlm, dc.view(batch, 4, 2) if dc is not None else None, None, None, None) + _mlp_grads(g)


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
    ``forward(x, edge_index) -> x``.  The kernels are built for 128 -> 128 (default.yml:13-14); narrower layers -- the
    reference's signature defaults are 128 -> 64 -> 64 (models.py:286-301) -- run on the same kernels zero-padded to 128
    channels (a compatibility route: correct, differentiable, and half of its bytes are padding)."""

    def __init__(self, in_channels: int, out_channels: int, **kwargs):
        super().__init__()
        if not (1 <= in_channels <= C and 1 <= out_channels <= C):
            raise NotImplementedError(f"the HIP GCNConv is built for up to {C} channels, got {in_channels}->{out_channels}")
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
        if self.in_channels == C and self.out_channels == C:
            return _GCNConvFn.apply(x, self.lin.weight, self.bias, graph, batch)
        # narrower than the kernels' 128 channels: zero-padded input columns / weight rows and columns / bias, sliced output
        # (autograd runs through the pads and the slice, so the gradients of the real entries are the kernels' own)
        xp = F.pad(x, (0, C - self.in_channels))
        wp = F.pad(self.lin.weight, (0, C - self.in_channels, 0, C - self.out_channels))
        bp = F.pad(self.bias, (0, C - self.out_channels))
        return _GCNConvFn.apply(xp, wp, bp, graph, batch)[:, :self.out_channels]


class Sequential(nn.Module):
    """Counterpart of ``torch_geometric.nn.Sequential('x, edge_index', [(conv, 'x, edge_index -> x'), m, ...])``
    (src/core/models.py:329-335): children are registered as ``module_{i}``.

    The reference's layer -- ``[GCNConv, BatchNorm1d(128), Dropout, ReLU | Identity]`` -- is recognised and runs as ONE
    kernel launch through this module's own ``forward(x, edge_index)``, so the reference's ``models.py`` loop
    (``self.gnn_layers[i](hidden_embeds[i], edge_index)``, :431) reaches the fused kernels unchanged:
      * eval mode, no gradient wanted: ``eg_gcn_layer_fwd`` with bias + BatchNorm folded into scale / shift and the ReLU in
        the kernel's epilogue (three ``[B*N,128]`` elementwise passes fewer per layer);
      * train mode (BatchNorm on batch statistics, Dropout on): the ``_LayerTrainFn`` composite
        (``eg_gcn_layer_train_fwd`` / ``eg_gcn_layer_bwd``) without a residual -- that stays the caller's
        ``h + hidden_embeds[i]`` (:434-435);
      * anything else (frozen sub-modules inside a training model, eval with gradients, forward hooks on a child, other
        module lists): module by module, as before.  ``EG_SEQ_FUSED=0`` forces that route."""

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
        self._fold: Optional[tuple] = None

    def __len__(self):
        return len(self._takes_graph)

    def __getitem__(self, i):
        return getattr(self, f"module_{i}")

    # ---- the reference's layer as one launch ------------------------------------------------------------------------
    def _reference_layer(self):
        """(conv, bn, dropout, relu?) when the children are exactly models.py:329-335's list and none of them is observed
        through a hook (a hook must see the intermediate tensor it was registered for), else None."""
        if self._takes_graph != [True, False, False, False] or os.environ.get("EG_SEQ_FUSED", "1") == "0":
            return None
        conv, bn, drop, act = self.module_0, self.module_1, self.module_2, self.module_3
        if type(conv) is not GCNConv or type(bn) is not nn.BatchNorm1d or bn.num_features != C or type(drop) is not nn.Dropout \
                or type(act) not in (nn.ReLU, nn.Identity):
            return None
        for m in (conv, bn, drop, act):
            if m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks:
                return None
        return conv, bn, drop, type(act) is nn.ReLU

    def _folded(self, conv, bn):
        key = _versions(self)
        if self._fold is None or self._fold[0] != key:
            with torch.no_grad():
                gamma = bn.weight if bn.affine else torch.ones_like(bn.running_var)
                scale = gamma / torch.sqrt(bn.running_var + bn.eps)
                shift = (bn.bias if bn.affine else 0.0) - bn.running_mean * scale
                if conv.bias is not None:
                    shift = shift + conv.bias * scale
                self._fold = (key, conv.lin.weight.detach().contiguous(), scale.contiguous(), shift.contiguous())
        return self._fold[1:]

    def _fused(self, x, graph: ops.Graph, batch: int):
        """The layer as one launch, or None when this call is not one of the two fused cases."""
        ref = self._reference_layer()
        if ref is None or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] != C:
            return None
        conv, bn, drop, relu = ref
        wants_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not bn.training and bn.running_mean is not None and not wants_grad and not (drop.training and drop.p > 0):
            w, scale, shift = self._folded(conv, bn)
            return ops.gcn_layer_fwd(graph, batch, x.contiguous(), w, scale, shift, None, relu)
        if bn.training and bn.affine and drop.training and torch.is_grad_enabled():
            p = float(drop.p)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0          # host RNG, like the model's own route
            _, momentum = _bn_step(bn)
            return _LayerTrainFn.apply(x, conv.lin.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                       graph, batch, relu, p, momentum, bn.eps, seed, False, (None, None), None)
        return None

    def forward(self, x, edge_index):
        if self._reference_layer() is not None:
            graph, batch = _SHARED_RESOLVER.resolve(edge_index, x.shape[0])
            out = self._fused(x, graph, batch)
            if out is not None:
                return out
        for i, takes in enumerate(self._takes_graph):
            m = getattr(self, f"module_{i}")
            x = m(x, edge_index) if takes else m(x)
        return x

    def forward_graph(self, x, graph: ops.Graph, batch: int):
        out = self._fused(x, graph, batch)
        if out is not None:
            return out
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
        if node_embedding_dim != C:
            raise NotImplementedError(f"the HIP path is built for node_embedding_dim = {C} (node-feature packing, models.py:498-537)")
        if not (1 <= node_hidden_dim <= C):
            raise NotImplementedError(f"node_hidden_dim must be in [1, {C}]")
        # Widths other than configs/default.yml's 128 / 32 -- the reference's signature defaults are node_hidden_dim = 64,
        # classifier_hidden_dim = 16 (models.py:286-301) -- take a COMPATIBILITY route: GCNConv on the 128-channel kernels with
        # zero padding, BatchNorm / Dropout / activation / heads as the torch modules they are; none of the fused kernels.
        # With the coordinate graph on that route the landmark MLP (Linear(hidden + 8, cls_hidden) ...) runs as its torch modules and the
        # coordinate rows are resampled by eg_bilinear4_* from the node rows zero-padded to 128 channels.
        self._narrow = node_hidden_dim != C or classifier_hidden_dim != 32
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
        self._static_feats: Dict[tuple, torch.Tensor] = {}
        self.hip_graph_captures = 0
        self.use_hip_graph = False
        # eval path: layer i leaves the child sums of its output in a side buffer for layer i+1
        # (eg_gcn_layer_fwd_chain); False runs every layer on its own
        self.chain_layers = bool(ROUTES.chain_layers)
        # ... and the last layer runs the classifier heads on its output tile inside the kernel (False: separate)
        self.fuse_classifier = bool(ROUTES.fuse_classifier)
        self._kidsum: Dict[tuple, tuple] = {}
        # optional callable (layer index, layer output incl. residual and coordinate rows) -> None, called by forward_nodes
        self.layer_output_hook = None
        # optional callable (kind, module, seeds) -> None, called whenever a train-mode forward draws the seeds of a Dropout site:
        # ("gnn", gnn_layers[i], (seed,)), ("coord_mlp", node_coordinate_mlp[i], (seed1, seed2)), ("heads", node_classifiers,
        # (seed1, seed2)).  The mask of a site is a pure function of (seed, element index) (csrc/train_common.h): a test
        # regenerates the kernels' masks from these seeds and injects them into the oracle
        self.dropout_seed_hook = None

    def enable_hip_graph(self, flag: bool = True) -> "HierarchicalPatchModel":
        """Inference only: capture the kernel sequence of ``forward_nodes`` (3 fused layers + classifier
        + queue resets) into a HIP graph and replay it on later calls.

        Through ``forward()`` -- ``model(x=frames, edge_index=...)`` / ``model(data_batch)``, the calls engine.py:251-255,
        :394-398 make -- the node-feature packing in front of the stack (``pack_node_features`` / ``_linear``) writes into ONE static
        ``[B*N,128]`` buffer per (batch size, device), so every new batch of frames replays the SAME captured graph: the key
        is (that buffer's address, B, the resolved graph handle, parameter versions), never the identity of an input tensor.
        ``forward_nodes`` on a caller-owned buffer is captured once per buffer address (the buffer is kept alive by the entry).
        The returned logits tensor -- and, through ``forward()``, the node features -- are owned by the model and overwritten by
        the next call.  ``hip_graph_captures`` counts captures (a test asserts 1 over many batches)."""
        self.use_hip_graph = bool(flag)
        self._hip_graphs.clear()
        self._static_feats.clear()
        return self

    def _static_node_feats(self, B: int, inputs) -> Optional[torch.Tensor]:
        """The static node-feature buffer ``forward()``'s packing writes into when the call will take the replayed route
        (eval, nothing wants a gradient, no coordinate graph / narrow widths / hook), else None."""
        if not self.use_hip_graph or self.training or self._narrow or self.use_coordinate_graph or self.layer_output_hook is not None:
            return None
        if torch.is_grad_enabled() and (any(t is not None and t.requires_grad for t in inputs) or
                                        any(p.requires_grad for p in self.parameters())):
            return None
        if torch.cuda.is_current_stream_capturing():
            return None
        dev = inputs[0].device
        if dev.type != "cuda":
            return None
        n = self._row_ranges()[0]
        key = (B, n, dev)
        buf = self._static_feats.get(key)
        if buf is None:
            if len(self._static_feats) >= 4:               # (entries of _hip_graphs keep the buffers their graphs read alive)
                self._static_feats.clear()
            buf = torch.empty(B * n, C, dtype=torch.float32, device=dev)
            self._static_feats[key] = buf
        return buf

    # ---- static row ranges (replace the reference's node_type host syncs, models.py:447,456,473,485)
    def _row_ranges(self):
        topo = get_topology(self.topology_spec)
        return topo.num_nodes, topo.n_conn, topo.num_valid_nodes, topo.main.base, topo.coord_base

    def train(self, mode: bool = True):
        """nn.Module.train + a fresh start for everything cached on in-place version counters (folded inference parameters, captured
        inference graphs): a training step replayed from a HIP graph (engine.GraphedTrainStep) runs no host code, so the running
        statistics it updates on the device bump no counter -- the switch to eval() in front of an evaluation is where that shows."""
        if bool(mode) != self.training:                # (a change of mode only: eval() in front of every batch keeps its graphs)
            self.__dict__.get("_fold_cache", {}).clear()
            graphs = self.__dict__.get("_hip_graphs")
            if graphs:
                graphs.clear()
        return super().train(mode)

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
    def _layer_cfg(self, i: int):
        """(conv, bn, relu?, dropout p, seed) of layer i, or None when its BatchNorm / Dropout are frozen inside a training model."""
        layer = self.gnn_layers[i]
        conv, bn, drop = layer.module_0, layer.module_1, layer.module_2
        if not (bn.training and bn.affine and drop.training):
            return None
        p = float(drop.p)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0      # host RNG: reproducible under torch.manual_seed
        if self.dropout_seed_hook is not None:
            self.dropout_seed_hook("gnn", layer, (seed,))
        return conv, bn, i < self.num_gnn_layers - 1, p, seed

    def _layer_train(self, i: int, x_in: torch.Tensor, graph: ops.Graph, gb: int, kid=(None, None), down=None):
        cfg = self._layer_cfg(i)
        if cfg is None:
            # a frozen (eval-mode) BatchNorm / Dropout inside a training model: GCNConv kernel + the torch modules
            h = self.gnn_layers[i].forward_graph(x_in, graph, gb)
            return h + x_in if self.residual else h
        conv, bn, relu, p, seed = cfg
        _, momentum = _bn_step(bn)
        return _LayerTrainFn.apply(x_in, conv.lin.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   graph, gb, relu, p, momentum, bn.eps, seed, bool(self.residual), kid, down)

    def _sums_down_boxes(self, graph: ops.Graph):
        """One box per layer for the hand-down of the BatchNorm-backward sums (layer i's forward fills box i, the node of layer
        i + 1 reads it and its dX launch takes layer i's sums: _SUMS_DOWN), or None where that launch is not the producer /
        consumer kernel's.  EG_SUMS_DOWN=0: every layer takes its own sums (the round-5 step; A/B and fallback)."""
        _SUMS_DOWN.clear()
        if not (self.residual and ops.lower_sums_supported(graph.bwd) and os.environ.get("EG_SUMS_DOWN", "1") != "0"):
            return None
        return [[] for _ in range(self.num_gnn_layers)]

    # ---- coordinate-graph update (models.py:438-473), explicit form ------------------------------------------------------
    def _coordinate_update(self, i: int, h: torch.Tensor, node_coords: torch.Tensor, batch: int):
        """The update as autograd nodes of its own (eval mode; train mode with hooks, JumpingKnowledge or frozen sub-modules):
        nothing is modified in place under autograd -- the overwrite of the coordinate rows copies h.  The training step's
        route folds the update into the node that consumes h instead (_CoordLayerTrainFn, _CoordClassifierTrainFn)."""
        n, _, _, main_base, coord_base = self._row_ranges()
        fs = self.frame_size
        hd = h.shape[1]
        if hd != C:
            # the compatibility route (node_hidden_dim < 128): the torch modules on the landmark rows, the 4-tap sample on zero-padded rows
            lm = h.view(batch, n, hd)[:, coord_base:, :].reshape(batch * 4, hd)
            shape_feats = (node_coords.unsqueeze(1) - node_coords.unsqueeze(2)).reshape(batch * 4, 8)
            delta = self.node_coordinate_mlp[i](torch.cat((lm, shape_feats), dim=1))
            node_coords = torch.clamp(node_coords + delta.view(batch, 4, 2), min=0, max=fs - 1)
            new_feats = ops.bilinear4(F.pad(h, (0, C - hd)), node_coords, batch, n, main_base, fs)[:, :hd]
            out = h.clone()
            out.view(batch, n, hd)[:, coord_base:coord_base + 4, :] = new_feats.reshape(batch, 4, hd)
            return out, node_coords
        # pairwise (other - self) offsets per frame, flattened to 8 numbers per landmark (:441-444)
        lm = h.view(batch, n, C)[:, coord_base:, :].reshape(batch * 4, C).clone()
        new_coords = self._coord_mlp_kernel(self.node_coordinate_mlp[i], lm, node_coords, batch, fs)
        if new_coords is None:
            # a mix of frozen and training sub-modules, or eval mode with gradients: the torch modules, op by op
            shape_feats = (node_coords.unsqueeze(1) - node_coords.unsqueeze(2)).reshape(batch * 4, 8)
            delta = self.node_coordinate_mlp[i](torch.cat((lm, shape_feats), dim=1))
            new_coords = torch.clamp(node_coords + delta.view(batch, 4, 2), min=0, max=fs - 1)
        node_coords = new_coords
        new_feats = ops.bilinear4(h, node_coords, batch, n, main_base, fs)            # [4B, 128]
        h = ops.scatter_coord_rows(h, new_feats, batch, n, coord_base)
        return h, node_coords

    def _coord_mlp_cfg(self, mlp: nn.Sequential):
        """(cfg, params) of node_coordinate_mlp[i] for the kernels when every sub-module is in plain train state, else None."""
        if self._narrow or self.node_embedding_dim != C or not ROUTES.coord_mlp_kernel:
            return None
        m = mlp._modules                       # (nn.Sequential.__getitem__ walks an islice: ~170 of them per step were 0.2 ms of host time)
        bn1, bn2, d1, d2 = m["1"], m["5"], m["3"], m["7"]
        if not (bn1.affine and bn2.affine and bn1.training and bn2.training and d1.training and d2.training):
            return None
        params = _seq_params(mlp)
        cfg = dict(eps1=bn1.eps, eps2=bn2.eps, p1=float(d1.p), p2=float(d2.p), seed1=0, seed2=0,
                   running_mean1=bn1.running_mean, running_var1=bn1.running_var, running_mean2=bn2.running_mean,
                   running_var2=bn2.running_var)
        return cfg, params

    def _coord_mlp_train_cfg(self, mlp: nn.Sequential, pending: Optional[list] = None):
        """The same with this step's dropout seeds drawn and the BatchNorm batches counted (call once per forward)."""
        cfg, params = self._coord_mlp_cfg(mlp)
        if cfg["p1"] > 0 or cfg["p2"] > 0:
            cfg["seed1"], cfg["seed2"] = torch.randint(0, 2 ** 62, (2,)).tolist()     # host RNG, like the layers
        if self.dropout_seed_hook is not None:
            self.dropout_seed_hook("coord_mlp", mlp, (cfg["seed1"], cfg["seed2"]))
        _, cfg["momentum1"] = _bn_step(mlp[1], pending)
        _, cfg["momentum2"] = _bn_step(mlp[5], pending)
        return cfg, params

    def _coord_mlp_kernel(self, mlp: nn.Sequential, lm: torch.Tensor, node_coords: torch.Tensor, batch: int, frame: int):
        """models.py:441-453 on eg_coord_mlp_fwd / _bwd (one launch each way) -> new coords [B,4,2], or None when the
        module states are not ones the kernel implements."""
        if self._narrow or self.node_embedding_dim != C or node_coords.shape[-1] != 2 or node_coords.dtype != torch.float32 or \
                not ROUTES.coord_mlp_kernel:
            return None
        bn1, bn2, d1, d2 = mlp[1], mlp[5], mlp[3], mlp[7]
        if not (bn1.affine and bn2.affine):
            return None
        if self._coord_mlp_cfg(mlp) is not None:
            cfg, params = self._coord_mlp_train_cfg(mlp)
            return _CoordMlpFn.apply(lm, node_coords, batch, frame, cfg, *params)
        params = _seq_params(mlp)
        cfg = dict(eps1=bn1.eps, eps2=bn2.eps, p1=float(d1.p), p2=float(d2.p), seed1=0, seed2=0,
                   running_mean1=bn1.running_mean, running_var1=bn1.running_var, running_mean2=bn2.running_mean,
                   running_var2=bn2.running_var)
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

    # ---- the training step's route with the coordinate graph on: every update folded into the consuming node -----------------
    def _train_coord_fused_ok(self, node_coords) -> bool:
        if not (self.training and self.use_coordinate_graph and self.layer_output_hook is None and self.jk is None):
            return False
        if node_coords is None or node_coords.shape[-1] != 2 or node_coords.dtype != torch.float32:
            return False
        if not ROUTES.coord_fused or not self._stacked_heads_ok():
            return False
        for l in self.gnn_layers:
            if not (l.module_1.training and l.module_1.affine and l.module_2.training):
                return False
        return all(self._coord_mlp_cfg(m) is not None for m in self.node_coordinate_mlp)

    def _forward_train_coord_fused(self, x0: torch.Tensor, graph: ops.Graph, gb: int, B: int, node_coords: torch.Tensor):
        n, n_conn, n_valid, main_base, coord_base = self._row_ranges()
        dims = (B, n, main_base, self.frame_size, coord_base)
        kids = self._train_kidsums(graph, gb)
        h, coords = x0.contiguous(), node_coords
        L = self.num_gnn_layers
        boxes = self._sums_down_boxes(graph)
        down_of = (lambda i: None) if boxes is None else (lambda i: (boxes[i] if i < L - 1 else None, boxes[i - 1] if i > 0 else None, coord_base))
        counters = []                      # num_batches_tracked of every BatchNorm of the step: bumped together by finish(counters)
        act_in_heads = bool(ROUTES.act_in_heads)      # the last layer's activation pass inside the heads' first kernel
        for i in range(L):
            conv, bn, relu, p, seed = self._layer_cfg(i)
            _, momentum = _bn_step(bn, counters)
            kid = (kids[(i + 1) & 1] if i > 0 else None, kids[i & 1] if i < self.num_gnn_layers - 1 else None)
            if i == L - 1 and act_in_heads:
                mlp_prev_cfg, mlp_prev = self._coord_mlp_train_cfg(self.node_coordinate_mlp[i - 1], counters) if i > 0 else (None, [])
                mlp_cfg, mlp_params = self._coord_mlp_train_cfg(self.node_coordinate_mlp[i], counters)
                cls_cfg, head_params, finish = self._classifier_train_cfg()
                cfg = (graph, gb, relu, p, momentum, bn.eps, seed, bool(self.residual), kid[0], (B, n, n_conn, n_valid),
                       self.output_activation == "sigmoid", cls_cfg, dims, mlp_prev_cfg, mlp_cfg, down_of(i))
                logits, coords = _LastLayerHeadsTrainFn.apply(h, coords, conv.lin.weight, conv.bias, bn.weight, bn.bias,
                                                              bn.running_mean, bn.running_var, cfg, *mlp_prev, *mlp_params,
                                                              *head_params)
                finish(counters)
                return logits.squeeze(1), coords.reshape(B * 4, -1)
            if i == 0:
                h = _LayerTrainFn.apply(h, conv.lin.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                        graph, gb, relu, p, momentum, bn.eps, seed, bool(self.residual), kid, down_of(i))
            else:
                mlp_cfg, mlp_params = self._coord_mlp_train_cfg(self.node_coordinate_mlp[i - 1], counters)
                cfg = (graph, gb, relu, p, momentum, bn.eps, seed, bool(self.residual), dims, mlp_cfg, kid, down_of(i))
                h, coords = _CoordLayerTrainFn.apply(h, coords, conv.lin.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                                     bn.running_var, cfg, *mlp_params)
        mlp_cfg, mlp_params = self._coord_mlp_train_cfg(self.node_coordinate_mlp[self.num_gnn_layers - 1], counters)
        cls_cfg, head_params, finish = self._classifier_train_cfg()
        logits, coords = _CoordClassifierTrainFn.apply(h, coords, (B, n, n_conn, n_valid), self.output_activation == "sigmoid",
                                                       cls_cfg, dims, mlp_cfg, *mlp_params, *head_params)
        finish(counters)
        return logits.squeeze(1), coords.reshape(B * 4, -1)

    def _train_kidsums(self, graph: ops.Graph, gb: int):
        """Child-sum side buffers of the chained train forward (layer i leaves the child sums of its output for layer i + 1:
        eg_gcn_layer_train_fwd), or (None, None)."""
        if graph.kidsum_rows == 0 or self.num_gnn_layers < 2 or not ROUTES.train_chain or \
                not ops.train_chain_supported():
            return None, None
        return self._kidsum_buffers(graph, gb)

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
        fused = (not self.training) and (not torch.is_grad_enabled() or not node_feats.requires_grad) and not self._narrow
        fused = fused and self.layer_output_hook is None and not any(
            p.requires_grad and torch.is_grad_enabled() for p in self.parameters())
        # JumpingKnowledge('max') stays on the fused path as a running maximum written by the layer kernels
        # (eg_gcn_layer_fwd_jk); where those do not cover the handle (CSR graphs, coordinate / connection nodes) the
        # layers run one by one and torch takes the maximum, as before
        jk_fused = (fused and self.jk is not None and graph.fused_classifier_ok and not self.use_coordinate_graph
                    and not graph.hybrid and ROUTES.jk_fused)
        fused = fused and (self.jk is None or jk_fused)
        if not fused and self._train_coord_fused_ok(node_coords):
            return self._forward_train_coord_fused(node_feats, graph, gb, B, node_coords)
        if fused and self.use_hip_graph and not self.use_coordinate_graph and node_feats.is_contiguous() \
                and not torch.cuda.is_current_stream_capturing():
            return self._forward_nodes_graphed(node_feats, edge_index, B), None
        hidden = [node_feats.contiguous()]
        kid = (None, None)
        if fused:
            folded = self._folded_layers()
            # chained layers: each layer leaves the child sums of its output behind for the next one
            if self.chain_layers and not self.use_coordinate_graph and graph.kidsum_rows > 0 and self.num_gnn_layers > 1:
                kid = self._kidsum_buffers(graph, gb)
        fuse_cls = (fused and self.fuse_classifier and graph.fused_classifier_ok and not self.use_coordinate_graph
                    and (kid[0] is not None or graph.kidsum_rows == 0) and n_conn == graph.num_conn and n_valid == n - n_conn
                    and self.num_output_channels == 4 and self.classifier_hidden_dim == 32)
        jkb = self._jk_buffers(graph, gb, node_feats) if jk_fused else None
        train_kids = (None, None)
        boxes = None
        if self.training and not fused and not self._narrow and all(self._layer_cfg_static_ok(i) for i in range(self.num_gnn_layers)):
            train_kids = self._train_kidsums(graph, gb)
            # (nothing but the next layer may consume a layer's output: a hook, JumpingKnowledge or the explicit coordinate update
            # would put other gradients or row patches between the dX launch and the layer below)
            if not self.use_coordinate_graph and self.layer_output_hook is None and self.jk is None:
                boxes = self._sums_down_boxes(graph)
        L_ = self.num_gnn_layers
        down_of = (lambda i: None) if boxes is None else (lambda i: (boxes[i] if i < L_ - 1 else None, boxes[i - 1] if i > 0 else None, n))
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
            elif self.training and i == self.num_gnn_layers - 1 and self._act_in_heads_ok():
                # the last layer and the heads as one node: the layer's activation pass runs inside the heads' first kernel
                conv, bn, relu, p, seed = self._layer_cfg(i)
                _, momentum = _bn_step(bn)
                cls_cfg, head_params, finish = self._classifier_train_cfg()
                cfg = (graph, gb, relu, p, momentum, bn.eps, seed, bool(self.residual), train_kids[(i + 1) & 1] if i > 0 else None,
                       (B, n, n_conn, n_valid), self.output_activation == "sigmoid", cls_cfg, None, None, None, down_of(i))
                out, _ = _LastLayerHeadsTrainFn.apply(x_in, None, conv.lin.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                                                      bn.running_var, cfg, *head_params)
                finish()
                return out.squeeze(1), None
            elif self.training and not self._narrow:
                tk = train_kids if not self.use_coordinate_graph else (None, None)      # (the explicit coordinate update rewrites rows)
                h = self._layer_train(i, x_in, graph, gb, (tk[(i + 1) & 1] if i > 0 else None,
                                                           tk[i & 1] if i < self.num_gnn_layers - 1 else None), down_of(i))
            else:
                h = self.gnn_layers[i].forward_graph(x_in, graph, gb)
                if self.residual and h.shape[1] == x_in.shape[1]:
                    h = h + x_in
            if self.use_coordinate_graph:
                h, node_coords = self._coordinate_update(i, h, node_coords, B)
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
                hv = h.view(B, n, h.shape[1])[:, n_conn:n_conn + n_valid, :].reshape(B * n_valid, h.shape[1])
                out = torch.cat([clf(hv) for clf in self.node_classifiers], dim=1)
        if self.use_coordinate_graph:
            node_coords = node_coords.reshape(B * 4, -1)
        return out.squeeze(1), node_coords

    def _act_in_heads_ok(self) -> bool:
        """Train mode without the coordinate graph: may the last layer + the heads run as _LastLayerHeadsTrainFn?"""
        return (not self._narrow and not self.use_coordinate_graph and self.layer_output_hook is None and self.jk is None
                and self._stacked_heads_ok() and self._layer_cfg_static_ok(self.num_gnn_layers - 1)
                and ROUTES.act_in_heads)

    # ---- the 4 classifier heads in train mode as ONE stacked network ----------------------------------------
    def _stacked_heads_ok(self) -> bool:
        mods = [hd._modules for hd in self.node_classifiers]
        ref = mods[0]["1"]
        plain_bn = all(m.training and m.affine and m.track_running_stats and m.momentum is not None and
                       m.momentum == ref.momentum and m.eps == ref.eps for md in mods for m in (md["1"], md["5"]))
        drops_on = all(m.training for md in mods for m in (md["3"], md["7"]))
        return (not self._narrow and self.num_output_channels == 4 and self.classifier_hidden_dim == 32 and self.node_embedding_dim == C
                and plain_bn and drops_on and ROUTES.stacked_heads)

    def _classifier_train_cfg(self):
        """(cfg, the 40 head parameters, finish()) for _ClassifierTrainFn / _CoordClassifierTrainFn.  Running statistics: the
        kernels update stacked copies, which finish() writes back to the 8 BatchNorm modules with two multi-tensor copies."""
        heads = list(self.node_classifiers)
        bn1, bn2 = [hd._modules["1"] for hd in heads], [hd._modules["5"] for hd in heads]
        p1, p2 = float(heads[0]._modules["3"].p), float(heads[0]._modules["7"].p)
        seeds = torch.randint(0, 2 ** 62, (2,)).tolist() if (p1 > 0 or p2 > 0) else [0, 0]      # host RNG, like the layers
        if self.dropout_seed_hook is not None:
            self.dropout_seed_hook("heads", self.node_classifiers, tuple(seeds))
        params = [p for hd in heads for p in _seq_params(hd)]
        stat_list = [b.running_mean for b in bn1] + [b.running_var for b in bn1] + [b.running_mean for b in bn2] + [b.running_var for b in bn2]
        banks = self._heads_in_place(params, stat_list, bn1, bn2)
        with torch.no_grad():
            if banks is not None:                          # the modules' running statistics ARE the stacked arrays: nothing to copy, either way
                stats = banks[1]
            else:                                          # stacked copies of the running statistics: one multi-tensor copy
                stats = torch.empty(2 * 128 + 2 * 64, dtype=torch.float32, device=bn1[0].running_mean.device)
            rm1, rv1, rm2, rv2 = stats[:128], stats[128:256], stats[256:320], stats[320:384]
            if banks is None:
                torch._foreach_copy_(list(rm1.split(32)) + list(rv1.split(32)) + list(rm2.split(16)) + list(rv2.split(16)), stat_list)
        cfg = dict(running_mean1=rm1, running_var1=rv1, running_mean2=rm2, running_var2=rv2, eps1=bn1[0].eps, eps2=bn2[0].eps,
                   momentum1=bn1[0].momentum, momentum2=bn2[0].momentum, p1=p1, p2=p2, seed1=seeds[0], seed2=seeds[1])
        if banks is not None:
            cfg["_param_bank"] = banks[0]

        def finish(more_counters=()):
            with torch.no_grad():
                if banks is None:
                    torch._foreach_copy_(stat_list, list(rm1.split(32)) + list(rv1.split(32)) + list(rm2.split(16)) + list(rv2.split(16)))
                torch._foreach_add_([b.num_batches_tracked for b in bn1 + bn2] + list(more_counters), 1)
        return cfg, params, finish

    def _heads_in_place(self, params, stat_list, bn1, bn2):
        """(parameter bank [4 * sum(_HEAD_SIZES)], statistics bank [384]) with the 40 head parameters and the 16 running-statistics
        buffers living INSIDE them, in the stacked layout the kernels take -- or None (ROUTES.heads_state_in_place off, tensors that are
        not CUDA float32, a stream capture under way).  Stacking them per step was three multi-tensor copy launches; here the tensors
        are moved into the banks once (``p.data`` / the buffer re-pointed at its slice: same values, same Parameter objects, so
        optimizers, state_dict() and load_state_dict() see no difference) and every later step only checks addresses.  Whatever
        re-allocates them (``.to()``, a Parameter assigned by hand, copy.deepcopy of the model) is noticed by that check and they
        are moved again."""
        if not ROUTES.heads_state_in_place:
            return None
        offs_p = _head_param_offsets()
        offs_s = [32 * k for k in range(4)] + [128 + 32 * k for k in range(4)] + [256 + 16 * k for k in range(4)] + [320 + 16 * k for k in range(4)]
        banks = self.__dict__.get("_head_banks")
        if banks is not None and _views_of(banks[0], params, offs_p) and _views_of(banks[1], stat_list, offs_s):
            return banks
        dev = params[0].device
        if dev.type != "cuda" or any(t.dtype != torch.float32 or t.device != dev for t in list(params) + list(stat_list)) or \
                torch.cuda.is_current_stream_capturing():
            return None
        pbank = torch.empty(4 * sum(_HEAD_SIZES), dtype=torch.float32, device=dev)
        sbank = torch.empty(384, dtype=torch.float32, device=dev)

        def set_param(i, view):
            params[i].data = view

        def set_stat(i, view):
            kind, k = divmod(i, 4)
            setattr((bn1 if kind < 2 else bn2)[k], "running_mean" if kind % 2 == 0 else "running_var", view)
            stat_list[i] = view
        _move_into(pbank, params, offs_p, set_param)
        _move_into(sbank, stat_list, offs_s, set_stat)
        banks = self.__dict__["_head_banks"] = (pbank, sbank)
        return banks

    def _classifier_train(self, h: torch.Tensor, B: int, n: int, row_lo: int, n_valid: int) -> torch.Tensor:
        """models.py:363-377, :485-490 in train mode on the HIP kernels (_ClassifierTrainFn): the node-type filter is a row
        range, the four heads run as one stacked network."""
        cfg, params, finish = self._classifier_train_cfg()
        out = _ClassifierTrainFn.apply(h, B, n, row_lo, n_valid, self.output_activation == "sigmoid", cfg, *params)
        finish()
        return out

    def _layer_cfg_static_ok(self, i: int) -> bool:
        l = self.gnn_layers[i]
        return bool(l.module_1.training and l.module_1.affine and l.module_2.training)

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
        # keyed on what the captured kernels actually point at: the input buffer's ADDRESS (the entry holds a reference to the
        # tensor it was captured on, so that address cannot be handed to anybody else while the entry lives: another tensor
        # object with the same data_ptr is a view of the same storage), the graph handle the edge_index resolves to (an equal
        # edge_index in a fresh tensor replays too), and the parameter versions -- not on the identity of either input tensor
        graph, gb = self._resolver.resolve(edge_index, node_feats.shape[0])
        banks = self.__dict__.get("_head_banks")          # (the heads' running statistics move when _heads_in_place re-banks them)
        key = (node_feats.data_ptr(), tuple(node_feats.shape), node_feats.device, id(graph), gb, B,
               tuple(_versions(l) for l in self.gnn_layers), tuple(_versions(c) for c in self.node_classifiers),
               None if banks is None else banks[1].data_ptr())
        hit = self._hip_graphs.get(key)
        if hit is not None and hit[4][0] is not graph:
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
            self.hip_graph_captures += 1
            if len(self._hip_graphs) > 8:
                self._hip_graphs.clear()
            # everything the captured kernels point at stays alive with the entry: the input buffer, the graph handle,
            # the child-sum side buffers and the folded / packed parameters (their caches may evict independently)
            keep = (graph, self._kidsum.get((id(graph), gb)), self._kidsum.get(("jk", id(graph), gb, tuple(node_feats.shape))),
                    self._fold_cache.get("layers"), self._fold_cache.get("cls"))
            hit = (g, out, node_feats, None, keep)
            self._hip_graphs[key] = hit
        hit[0].replay()
        return hit[1]

    # ---- avg-pool node features (models.py:498-537): the step in front of the hot path ---------
    def create_node_pixels(self, echo_frames: torch.Tensor, num_samples_per_batch: int, node_coords=None):
        """models.py:498-537: average-pooled pyramid of the frame embedding + the frame itself, node-major."""
        B = int(num_samples_per_batch)
        conn = None
        if self.use_connection_nodes and not self.use_main_graph_only:
            conn = echo_frames.mean(dim=(2, 3)).unsqueeze(1).expand(B, self.num_aux_graphs + 1, C)
        sides = [] if self.use_main_graph_only else [2 ** g for g in range(1, self.num_aux_graphs + 1)]
        if sides and echo_frames.shape[1] == C and ops.pyramid_supported(echo_frames, sides) and os.environ.get("EG_POOL_PYRAMID", "1") != "0":
            # the pooled pyramid and the packing as ONE autograd node, two launches each way (eg_avg_pool_pyramid_*, eg_pack_levels):
            # torch's adaptive_avg_pool2d is a launch per level (219 us each at 224 x 224, batch 1) and, backwards, a launch of float
            # atomics per level (254 us each) -- 3.3 ms of a batch-1 training step whose GNN stack takes 1 ms
            n, n_conn, _, main_base, coord_base = self._row_ranges()
            x = echo_frames.float()
            feats = ops.pyramid_pack(x, sides, B, n, n_conn, out=self._static_node_feats(B, [x, conn]))
            return self._finish_node_features(feats, B, node_coords, conn)
        maps = [F.adaptive_avg_pool2d(echo_frames, output_size=(p, p)) for p in sides]
        maps.append(echo_frames)
        return self.pack_node_features(maps, B, node_coords, conn)

    def _finish_node_features(self, feats, B, node_coords, connection_embed):
        """Connection-node rows and coordinate-node samples on top of the packed levels (models.py:524-533)."""
        n, n_conn, _, main_base, coord_base = self._row_ranges()
        if n_conn:
            feats = feats.clone() if feats.requires_grad else feats
            feats.view(B, n, C)[:, :n_conn, :] = connection_embed
        if self.use_coordinate_graph and not self.use_main_graph_only:
            new = ops.bilinear4(feats, node_coords.reshape(B, 4, 2).contiguous(), B, n, main_base, self.frame_size)
            feats = ops.scatter_coord_rows(feats, new, B, n, coord_base)
        return feats

    def pack_node_features(self, level_maps, num_samples_per_batch: int, node_coords=None, connection_embed=None):
        """The tail every create_node_pixels variant of the reference shares (models.py:511-537, :603-636, :726-756):
        NCHW level maps (coarse to fine, the last one is the frame-sized map) -> [B*N, 128] in the GNN's node order,
        in one packing launch (eg_pack_levels) instead of a per-sample permute / cat loop.  The UNet / CNN variants
        pass their own per-level feature maps and connection-node embeddings [B, naux+1, 128]."""
        B = int(num_samples_per_batch)
        n, n_conn, _, main_base, coord_base = self._row_ranges()
        fs = self.frame_size
        maps = [m.float() for m in level_maps]
        feats = ops.pack_levels(maps, B, n, n_conn, out=self._static_node_feats(B, maps + [connection_embed]))
        return self._finish_node_features(feats, B, node_coords, connection_embed)

    def pack_node_features_linear(self, features, linears, num_samples_per_batch: int, node_coords=None, connection_embed=None):
        """The UNet variant's whole tail (models.py:707-756): ``F.relu(self.linears[i](features[i]))`` for every level (1x1
        convolutions to 128 channels) AND the node-major packing in one launch (eg_conv1x1_relu_pack_levels); the 128-channel
        NCHW maps are never formed.  ``features``: decoder maps coarse to fine [B, C_l, p_l, p_l] (the last one frame-sized),
        ``linears``: the matching ``nn.Conv2d(C_l, 128, kernel_size=1)`` modules.  Connection-node embeddings
        [B, naux+1, 128] (means of the activated maps) come from the caller, as in ``pack_node_features``."""
        B = int(num_samples_per_batch)
        n, n_conn, _, main_base, coord_base = self._row_ranges()
        fs = self.frame_size
        fl = [f.float() for f in features]
        ws, bs = [m.weight for m in linears], [m.bias for m in linears]
        feats = ops.conv1x1_relu_pack_levels(fl, ws, bs, B, n, n_conn,
                                             out=self._static_node_feats(B, fl + ws + bs + [connection_embed]))
        return self._finish_node_features(feats, B, node_coords, connection_embed)

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
