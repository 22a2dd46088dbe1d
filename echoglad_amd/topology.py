"""Closed-form hierarchical graph topology (host logic, numpy only).

The reference builds the per-frame multi-resolution graph with networkx in the
dataset constructor (reference: src/core/datasets.py:1441-1584, identical
copies at :375-521, :739-885, :1142-1288) and converts it with
``from_networkx`` on every ``__getitem__`` (:1392).  The graph is a pure
function of ``(frame_size, num_aux_graphs, flags)``; this module states it in
closed form in O(N) so the HIP path can (a) verify an incoming ``edge_index``
against it and (b) run the implicit-stencil kernels that never read an edge
list.

Node numbering per frame (matches the insertion order of the reference's
``nx.compose`` chain followed by ``convert_node_labels_to_integers``):

    [connection nodes: naux+1]           node_type 2   (use_connection_nodes)
    level k = 1..naux: p x p grid, p=2^k node_type 0   row-major, id = base_k + r*p + c
    main grid F x F                      node_type 0   id = main_base + r*F + c
    [coordinate nodes: 4]                node_type 1   (use_coordinate_graph)

Edges (undirected; both directions are emitted in ``edge_index``):
  * 4-neighbour inside every level (+ both diagonals for 'grid-diagonal'),
  * parent (r,c)@k <-> children (2r+dr, 2c+dc)@k+1,
  * last aux level <-> main grid through a centre crop that uses *Python slice
    semantics* (datasets.py:1565-1567: negative start wraps / clamps),
  * K4 among the coordinate nodes, isolated from everything else (:1517-1523),
  * K_{naux+1} among connection nodes and connection node g-1 <-> every node of
    aux level g for g = 1..naux-1 (:1452-1456, :1512-1515).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass
from functools import lru_cache
from typing import List, Tuple

import numpy as np


@dataclass(frozen=True)
class TopologySpec:
    """The knobs of the reference dataset constructor that decide the graph
    (reference: src/core/datasets.py:1341-1372; configs/default.yml:69-79)."""
    frame_size: int = 224
    num_aux_graphs: int = 7
    use_main_graph_only: bool = False
    use_coordinate_graph: bool = False
    use_connection_nodes: bool = False
    main_graph_type: str = "grid"
    aux_graph_type: str = "grid"

    def __post_init__(self):
        for t in (self.main_graph_type, self.aux_graph_type):
            if t not in ("grid", "grid-diagonal"):
                raise ValueError(f"unsupported graph type {t!r}")
        if self.frame_size < 2:
            raise ValueError("frame_size must be >= 2")
        if not self.use_main_graph_only and self.num_aux_graphs < 1:
            raise ValueError("num_aux_graphs must be >= 1")


@dataclass(frozen=True)
class Level:
    """One grid level: ``side x side`` nodes starting at node id ``base``."""
    base: int
    side: int

    @property
    def size(self) -> int:
        return self.side * self.side


class HierTopology:
    """Per-frame topology in closed form."""

    def __init__(self, spec: TopologySpec):
        self.spec = spec
        F = spec.frame_size
        aux_on = not spec.use_main_graph_only
        # the reference only adds coordinate / connection nodes inside the
        # `if not self.use_main_graph_only` branches (datasets.py:1450,1508)
        self.n_conn = (spec.num_aux_graphs + 1) if (aux_on and spec.use_connection_nodes) else 0
        self.n_coord = 4 if (aux_on and spec.use_coordinate_graph) else 0
        self.aux_levels: List[Level] = []
        nid = self.n_conn
        if aux_on:
            for k in range(1, spec.num_aux_graphs + 1):
                p = 2 ** k
                self.aux_levels.append(Level(nid, p))
                nid += p * p
        self.main = Level(nid, F)
        nid += F * F
        self.coord_base = nid
        nid += self.n_coord
        self.num_nodes = nid
        # crop of the last aux level that is wired to the main grid
        if aux_on:
            p = self.aux_levels[-1].side
            half = F // 2
            c0 = (p - half) // 2
            rows = list(range(p))[c0:c0 + half]      # Python slice semantics on purpose
            self.crop_rows: List[int] = rows         # identical for rows and cols
        else:
            self.crop_rows = []

    # ------------------------------------------------------------------ edges
    @staticmethod
    def _grid_edges(level: Level, diagonal: bool) -> np.ndarray:
        p, b = level.side, level.base
        ids = b + np.arange(p * p, dtype=np.int64).reshape(p, p)
        parts = [np.stack([ids[:, :-1].ravel(), ids[:, 1:].ravel()]),
                 np.stack([ids[:-1, :].ravel(), ids[1:, :].ravel()])]
        if diagonal:
            parts.append(np.stack([ids[:-1, :-1].ravel(), ids[1:, 1:].ravel()]))
            parts.append(np.stack([ids[1:, :-1].ravel(), ids[:-1, 1:].ravel()]))
        return np.concatenate(parts, axis=1)

    @staticmethod
    def _parent_child_edges(parent_ids: np.ndarray, child: Level) -> np.ndarray:
        """parent_ids: [h, w] node ids; parent (x,y) <-> child (2x+dx, 2y+dy)."""
        h, w = parent_ids.shape
        x = np.arange(h, dtype=np.int64)[:, None]
        y = np.arange(w, dtype=np.int64)[None, :]
        parts = []
        for dx in (0, 1):
            for dy in (0, 1):
                cid = child.base + (2 * x + dx) * child.side + (2 * y + dy)
                parts.append(np.stack([parent_ids.ravel(), np.broadcast_to(cid, (h, w)).ravel()]))
        return np.concatenate(parts, axis=1)

    def undirected_edges(self) -> np.ndarray:
        """[2, E_undirected] int64, each undirected edge once (u, v), unordered."""
        s = self.spec
        parts = []
        if self.n_conn:
            c = np.arange(self.n_conn, dtype=np.int64)
            iu, ju = np.triu_indices(self.n_conn, k=1)
            parts.append(np.stack([c[iu], c[ju]]))
        for lv in self.aux_levels:
            parts.append(self._grid_edges(lv, s.aux_graph_type == "grid-diagonal"))
        parts.append(self._grid_edges(self.main, s.main_graph_type == "grid-diagonal"))
        for k in range(len(self.aux_levels) - 1):
            par = self.aux_levels[k]
            ids = par.base + np.arange(par.size, dtype=np.int64).reshape(par.side, par.side)
            parts.append(self._parent_child_edges(ids, self.aux_levels[k + 1]))
        if self.aux_levels:
            last = self.aux_levels[-1]
            rows = np.asarray(self.crop_rows, dtype=np.int64)
            if rows.size:
                ids = last.base + rows[:, None] * last.side + rows[None, :]
                parts.append(self._parent_child_edges(ids, self.main))
        if self.n_conn:
            for g in range(1, self.spec.num_aux_graphs):
                lv = self.aux_levels[g - 1]
                nodes = lv.base + np.arange(lv.size, dtype=np.int64)
                parts.append(np.stack([np.full(lv.size, g - 1, dtype=np.int64), nodes]))
        if self.n_coord:
            c = self.coord_base + np.arange(4, dtype=np.int64)
            iu, ju = np.triu_indices(4, k=1)
            parts.append(np.stack([c[iu], c[ju]]))
        return np.concatenate(parts, axis=1)

    def edge_index(self) -> np.ndarray:
        """[2, E_dir] int64, both directions, sorted by (source, target).

        The reference's order (``from_networkx``: per source node in insertion
        order, neighbours in adjacency-insertion order) differs only in the
        order of neighbours inside one source node, which affects nothing but
        floating-point summation order."""
        e = self.undirected_edges()
        both = np.concatenate([e, e[::-1]], axis=1)
        order = np.lexsort((both[1], both[0]))
        return np.ascontiguousarray(both[:, order])

    def node_type(self) -> np.ndarray:
        """float64 like the reference (np.zeros / np.ones defaults, datasets.py:1477,1520)."""
        t = np.zeros(self.num_nodes, dtype=np.float64)
        t[:self.n_conn] = 2.0
        if self.n_coord:
            t[self.coord_base:] = 1.0
        return t

    def degree(self) -> np.ndarray:
        """In-degree without the self loop, int64 [N]."""
        e = self.undirected_edges()
        return (np.bincount(e[0], minlength=self.num_nodes)
                + np.bincount(e[1], minlength=self.num_nodes)).astype(np.int64)

    def deg_inv_sqrt(self) -> np.ndarray:
        """(deg+1)^-1/2 as float32 — the GCN symmetric normalisation with self loops."""
        d = (self.degree() + 1).astype(np.float32)
        return (d ** np.float32(-0.5)).astype(np.float32)

    # -------------------------------------------------------------- summaries
    def count_undirected_edges(self) -> int:
        """Closed-form edge count (no arrays): what `undirected_edges()` would return, in O(levels)."""
        s = self.spec

        def grid(p, diagonal):
            return 2 * p * (p - 1) + (2 * (p - 1) * (p - 1) if diagonal else 0)

        e = grid(self.main.side, s.main_graph_type == "grid-diagonal")
        for lv in self.aux_levels:
            e += grid(lv.side, s.aux_graph_type == "grid-diagonal")
        for lv in self.aux_levels[:-1]:
            e += 4 * lv.size                                  # parent <-> its 4 children
        e += 4 * len(self.crop_rows) ** 2                     # last aux level <-> main grid through the crop
        if self.n_conn:
            e += self.n_conn * (self.n_conn - 1) // 2
            e += sum(self.aux_levels[g - 1].size for g in range(1, s.num_aux_graphs))
        if self.n_coord:
            e += 6
        return e

    @property
    def num_undirected_edges(self) -> int:
        return self.count_undirected_edges()

    @property
    def num_valid_nodes(self) -> int:
        """rows that survive the node_type == 0 filter (models.py:485)."""
        return self.num_nodes - self.n_conn - self.n_coord

    def edge_set_digest(self) -> str:
        """sha256 over the sorted undirected edge list (min,max) as little-endian int64."""
        e = self.undirected_edges()
        lo, hi = np.minimum(e[0], e[1]), np.maximum(e[0], e[1])
        order = np.lexsort((hi, lo))
        arr = np.ascontiguousarray(np.stack([lo[order], hi[order]], axis=1).astype("<i8"))
        return hashlib.sha256(arr.tobytes()).hexdigest()

    def degree_histogram(self) -> dict:
        d = self.degree()
        vals, cnt = np.unique(d, return_counts=True)
        return {int(v): int(c) for v, c in zip(vals, cnt)}

    # --------------------------------------------------------- batched forms
    def batched_edge_index(self, batch: int) -> np.ndarray:
        """Disjoint union of ``batch`` frames: per-frame node offset N (PyG collate).  The array of a batch size is built once and
        handed out again (read-only: 0.4 s of numpy at 224/7, batch 8 -- per training step if it were rebuilt in every collate);
        the four most recent batch sizes are kept."""
        cache = self.__dict__.setdefault("_batched_ei", {})
        hit = cache.get(batch)
        if hit is not None:
            return hit
        e = self.edge_index()
        off = (np.arange(batch, dtype=np.int64) * self.num_nodes)[:, None, None]
        out = np.ascontiguousarray((e[None] + off).transpose(1, 0, 2).reshape(2, -1))
        out.setflags(write=False)
        while len(cache) >= 4:
            cache.pop(next(iter(cache)))
        cache[batch] = out
        return out

    def level_table(self) -> np.ndarray:
        """int32 [n_levels, 2] = (base, side) for aux levels then the main grid."""
        rows = [(lv.base, lv.side) for lv in self.aux_levels] + [(self.main.base, self.main.side)]
        return np.asarray(rows, dtype=np.int32)

    def is_structured(self) -> bool:
        """True when the implicit-stencil kernels cover this graph: 4-neighbour or 'grid-diagonal'
        (8-neighbour) grids, with or without coordinate / connection nodes (round 4; every closed
        form of the reference's builder), and a crop that is one contiguous run of the last aux
        level (always the case with Python slices of a range)."""
        return True


@lru_cache(maxsize=32)
def get_topology(spec: TopologySpec) -> HierTopology:
    return HierTopology(spec)


def candidate_specs(num_rows: int, num_directed_edges: int, max_frame: int = 4096, max_aux: int = 12):
    """Every structured closed-form topology (plain grids, no connection nodes) and batch size whose node and edge
    counts equal the given totals: [(TopologySpec, batch)].  Lets a stand-alone ``GCNConv`` — which is constructed
    without any graph information (models.py:330-331) — recognise the reference's graphs from an incoming
    ``edge_index``; a candidate still has to pass the edge-digest check before it is used."""
    out = []
    if num_rows <= 0 or num_directed_edges <= 0 or num_directed_edges % 2:
        return out
    F = np.arange(2, max_frame + 1, dtype=np.int64)
    variants = [(True, 1, 0, 0)] + [(False, a, c, k) for a in range(1, max_aux + 1) for c in (0, 4) for k in (0, a + 1)]
    for main_only, naux, n_coord, n_conn in variants:
        aux_nodes = 0 if main_only else sum(4 ** k for k in range(1, naux + 1))
        n = aux_nodes + F * F + n_coord + n_conn
        for f in F[(n <= num_rows) & (num_rows % n == 0)]:
            for spec in graph_type_variants(TopologySpec(int(f), naux, main_only, n_coord > 0, n_conn > 0)):
                topo = HierTopology(spec)
                batch = num_rows // topo.num_nodes
                if 2 * batch * topo.count_undirected_edges() == num_directed_edges:
                    out.append((spec, batch))
    return out


def graph_type_variants(spec: TopologySpec):
    """The spec with every combination of 'grid' / 'grid-diagonal' levels (the graph TYPE is dataset configuration,
    datasets.py:1441: a model is constructed without it and meets it only in the edge_index)."""
    from dataclasses import replace
    aux_types = ("grid",) if spec.use_main_graph_only else ("grid", "grid-diagonal")
    return [replace(spec, main_graph_type=m, aux_graph_type=a) for m in ("grid", "grid-diagonal") for a in aux_types]


def commutative_edge_hash(edge_index: np.ndarray) -> Tuple[int, int]:
    """(E_dir, order-independent 64-bit hash) — the host mirror of the device
    kernel used to verify an incoming edge_index against the closed form."""
    r = edge_index[0].astype(np.uint64)
    c = edge_index[1].astype(np.uint64)
    with np.errstate(over="ignore"):
        h = r * np.uint64(0x9E3779B97F4A7C15) + c * np.uint64(0xC2B2AE3D27D4EB4F) + np.uint64(0x165667B19E3779F9)
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(32)
        total = np.add.reduce(h, dtype=np.uint64)
    return int(edge_index.shape[1]), int(total)
