"""Route A of INTEGRATION.md for the reference's UNet variant, as a runnable example.

``UNetNodeFeatureModel`` has the shape of the reference's ``UNETHierarchicalPatchModel`` (src/core/models.py:639-756; the class
``configs/default.yml`` names as ``unet_hierarchical_patch``): a convolutional encoder / decoder in front of the GNN whose
decoder maps become the node features of the levels.  The front-end is dense convolution work that stock PyTorch-ROCm (MIOpen)
runs as it is -- it is OUT of the hot path's scope and is written here with plain torch modules only to
show where the HIP path begins: at the tail of ``create_node_pixels``, where the reference applies a 1x1 convolution + ReLU to
every decoder map and concatenates the permuted maps per sample in a Python loop (models.py:707-756).  That tail is ONE launch
here (``pack_node_features_linear`` -> eg_conv1x1_relu_pack_levels) and writes the GNN's input layout directly; everything
after it is the base class's ``forward_nodes``.

Module names (``down_convs.{i}.conv1 / BN1 / conv2 / BN2``, ``up_convs.{i}.*``, ``linears.{i}``) are the reference's, so a
reference checkpoint of this variant loads with ``strict=True`` (src/core/checkpointers.py:94-98)."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .nn import C, HierarchicalPatchModel
from .topology import get_topology


class _Down(nn.Module):
    """(conv3x3 -> ReLU -> BN) x 2 -> adaptive max pool to ``out_side`` (the reference's DownConv, models.py:841-856)."""

    def __init__(self, c_in: int, c_out: int, out_side: int):
        super().__init__()
        self.conv1, self.BN1 = nn.Conv2d(c_in, c_out, 3, padding=1), nn.BatchNorm2d(c_out)
        self.conv2, self.BN2 = nn.Conv2d(c_out, c_out, 3, padding=1), nn.BatchNorm2d(c_out)
        self.pool = nn.AdaptiveMaxPool2d(out_side)

    def forward(self, x):
        x = self.BN1(F.relu(self.conv1(x)))
        return self.pool(self.BN2(F.relu(self.conv2(x))))


class _Up(nn.Module):
    """upsample to ``out_side`` -> conv3x3 (halves the channels) -> cat(skip) -> conv3x3 (models.py:859-876)."""

    def __init__(self, c_in: int, c_out: int, out_side: int):
        super().__init__()
        self.upsample = nn.Upsample(size=out_side)
        self.conv1, self.BN1 = nn.Conv2d(c_in, c_out, 3, padding=1), nn.BatchNorm2d(c_out)
        self.conv2, self.BN2 = nn.Conv2d(c_in, c_out, 3, padding=1), nn.BatchNorm2d(c_out)

    def forward(self, x, skip):
        x = self.BN1(F.relu(self.conv1(self.upsample(x))))
        return self.BN2(F.relu(self.conv2(torch.cat([x, skip], dim=1))))


class UNetNodeFeatureModel(HierarchicalPatchModel):
    """``HierarchicalPatchModel`` whose node features come from a UNet decoder (same constructor as the reference class:
    ``encoder_embedding_widths`` / ``encoder_embedding_dims`` + the base class's keyword arguments; ``forward`` takes the
    embedder's ``[B, dims[0] // 2, F, F]`` frames, engine.py:240)."""

    def __init__(self, encoder_embedding_widths: Optional[List[int]] = None, encoder_embedding_dims: Optional[List[int]] = None,
                 **kwargs):
        super().__init__(**kwargs)
        widths = [128, 64, 32, 16, 8, 4, 2] if encoder_embedding_widths is None else list(encoder_embedding_widths)
        dims = [8, 16, 32, 64, 128, 256, 512] if encoder_embedding_dims is None else list(encoder_embedding_dims)
        if len(widths) != len(dims):
            raise ValueError("encoder_embedding_widths and encoder_embedding_dims must have the same length")
        if not self.use_main_graph_only and self.num_aux_graphs != len(widths):
            raise ValueError(f"the decoder yields {len(widths)} coarse maps (sides {sorted(widths)}) and the frame-sized one; "
                             f"num_aux_graphs={self.num_aux_graphs} levels need exactly that many")
        if not self.use_main_graph_only and sorted(widths) != [2 ** g for g in range(1, self.num_aux_graphs + 1)]:
            raise ValueError("level g of the graph is a 2^g x 2^g grid: the encoder widths must be those sides")
        self.down_convs = nn.ModuleList(_Down(d // 2, d, w) for d, w in zip(dims, widths))
        up_sides = list(reversed(widths))[1:] + [self.frame_size]
        self.up_convs = nn.ModuleList(_Up(d, d // 2, s) for d, s in zip(reversed(dims), up_sides))
        feats_in = list(reversed(dims)) + [dims[0] // 2]
        self.linears = nn.ModuleList(nn.Conv2d(c, self.node_embedding_dim, kernel_size=1) for c in feats_in)

    def decoder_maps(self, frames: torch.Tensor) -> List[torch.Tensor]:
        """[B, dims[0] // 2, F, F] -> the decoder's maps, coarse to fine: sides 2, 4, ..., 2^naux, F."""
        x, skips = frames, []
        for down in self.down_convs:
            skips.append(x)
            x = down(x)
        feats = [x]
        for up in self.up_convs:
            x = up(x, skips.pop())
            feats.append(x)
        return feats

    def create_node_pixels(self, echo_frames: torch.Tensor, num_samples_per_batch: int, node_coords=None):
        B = int(num_samples_per_batch)
        feats = self.decoder_maps(echo_frames)
        lin = list(self.linears)
        if self.use_main_graph_only:
            feats, lin = feats[-1:], lin[-1:]
        conn = None
        if self.use_connection_nodes and not self.use_main_graph_only:
            # the connection nodes start from the mean of every level's ACTIVATED map (models.py:737-752): activated here, once
            # more, on the coarse maps only -- they are a few percent of the frame-sized one -- and on the frame-sized map's mean
            with torch.set_grad_enabled(torch.is_grad_enabled()):
                conn = torch.stack([F.relu(m(f)).mean(dim=(2, 3)) for f, m in zip(feats, lin)], dim=1)        # [B, naux + 1, 128]
        return self.pack_node_features_linear(feats, lin, B, node_coords, conn)


def reference_tail(model: UNetNodeFeatureModel, feats: List[torch.Tensor], B: int) -> torch.Tensor:
    """What the reference's tail computes with torch ops (plain levels, no connection / coordinate nodes): the check of the
    fused launch in tests/test_gpu_pack.py and the `unfused` leg of bench.py's end-to-end context number."""
    maps = [F.relu(m(f)) for f, m in zip(feats, model.linears)]
    rows = []
    for i in range(B):
        rows += [m[i].permute(1, 2, 0).reshape(-1, C) for m in maps]
    return torch.cat(rows, dim=0)
