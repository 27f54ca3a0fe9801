"""Voxel de-duplication of raw clouds on the GPU: the reference's ``ME.utils.sparse_quantize`` call
(Experiments/dataloader/generic_balanced_loader.py:62-63) behind the same name, on top of ``lr_voxel_dedup``.

MinkowskiEngine is not needed (and not available): the kept-index set -- first point of every occupied cell, ascending --
is computed by a hash-grid HIP kernel (csrc/lr_voxel.hip).  No CPU fallback.
"""
import torch

from . import _ext
from .matching import _device, _stream

VOXEL_SIZE = 0.3            # Experiments/dataloader/generic_balanced_loader.py voxel_size of the balanced loaders (config.voxel_size)


def sparse_quantize(coordinates, return_index=True):
    """ME.utils.sparse_quantize(coordinates, return_index=True): (unique integer cells [M,3] int32, index [M] int64), both on
    the HIP device; ``coordinates`` is [N,3] (any float dtype; evaluated in float64 like the reference's numpy clouds)."""
    dev = _device()
    c = torch.as_tensor(coordinates).to(device=dev, dtype=torch.float64).contiguous()
    n = int(c.shape[0])
    L = _ext.lib()
    scratch = torch.empty(int(L.lr_voxel_dedup_scratch_bytes(n)), dtype=torch.uint8, device=dev)
    sel = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    cells = torch.empty((max(n, 1), 3), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    _ext.check(L.lr_voxel_dedup(c.data_ptr(), n, sel.data_ptr(), cnt.data_ptr(), cells.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream()))
    m = int(cnt.item())
    if return_index:
        return cells[:m], sel[:m].long()
    return cells[:m]


def voxel_downsample(xyz, voxel_size=VOXEL_SIZE):
    """xyz [N,3] float64 (a cloud of the reference's cache) -> (xyz[sel] as float32 device tensor, sel int64 device tensor):
    what generic_balanced_loader.py:62-66,102 hands to the network and, through it, to FR()."""
    dev = _device()
    x = torch.as_tensor(xyz).to(device=dev, dtype=torch.float64)
    _, sel = sparse_quantize(x / voxel_size, return_index=True)
    return x[sel].float().contiguous(), sel
