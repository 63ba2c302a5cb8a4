"""``scatter_mean``: the ``torch_scatter`` operator the reference imports (``FeatureBank.py:5,78,92``).

    scatter_mean(src[D,S], index int64[D,S], dim=1, out=out[D,B]) -> out

with torch-scatter 2.0.8 semantics for ``out=``: ``out += scatter_add(src)``, ``count =
scatter_add(ones)``, ``count.clamp_(min=1)``, ``out /= count``.  The reference always passes a
row-broadcast index (``idx.unsqueeze(0).expand(D, -1)``), which is the only form the HIP kernel
implements; anything else raises.  Sums run in ascending source order (deterministic, unlike the
float-atomic CUDA kernel of torch_scatter).
"""
import torch

from . import ops


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    if out is None:
        raise RuntimeError('scatter_mean: the HIP operator implements the out= form used by FeatureBank.update')
    if src.dim() != 2 or dim not in (1, -1):
        raise RuntimeError('scatter_mean: only src[D,S] with dim=1 is implemented (FeatureBank.py:78,92)')
    if not (src.is_cuda and out.is_cuda and index.is_cuda):
        raise RuntimeError('scatter_mean: tensors must live on the GPU (HIP kernel, no CPU fallback)')
    if src.dtype != torch.float32 or out.dtype != torch.float32:
        raise RuntimeError('scatter_mean: float32 only')
    if index.dim() == 2:
        if index.shape[0] > 1 and index.stride(0) != 0:
            # a materialised [D,S] index must still be row-constant
            if not bool((index == index[0:1]).all()):
                raise RuntimeError('scatter_mean: only a row-broadcast index is supported')
        index_row = index[0]
    else:
        index_row = index
    if index_row.shape[0] != src.shape[1]:
        raise RuntimeError('scatter_mean: index / src size mismatch')
    if src.shape[1] == 0:
        return out            # nothing selected (an all-append frame): out/1 is out
    index_row = index_row.to(torch.int64).contiguous()
    lo, hi = int(index_row.min()), int(index_row.max())       # (torch_scatter raises on an out-of-range index too)
    if lo < 0 or hi >= out.shape[1]:
        raise RuntimeError(f'scatter_mean: index range [{lo}, {hi}] outside out.shape[1] = {out.shape[1]}')
    ops.scatter_mean_launch(src, index_row, out)
    return out
