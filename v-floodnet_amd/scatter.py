"""``scatter_mean``: the ``torch_scatter`` operator the reference imports (``FeatureBank.py:5,78,92``).

    scatter_mean(src[D,S], index int64[D,S], dim=1, out=out[D,B]) -> out

with torch-scatter 2.0.8 semantics for ``out=``: ``out += scatter_add(src)``, ``count =
scatter_add(ones)``, ``count.clamp_(min=1)``, ``out /= count``.  The reference always passes a
row-broadcast index (``idx.unsqueeze(0).expand(D, -1)``), which is the only form the HIP kernel
implements.  Sums run in ascending source order (deterministic, unlike the float-atomic CUDA kernel of
torch_scatter).  An out-of-range index or an index that is not row-broadcast is detected BY THE KERNEL (the element
is skipped, a sticky device flag is set).  By default (``STRICT``, ``VFN_SCATTER_STRICT=1``) the operator reads that flag
back and raises from the offending call -- one host synchronisation per call, next to the three or four
``FeatureBank.update`` in the reference already has per object (``nonzero`` / ``unique``, ``FeatureBank.py:73,85,125``); the
library's own ``FeatureBank`` does not go through this module (its merge runs in ``csrc/bank.hip`` with device-side
bookkeeping), so the hot path pays nothing.  ``VFN_SCATTER_STRICT=0``: the call never synchronises and the caller polls
``check_status()`` wherever it synchronises anyway -- torch_scatter's CUDA kernel reports the same conditions through a
device-side assert, i.e. asynchronously too.
"""
import os

import torch

from . import ops

STRICT = os.environ.get('VFN_SCATTER_STRICT', '1') != '0'     # raise from the call that met a bad index (a sync per call)
_status = {}          # device -> int32[1], sticky flags written by the kernel (zero at rest)


def _status_word(device):
    key = str(device)
    if key not in _status:
        _status[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return _status[key]


def check_status(device=None):
    """Raise if a ``scatter_mean`` call since the last check met an index outside ``out.shape[1]`` (that element was
    skipped) or a materialised index whose rows differ.  The operator itself never synchronises the host: torch_scatter's
    CUDA kernel reports the same conditions through a device-side assert, i.e. also at the next synchronisation.  Call this
    wherever the host synchronises anyway (it reads one int: one synchronisation per call)."""
    for key, word in _status.items():
        if device is not None and key != str(device):
            continue
        flags = int(word.item())
        if flags:
            word.zero_()
            what = []
            if flags & 1:
                what.append('an index outside [0, out.shape[1])')
            if flags & 2:
                what.append('an index that is not row-broadcast')
            raise RuntimeError('scatter_mean: ' + ' and '.join(what) + ' (reported by the device; the offending elements were skipped)')


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    if out is None:
        raise RuntimeError('scatter_mean: the HIP operator implements the out= form used by FeatureBank.update')
    if src.dim() != 2 or dim not in (1, -1):
        raise RuntimeError('scatter_mean: only src[D,S] with dim=1 is implemented (FeatureBank.py:78,92)')
    if not (src.is_cuda and out.is_cuda and index.is_cuda):
        raise RuntimeError('scatter_mean: tensors must live on the GPU (HIP kernel, no CPU fallback)')
    if src.dtype != torch.float32 or out.dtype != torch.float32:
        raise RuntimeError('scatter_mean: float32 only')
    if index.dim() not in (1, 2) or index.shape[-1] != src.shape[1] or (index.dim() == 2 and index.shape[0] not in (1, src.shape[0])):
        raise RuntimeError('scatter_mean: index / src size mismatch')
    if src.shape[1] == 0:
        return out            # nothing selected (an all-append frame): out/1 is out
    index = index.to(torch.int64)
    index_s0 = 0
    if index.dim() == 2 and index.shape[0] > 1 and index.stride(0) != 0:
        # a materialised [D,S] index must still be row-constant: the kernel compares every row with row 0
        if index.stride(1) != 1:
            index = index.contiguous()
        index_s0 = index.stride(0)
    else:
        index = (index[0] if index.dim() == 2 else index).contiguous()
    # range and row-constancy are checked by the kernel (sticky device flags, read by check_status)
    ops.scatter_mean_checked_launch(src, index, index_s0, out, _status_word(src.device))
    if STRICT:
        check_status(src.device)
    return out
