"""MI355X-native drop-in for V-FloodNet's video water-segmentation hot path.

Mirrors ``video_module.model`` (``AFB_URR``, ``FeatureBank``), the
``torch_scatter.scatter_mean`` call sites and the ``test_video_seg.main`` loop of
the reference; all math runs in hand-written HIP kernels for gfx950 behind the
C-ABI declared in ``include/vfn_hip.h``.
"""
from .model import AFB_URR            # noqa: F401
from .feature_bank import FeatureBank  # noqa: F401
from .scatter import scatter_mean      # noqa: F401

__all__ = ['AFB_URR', 'FeatureBank', 'scatter_mean']
