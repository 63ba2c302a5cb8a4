"""A deterministic stand-in for the pickled LinkNet of ``test_image_seg.py:133`` (un-vendored
``segmentation_models_pytorch`` + absent weights): any object with the smp ``predict`` API exercises the plumbing
around it.  Shared by ``oracle/gen_image_seg_golden.py`` (driving the REFERENCE's ``predict_pil``) and
``tests/test_image_seg_plumbing.py`` (driving ours), so both see the same "network"."""
import torch


class StandIn:
    """predict(x[1,3,416,416]) -> prob[1,1,416,416] in [0,1]: a smooth function of the de-normalised image with values
    on both sides of 0.5 (so that the bilinear resize back + ``round`` matter) and several disconnected blobs (so that
    ``postprocessing_pred`` matters)."""

    def predict(self, x):
        assert tuple(x.shape) == (1, 3, 416, 416) and x.dtype == torch.float32
        std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        g = (x.cpu() * std + mean)
        blue = g[:, 2:3] - 0.5 * (g[:, 0:1] + g[:, 1:2])
        blue = torch.nn.functional.avg_pool2d(blue, 9, 1, 4)
        return torch.sigmoid(40.0 * (blue - blue.median()))
