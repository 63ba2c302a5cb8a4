"""Size-independent properties of the hot-path kernels at BASELINE.json's full C2 sizes (480x854 -> 480x864,
HW = 1620, 100 000-entry bank), where the CPU oracle is too slow to be the checker."""
import math
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _conv(ops, weights, x, w, **kw):
    wp = ops.pad_rows(weights.pack_conv_weight(w.cpu())).to(x.device)
    return ops.conv2d_nhwc(x, wp, w.shape[0], w.shape[2], w.shape[3], 1, w.shape[2] // 2, cfg=kw.get('cfg', 8))


def test_conv_linearity_full_size(gpu):
    """conv(a*x + b*y) == a*conv(x) + b*conv(y) on the largest decoder layer (2 x 120 x 216 pixels, 3x3, 256 -> 256)."""
    from vfloodnet_amd import ops, weights
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn(2, 120, 216, 256, generator=g).to(gpu)
    y = torch.randn(2, 120, 216, 256, generator=g).to(gpu)
    w = (torch.randn(256, 256, 3, 3, generator=g) / 48).to(gpu)
    a, b = 0.75, -1.5
    lhs = _conv(ops, weights, a * x + b * y, w)
    rhs = a * _conv(ops, weights, x, w) + b * _conv(ops, weights, y, w)
    assert (lhs - rhs).abs().max().item() < 2e-4 * max(1.0, rhs.abs().max().item())
    # and the tile configuration is immaterial up to summation order
    alt = _conv(ops, weights, x, w, cfg=10)
    ref = _conv(ops, weights, x, w, cfg=8)
    assert (alt - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


def _memory_read(gpu, keys, vals, kvq, mode=0, update_bank=True):
    from vfloodnet_amd.feature_bank import FeatureBank
    from vfloodnet_amd.engine import Engine
    B, HW = keys[0].shape[1], kvq.shape[1]
    fb = FeatureBank(2, 250000, gpu)
    fb._hw = HW
    fb._alloc(HW, B)
    fb._write_columns([k.to(gpu) for k in keys], [v.to(gpu) for v in vals], [0, 0], 0, 0.0)
    fb._set_lengths([B, B])
    plan = types.SimpleNamespace(HW=HW, kv_q=kvq.to(gpu), ml=torch.empty(2, HW, 2, device=gpu),
                                 ml_part=torch.empty(2, 256, HW, 2, device=gpu), work=torch.zeros(4, dtype=torch.int32, device=gpu),
                                 o_part=torch.empty(2, 20, HW, 512, device=gpu),
                                 dec_in=torch.empty(2, HW, 512, device=gpu))
    Engine._memory_read(types.SimpleNamespace(mode=mode), plan, fb, update_bank)
    torch.cuda.synchronize()
    return plan.dec_in.clone(), fb


def test_memory_read_properties_full_bank(gpu):
    """At B = 100 000 entries / object, HW = 1620: (1) the read-out is a convex combination -- with all values equal to
    one vector it returns that vector; (2) permuting the bank permutes nothing in the output and permutes the hit
    counts with it; (3) sum of hit-count bumps == what the softmax implies for a few probed queries."""
    B, HW = 100000, 1620
    g = torch.Generator().manual_seed(9)
    keys = [torch.randn(128, B, generator=g) for _ in range(2)]
    kvq = torch.randn(1, HW, 640, generator=g)
    const = torch.randn(512, generator=g)
    vals = [const[:, None].expand(512, B).contiguous() for _ in range(2)]
    out, _ = _memory_read(gpu, keys, vals, kvq, update_bank=False)
    assert (out.cpu() - const[None, None, :]).abs().max().item() < 2e-5 * max(1.0, const.abs().max().item())

    vals = [torch.randn(512, B, generator=g) for _ in range(2)]
    out1, fb1 = _memory_read(gpu, keys, vals, kvq)
    perm = torch.randperm(B, generator=g)
    out2, fb2 = _memory_read(gpu, [k[:, perm] for k in keys], [v[:, perm] for v in vals], kvq)
    assert (out1 - out2).abs().max().item() < 5e-5 * max(1.0, out1.abs().max().item())
    for i in range(2):
        i1, i2 = fb1.info[i][:, 1].cpu(), fb2.info[i][:, 1].cpu()
        assert (i1[perm] - i2).abs().max().item() < 1e-5
        # probe: exact softmax for 8 queries on the host
        qs = torch.arange(0, HW, HW // 8)[:8]
        s = (keys[i].t().double() @ kvq[0, qs, :128].t().double()) / math.sqrt(128)
        p = torch.softmax(s, dim=0)
        ref = (vals[i].double() @ p).t().float()
        assert (out1[i, qs].cpu() - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())


def test_ccl_idempotent_and_largest_full_size(gpu):
    """postprocessing_pred at 480x854: idempotent, the result is a subset of the input water, it is one 8-connected
    component, and no other component is larger (checked with scipy on the host)."""
    from scipy import ndimage
    from vfloodnet_amd import ops
    g = torch.Generator().manual_seed(2)
    blob = torch.nn.functional.avg_pool2d(torch.rand(1, 1, 480, 854, generator=g), 9, 1, 4)[0, 0]
    pred = (blob > 0.5).to(torch.uint8)
    out = ops.postprocess_pred_device(pred.to(gpu))
    again = ops.postprocess_pred_device(out)
    assert torch.equal(out, again)
    o, pnp = out.cpu().numpy(), pred.numpy()
    assert ((o == 1) <= (pnp == 1)).all()
    lab, n = ndimage.label(pnp, structure=np.ones((3, 3)))
    sizes = ndimage.sum(pnp, lab, index=range(1, n + 1))
    lab_o, n_o = ndimage.label(o, structure=np.ones((3, 3)))
    assert n_o == 1 and int(o.sum()) == int(max(sizes))


def test_softmax_and_argmax_consistency_full_size(gpu):
    """softmax over objects sums to one; resize+argmax at identity size equals a plain argmax."""
    from vfloodnet_amd import ops
    g = torch.Generator().manual_seed(4)
    score = (3 * torch.randn(1, 2, 480, 854, generator=g)).to(gpu)
    pm = ops.softmax_objects(score)
    assert (pm.sum(1) - 1).abs().max().item() < 1e-6
    lab = ops.resize_argmax(pm, 480, 854)
    assert torch.equal(lab, pm[0].argmax(0).to(torch.uint8))


def test_scatter_mean_properties(gpu):
    """torch_scatter semantics on a full-size update: an identity index averages src into out/1 ... i.e.
    out <- (out + src) / 1; constant columns stay constant; untouched columns are unchanged."""
    from vfloodnet_amd import scatter_mean
    g = torch.Generator().manual_seed(6)
    D, S, Bn = 512, 1620, 100000
    src = torch.randn(D, S, generator=g).to(gpu)
    out = torch.zeros(D, Bn, device=gpu)
    idx = torch.randperm(Bn, generator=g)[:S].to(gpu)                   # distinct targets
    res = scatter_mean(src, idx.unsqueeze(0).expand(D, S), dim=1, out=out)
    assert torch.equal(res[:, idx], src)                                # one contribution each: mean == value
    mask = torch.ones(Bn, dtype=torch.bool, device=gpu)
    mask[idx] = False
    assert float(res[:, mask].abs().max()) == 0.0
    # all sources into one target: the mean of the sources (plus the zero already there, count = S)
    out2 = torch.zeros(D, 4, device=gpu)
    res2 = scatter_mean(src, torch.full((D, S), 2, dtype=torch.int64, device=gpu), dim=1, out=out2)
    assert (res2[:, 2] - src.mean(1)).abs().max().item() < 1e-5
