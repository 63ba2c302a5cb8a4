"""Each HIP kernel against the plain torch CPU fp32 form of the same operator."""
import math

import numpy as np
import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize('with_mask,H0,W0', [(False, 30, 43), (True, 30, 43), (True, 32, 48), (False, 17, 70)])
def test_stem(gpu, with_mask, H0, W0):
    from vfloodnet_amd import ops
    from vfloodnet_amd.engine import pad_divide_by
    g = torch.Generator().manual_seed(3)
    N = 2 if with_mask else 1
    frame = torch.rand(3, H0, W0, generator=g)
    mask = torch.rand(N, H0, W0, generator=g) if with_mask else None
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    wm = torch.randn(64, 1, 7, 7, generator=g) / 7
    wo = torch.randn(64, 1, 7, 7, generator=g) / 7
    scale = 1 + 0.1 * torch.randn(64, generator=g)
    shift = 0.1 * torch.randn(64, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    pad, Hp, Wp = pad_divide_by(H0, W0)
    fp = F.pad(frame.unsqueeze(0), pad)
    f = (fp - mean) / std
    x = F.conv2d(f.expand(N, -1, -1, -1), w, stride=2, padding=3)
    if with_mask:
        mp = F.pad(mask.unsqueeze(1), pad)
        x = x + F.conv2d(mp, wm, stride=2, padding=3) + F.conv2d((1 - mp).clamp(0, 1), wo, stride=2, padding=3)
    ref = F.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    ws = [w, wm, wo] if with_mask else [w]
    wp = ops.pack_stem_weight(ws).to(gpu)
    out = torch.empty(N, Hp // 2, Wp // 2, 64, device=gpu)
    d = ops.make_stem_desc(frame.to(gpu), mask.to(gpu) if with_mask else None, wp, scale.to(gpu), shift.to(gpu), out,
                           mean.flatten().tolist(), std.flatten().tolist(), N, H0, W0, pad, Hp, Wp)
    ops.stem_launch(d)
    torch.cuda.synchronize()
    err = (nchw(out.cpu()) - ref).abs().max().item()
    assert err < 2e-4, err


def test_maxpool(gpu):
    from vfloodnet_amd import ops
    x = torch.randn(2, 64, 13, 18)
    ref = F.max_pool2d(x, 3, 2, 1)
    out = torch.empty(2, ref.shape[2], ref.shape[3], 64, device=gpu)
    ops.maxpool3x3s2(nhwc(x).to(gpu), out)
    assert torch.equal(nchw(out.cpu()), ref)


def test_upsample_add(gpu):
    from vfloodnet_amd import ops
    pm = torch.randn(2, 8, 5, 7)
    s = torch.randn(1, 8, 10, 14)
    ref = s + F.interpolate(pm, scale_factor=2, mode='bilinear', align_corners=False)
    out = torch.empty(2, 10, 14, 8, device=gpu)
    ops.upsample2x_add(nhwc(s).to(gpu), nhwc(pm).to(gpu), out, True)
    assert (nchw(out.cpu()) - ref).abs().max() < 1e-5


def test_decoder_tail(gpu):
    """rough/uncertainty, local window stats and the final logits against AFB_URR.py:214-237 in torch."""
    from vfloodnet_amd import ops
    g = torch.Generator().manual_seed(5)
    K, h, w = 2, 12, 18           # 1/4-res grid; r1 grid is 2h x 2w; frame is 4h x 4w
    p = torch.randn(K, 2, h, w, generator=g) * 2
    r1 = torch.rand(1, 64, 2 * h, 2 * w, generator=g)
    q = torch.randn(K, 2, 2 * h, 2 * w, generator=g)
    # reference
    pu = F.interpolate(p, scale_factor=2, mode='bilinear', align_corners=False)
    rough = F.softmax(pu, dim=1)[:, 1].view(1, K, 2 * h, 2 * w)
    rough = F.softmax(rough, dim=1)
    top, _ = rough.topk(k=2, dim=1)
    unc = torch.exp(1 - top[:, 0] / (top[:, 1] + 1e-8)).unsqueeze(1)
    rg = rough.view(K, 1, 2 * h, 2 * w)
    r1e = r1.expand(K, -1, -1, -1)
    r1_local = F.avg_pool2d(r1e * rg, 7, 1, 3) / (F.avg_pool2d(rg, 7, 1, 3) + 1e-8)
    conf = F.max_pool2d(rg, 7, 1, 3)
    p2 = pu + unc.expand(-1, K, -1, -1).reshape(K, 1, 2 * h, 2 * w) * (conf * q)
    out = F.softmax(F.interpolate(p2, scale_factor=2, mode='bilinear', align_corners=False), dim=1)[:, 1]
    out = out.clamp(1e-7, 1 - 1e-7)
    score = torch.log(out / (1 - out))
    pad = (3, 2, 1, 2)           # lw, uw, lh, uh
    H0, W0 = 4 * h - 3, 4 * w - 5
    score_ref = score[:, pad[2]:pad[2] + H0, pad[0]:pad[0] + W0]
    # HIP
    d = lambda t: t.to(gpu)
    p_up = torch.empty(K, 2 * h, 2 * w, 2, device=gpu)
    rough_d = torch.empty(K, 2 * h, 2 * w, device=gpu)
    unc_d = torch.empty(2 * h, 2 * w, device=gpu)
    ops.rough_uncertainty(d(nhwc(p)), p_up, rough_d, unc_d)
    assert (nchw(p_up.cpu()) - pu).abs().max() < 1e-5
    assert (rough_d.cpu() - rough[0]).abs().max() < 1e-6
    assert (unc_d.cpu() - unc[0, 0]).abs().max() < 1e-5
    hs = torch.empty(K, 2 * h, 2 * w, 64, device=gpu)
    hr = torch.empty(K, 2 * h, 2 * w, device=gpu)
    hm = torch.empty(K, 2 * h, 2 * w, device=gpu)
    lm = torch.empty(K, 2 * h, 2 * w, 64, device=gpu)
    cf = torch.empty(K, 2 * h, 2 * w, device=gpu)
    ops.local_stats(d(nhwc(r1)), rough_d, hs, hr, hm, lm, cf)
    assert (nchw(lm.cpu()) - r1_local).abs().max() < 1e-5
    assert (cf.cpu() - conf[:, 0]).abs().max() < 1e-6
    sc = torch.empty(1, K, H0, W0, device=gpu)
    ops.final_logits(p_up, unc_d, cf, d(nhwc(q)), sc, pad, H0, W0)
    assert (sc.cpu()[0] - score_ref).abs().max() < 2e-4


@pytest.mark.parametrize('Hi,Wi,Ho,Wo', [(30, 53, 48, 85), (108, 192, 48, 85), (48, 85, 48, 85)])
def test_resize_ops(gpu, Hi, Wi, Ho, Wo):
    from vfloodnet_amd import ops
    g = torch.Generator().manual_seed(7)
    x = torch.rand(1, 3, Hi, Wi, generator=g)
    ref = F.interpolate(x, size=[Ho, Wo], mode='bicubic', align_corners=False)
    out = ops.resize_bicubic(x.to(gpu), Ho, Wo)
    assert (out.cpu() - ref).abs().max() < 1e-5
    refn = F.interpolate(x, size=[Ho, Wo], mode='nearest')
    outn = ops.resize_nearest(x.to(gpu), Ho, Wo)
    assert torch.equal(outn.cpu(), refn)
    score = torch.randn(1, 2, Hi, Wi, generator=g) * 3
    pm = ops.softmax_objects(score.to(gpu))
    pref = F.softmax(score, dim=1)
    assert (pm.cpu() - pref).abs().max() < 1e-6
    lab = ops.resize_argmax(pm, Ho, Wo).cpu()
    r = F.interpolate(pref, size=[Ho, Wo], mode='bicubic', align_corners=False)[0]
    labref = r.argmax(0).to(torch.uint8)
    margin = (r[0] - r[1]).abs()
    assert torch.equal(lab[margin > 1e-5], labref[margin > 1e-5])


def _bank(gpu, B, HW, seed=0, cap_extra=300):
    g = torch.Generator().manual_seed(seed)
    K = torch.randn(2, B, 128, generator=g)
    V = torch.randn(2, B, 512, generator=g)
    return K, V


@pytest.mark.parametrize('B,HW', [(60, 60), (1000, 150), (5000, 1620)])
def test_memory_read(gpu, B, HW):
    """Two-pass fused memory read vs softmax/matmul in torch (AFB_URR.py:144-146,163-174)."""
    from vfloodnet_amd.feature_bank import FeatureBank
    from vfloodnet_amd.engine import Engine
    import types
    g = torch.Generator().manual_seed(B)
    keys = [torch.randn(128, B, generator=g) for _ in range(2)]
    vals = [torch.randn(512, B, generator=g) for _ in range(2)]
    kvq = torch.randn(1, HW, 640, generator=g)
    kvq[..., :128] *= 1.5
    fb = FeatureBank(2, 250000, gpu)
    fb._hw = HW
    # init with B columns although HW differs: allocate by hand
    fb._alloc(HW, B)
    fb._write_columns([k.to(gpu) for k in keys], [v.to(gpu) for v in vals], [0, 0], 0, 0.0)
    fb._set_lengths([B, B])
    fb._ibuf[:, :B, 1] = 0.5
    plan = types.SimpleNamespace(HW=HW, kv_q=kvq.to(gpu), ml=torch.empty(2, HW, 2, device=gpu),
                                 ml_part=torch.empty(2, 256, HW, 2, device=gpu), work=torch.zeros(4, dtype=torch.int32, device=gpu),
                                 o_part=torch.empty(2, 20, HW, 512, device=gpu),
                                 dec_in=torch.empty(2, HW, 512, device=gpu))
    Engine._memory_read(types.SimpleNamespace(mode=0), plan, fb, True)
    torch.cuda.synchronize()
    out = plan.dec_in.cpu()
    q_in = kvq[0, :, :128].t().unsqueeze(0)
    for i in range(2):
        p = torch.matmul(keys[i].t(), q_in) / math.sqrt(128)
        p = F.softmax(p, dim=1)
        mem = torch.matmul(vals[i], p)[0]                        # [512, HW]
        assert (out[i, :, :512].t() - mem).abs().max() < 5e-5 * max(1, mem.abs().max().item())
        cnt = (p > 1e-3).float().sum(dim=2)[0]
        info_ref = 0.5 + torch.log(cnt + 1)
        got = fb.info[i][:, 1].cpu()
        # a count may differ by one where p is within rounding of the threshold
        near = ((p[0] - 1e-3).abs() < 1e-7).any(dim=1)
        assert (got[~near] - info_ref[~near]).abs().max() < 1e-5
    assert int(fb._cnt.abs().sum()) == 0


def _rb(t):
    return t.bfloat16().float()


def _mm_reduced(a, b, mode):
    """f64 value of what the reduced-precision contraction forms: mode 1 = bf16-rounded operands;
    mode 2 (bf16x3) = (ah+al)(bh+bl) - al*bl."""
    if mode == 1:
        return torch.matmul(_rb(a).double(), _rb(b).double())
    ah, bh = _rb(a), _rb(b)
    al, bl = _rb(a - ah), _rb(b - bh)
    return torch.matmul(ah.double() + al.double(), bh.double() + bl.double()) - torch.matmul(al.double(), bl.double())


@pytest.mark.parametrize('mode', [1, 2])
@pytest.mark.parametrize('B,HW', [(60, 60), (1000, 150), (5000, 1620)])
def test_memory_read_reduced_precision(gpu, B, HW, mode):
    """bf16 / bf16x3 memory read against an f64 evaluation of the same rounded / split operands (scores,
    then P and V); softmax in between as the kernel does it (f32 statistics of the reduced-precision scores)."""
    from vfloodnet_amd.feature_bank import FeatureBank
    from vfloodnet_amd.engine import Engine
    import types
    g = torch.Generator().manual_seed(B + mode)
    keys = [torch.randn(128, B, generator=g) for _ in range(2)]
    vals = [torch.randn(512, B, generator=g) for _ in range(2)]
    kvq = torch.randn(1, HW, 640, generator=g)
    fb = FeatureBank(2, 250000, gpu)
    fb._hw = HW
    fb._alloc(HW, B)
    fb._write_columns([k.to(gpu) for k in keys], [v.to(gpu) for v in vals], [0, 0], 0, 0.0)
    fb._set_lengths([B, B])
    plan = types.SimpleNamespace(HW=HW, kv_q=kvq.to(gpu), ml=torch.empty(2, HW, 2, device=gpu),
                                 ml_part=torch.empty(2, 256, HW, 2, device=gpu), work=torch.zeros(4, dtype=torch.int32, device=gpu),
                                 o_part=torch.empty(2, 20, HW, 512, device=gpu),
                                 dec_in=torch.empty(2, HW, 512, device=gpu))
    Engine._memory_read(types.SimpleNamespace(mode=mode), plan, fb, True)
    torch.cuda.synchronize()
    out = plan.dec_in.cpu()
    q_in = kvq[0, :, :128].t()
    for i in range(2):
        s_ = _mm_reduced(keys[i].t(), q_in, mode) / math.sqrt(128)
        p = F.softmax(s_, dim=0).float()
        mem = _mm_reduced(vals[i], p, mode).float()             # [512, HW]
        exact = torch.matmul(vals[i].double(), F.softmax(torch.matmul(keys[i].t().double(), q_in.double()) / math.sqrt(128), dim=0)).float()
        err = (out[i].t() - mem).abs().max().item()
        assert err < 1e-4 * max(1, mem.abs().max().item()), err
        dev_exact = (out[i].t() - exact).abs().max().item() / max(1, exact.abs().max().item())
        assert dev_exact < (5e-2 if mode == 1 else 2e-4), dev_exact


def test_scatter_mean(gpu):
    from vfloodnet_amd import scatter_mean
    g = torch.Generator().manual_seed(11)
    D, S, B = 128, 200, 50
    src = torch.randn(D, S, generator=g)
    idx = torch.randint(0, B, (S,), generator=g)
    out0 = torch.zeros(D, B)
    ref = out0.clone()
    ref.scatter_add_(1, idx.unsqueeze(0).expand(D, -1), src)
    cnt = torch.zeros(D, B).scatter_add_(1, idx.unsqueeze(0).expand(D, -1), torch.ones(D, S)).clamp_(min=1)
    ref = ref / cnt
    out = torch.zeros(D, B, device=gpu)
    scatter_mean(src.to(gpu), idx.to(gpu).unsqueeze(0).expand(D, -1), dim=1, out=out)
    assert (out.cpu() - ref).abs().max() < 1e-5
    # empty selection (all-append frames): no-op
    out2 = torch.zeros(D, B, device=gpu)
    scatter_mean(src[:, :0].to(gpu), idx[:0].to(gpu).unsqueeze(0).expand(D, -1), dim=1, out=out2)
    assert float(out2.abs().sum()) == 0.0


def test_device_ccl_matches_host_and_oracle(gpu):
    """vfn_postprocess_pred_device_u8 == the host routine == the oracle's postprocessing_pred."""
    from vfloodnet_amd import ops
    from vfloodnet_amd.data import postprocessing_pred
    from oracle import afb_urr_ref as O
    rng = np.random.RandomState(3)
    cases = [np.zeros((20, 30), np.uint8), np.ones((20, 30), np.uint8)]
    a = np.zeros((24, 40), np.uint8); a[2:6, 3:9] = 1; a[10:20, 12:30] = 1; a[7, 9] = 1
    cases.append(a)
    b = np.zeros((9, 9), np.uint8); b[1:4, 1:4] = 1
    cases.append(b)
    # two blobs of EQUAL size: the first in raster order wins
    c = np.zeros((12, 12), np.uint8); c[1:3, 1:3] = 1; c[8:10, 5:7] = 1
    cases.append(c)
    for thr in (0.3, 0.45, 0.55, 0.62, 0.7):
        cases.append((rng.rand(96, 130) > thr).astype(np.uint8))
    cases.append((rng.rand(480, 854) > 0.58).astype(np.uint8))
    # spiral / snake: long dependency chains for the union-find
    s_ = np.zeros((64, 64), np.uint8)
    for r in range(0, 64, 4):
        s_[r, :] = 1
        s_[r:r + 4, 63 if (r // 4) % 2 == 0 else 0] = 1
    cases.append(s_)
    # run-based merging: diagonal-only connectivity, checkerboards, runs crossing / ending at the 64-pixel segment
    # boundaries of a row, single-pixel runs touching only by a corner
    d = np.zeros((70, 200), np.uint8)
    for k in range(70):
        d[k, k] = 1; d[k, 199 - k] = 1                     # two diagonals that cross (one component)
    d[5:9, 100:140] = 1
    cases.append(d)
    cases.append((np.indices((33, 131)).sum(0) % 2).astype(np.uint8))           # checkerboard: all 8-connected
    e = np.zeros((10, 300), np.uint8)
    e[2, 60:70] = 1; e[3, 70:130] = 1; e[4, 127:129] = 1; e[5, 129] = 1; e[6, 63] = 1; e[7, 64:128] = 1; e[8, 128] = 1
    cases.append(e)
    f_ = np.zeros((40, 64 * 3 + 5), np.uint8); f_[::2, :] = 1; f_[1::4, 63] = 1; f_[3::4, 64] = 1   # comb joined at a seam
    cases.append(f_)
    for thr in (0.4, 0.5, 0.6):
        cases.append((rng.rand(61, 257) > thr).astype(np.uint8))
    for x in cases:
        ref = O.postprocessing_pred(x.copy())
        assert np.array_equal(postprocessing_pred(x), ref)
        out = ops.postprocess_pred_device(torch.from_numpy(x).to(gpu)).cpu().numpy()
        assert np.array_equal(out, ref), x.shape


def test_to_tensor_device_bit_exact(gpu):
    """vfn_to_tensor_u8 == torchvision ToTensor (uint8 HWC -> float CHW / 255), bit for bit."""
    from vfloodnet_amd import ops
    from vfloodnet_amd.dataset import to_tensor
    g = torch.Generator().manual_seed(5)
    img = torch.randint(0, 256, (37, 53, 3), generator=g, dtype=torch.uint8)
    img[0, :, 0] = torch.arange(53, dtype=torch.uint8)        # includes 0 ... and 255 below
    img[1, 0, :] = 255
    got = ops.to_tensor_device(img.to(gpu)).cpu()
    assert torch.equal(got, to_tensor(img.numpy()))


@pytest.mark.parametrize('labels', ['two', 'three', 'no_background', 'single', 'gap'])
def test_overlay_device_matches_reference(gpu, labels):
    """vfn_overlay_u8 == the REFERENCE's add_overlay + the uint8 / BGR handling of save_overlay (myutils/data.py:56-84),
    byte for byte, on outputs the reference itself produced (oracle/gen_image_seg_golden.py)."""
    import os
    import numpy as np
    from vfloodnet_amd import ops
    from vfloodnet_amd.data import color_palette
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'overlay_cases.npz'))
    frame, mask = torch.from_numpy(g['frame']), torch.from_numpy(g['mask_' + labels])
    got = ops.overlay_device(frame.to(gpu), mask.to(gpu), color_palette).cpu().numpy()
    assert np.array_equal(got, g['bgr_out_' + labels][..., ::-1])


def test_overlay_device_on_reference_loop_output(gpu):
    """The overlay PNG the reference's main() wrote for frame 1 (tests/golden/main_loop_120x200.npz) from the
    reference's own frame and label map: byte-equal."""
    import os
    import numpy as np
    from vfloodnet_amd import ops
    from vfloodnet_amd.data import color_palette
    from vfloodnet_amd.dataset import to_tensor
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'main_loop_120x200.npz'))
    W = int(g['shape'][1])
    lab = np.unpackbits(g['labels'], axis=-1)[1, :, :W]
    frame = to_tensor(np.ascontiguousarray(g['frames_u8'][1].transpose(1, 2, 0)))
    got = ops.overlay_device(frame.to(gpu), torch.from_numpy(np.ascontiguousarray(lab)).to(gpu), color_palette).cpu().numpy()
    assert np.array_equal(got, g['overlay1'])


def test_scatter_mean_rejects_out_of_range_index(gpu, monkeypatch):
    """Validation happens on the device: the offending element is skipped, the others are scattered.  Default (strict): the
    call that met the bad index raises (a drop-in for torch_scatter must not produce a wrong merge silently);
    ``VFN_SCATTER_STRICT=0``: no host synchronisation inside the operator (as torch_scatter's device-side assert) and
    ``scatter.check_status`` raises."""
    from vfloodnet_amd import scatter_mean, scatter
    src = torch.ones(4, 6, device=gpu)
    assert scatter.STRICT
    for bad in (5, -1):
        out = torch.zeros(4, 5, device=gpu)
        idx = torch.tensor([0, 1, 2, bad, 3, 4], device=gpu).unsqueeze(0).expand(4, 6)
        with pytest.raises(RuntimeError, match='outside'):
            scatter_mean(src, idx, dim=1, out=out)
        scatter.check_status(gpu)                                   # the flag is cleared by the report
        assert torch.equal(out.cpu(), torch.ones(4, 5))             # the five valid targets got their element
    monkeypatch.setattr(scatter, 'STRICT', False)
    for bad in (5, -1):
        out = torch.zeros(4, 5, device=gpu)
        idx = torch.tensor([0, 1, 2, bad, 3, 4], device=gpu).unsqueeze(0).expand(4, 6)
        scatter_mean(src, idx, dim=1, out=out)
        with pytest.raises(RuntimeError, match='outside'):
            scatter.check_status(gpu)
        scatter.check_status(gpu)
        assert torch.equal(out.cpu(), torch.ones(4, 5))
    # a materialised [D,S] index whose rows differ is reported the same way; a row-constant one is accepted
    out = torch.zeros(4, 5, device=gpu)
    idx = torch.tensor([0, 1, 2, 2, 3, 4], device=gpu).unsqueeze(0).repeat(4, 1)
    scatter_mean(src, idx, dim=1, out=out)
    scatter.check_status(gpu)
    assert torch.equal(out.cpu(), torch.ones(4, 5))
    idx[2, 1] = 4
    scatter_mean(src, idx, dim=1, out=torch.zeros(4, 5, device=gpu))
    with pytest.raises(RuntimeError, match='row-broadcast'):
        scatter.check_status(gpu)
    monkeypatch.setattr(scatter, 'STRICT', True)
    with pytest.raises(RuntimeError, match='row-broadcast'):
        scatter_mean(src, idx, dim=1, out=torch.zeros(4, 5, device=gpu))


@pytest.mark.parametrize('K,h,w', [(2, 240, 432), (3, 37, 53), (1, 9, 70), (4, 50, 16), (2, 24, 17)])
def test_local_stats_fused_vs_torch_and_two_pass(gpu, K, h, w):
    """vfn_local_stats_f32 (one pass, no scratch) against avg_pool / max_pool in torch (AFB_URR.py:226-229) and against
    the two-pass kernels it replaces, at the full 1/2-resolution size and at ragged sizes (strip / row-chunk edges)."""
    from vfloodnet_amd import ops
    g = torch.Generator().manual_seed(K * 100 + h)
    r1 = torch.rand(1, 64, h, w, generator=g)
    rg = torch.softmax(torch.randn(1, K, h, w, generator=g) * 2, dim=1) if K > 1 else torch.rand(1, 1, h, w, generator=g)
    rg = rg.view(K, 1, h, w)
    ref_local = F.avg_pool2d(r1.expand(K, -1, -1, -1) * rg, 7, 1, 3) / (F.avg_pool2d(rg, 7, 1, 3) + 1e-8)
    ref_conf = F.max_pool2d(rg, 7, 1, 3)[:, 0]
    r1_d, rg_d = nhwc(r1).to(gpu), rg[:, 0].contiguous().to(gpu)
    lm = torch.empty(K, h, w, 64, device=gpu)
    cf = torch.empty(K, h, w, device=gpu)
    ops.local_stats(r1_d, rg_d, None, None, None, lm, cf)
    assert (nchw(lm.cpu()) - ref_local).abs().max() < 2e-5
    assert torch.equal(cf.cpu(), ref_conf)
    hs = torch.empty(K, h, w, 64, device=gpu); hr = torch.empty(K, h, w, device=gpu); hm = torch.empty(K, h, w, device=gpu)
    lm2 = torch.empty_like(lm); cf2 = torch.empty_like(cf)
    ops.local_stats_two_pass(r1_d, rg_d, hs, hr, hm, lm2, cf2)
    assert torch.equal(cf, cf2)
    assert (lm - lm2).abs().max() < 1e-6                     # same summation order: equal up to fma contraction


@pytest.mark.parametrize('N,h,w,cin', [(2, 30, 54, 256), (2, 25, 33, 32), (1, 7, 9, 256)])
def test_pred2_tap_gemm_matches_conv(gpu, N, h, w, cin):
    """pred2 / local_pred2 as 1x1 tap GEMM + gather == 3x3 convolution with two filters on relu(x) (AFB_URR.py:213,234),
    and == the direct two-filter kernel it replaces."""
    from vfloodnet_amd import ops
    from vfloodnet_amd.engine import Pred2Layer, choose_cfg, apply_choice
    g = torch.Generator().manual_seed(cin + h)
    conv = torch.nn.Conv2d(cin, 2, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(2, cin, 3, 3, generator=g) / (3 * cin ** 0.5))
        conv.bias.copy_(torch.randn(2, generator=g))
    x = torch.randn(N, cin, h, w, generator=g)
    ref = conv(F.relu(x)).detach()
    layer = Pred2Layer(conv, gpu)
    xd = nhwc(x).to(gpu)
    z = torch.empty(N, h, w, Pred2Layer.TAPS, device=gpu)
    d = ops.make_conv_desc(xd, layer.w, layer.cout, 1, 1, 1, 0, z, layer.scale, layer.shift, None, True, False)
    cfg = apply_choice(d, choose_cfg(d.M, layer.cout, cin, 0), None)
    ops.conv2d_launch(d, cfg, 0)
    out = torch.empty(N, h, w, 2, device=gpu)
    ops.pred2_gather(z, layer.bias, out)
    assert (nchw(out.cpu()) - ref).abs().max() < 2e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('case', [(2, 12, 20, 64, 64, 3, 1, True), (1, 13, 19, 128, 96, 3, 2, False), (2, 9, 14, 256, 256, 1, 1, False),
                                  (1, 16, 24, 256, 512, 1, 2, False), (2, 30, 40, 32, 32, 3, 1, True), (1, 25, 40, 1024, 640, 3, 1, False),
                                  (3, 50, 50, 64, 256, 1, 1, False), (1, 20, 30, 256, 32, 3, 1, True), (2, 11, 17, 64, 20, 1, 1, False)])
def test_conv_wgrad_implicit_gemm_vs_autograd(gpu, case):
    """vfn_conv_wgrad_f32 (weight gradient straight from the NHWC tensors, reduction over the pixels) against torch.autograd of
    F.conv2d in float64: with / without ReLU on the input, strides 1 / 2, a frozen-BatchNorm row scale, accumulation into an
    existing gradient, forced split factors (the fixed-order reduce) and run-to-run bit-reproducibility."""
    import torch.nn.functional as F
    from vfloodnet_amd import ops
    N, H, W, Cin, Cout, k, s, relu = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).double().requires_grad_()
    scale = 1 + 0.1 * torch.randn(Cout, generator=g)
    y = F.conv2d(F.relu(x.double()) if relu else x.double(), w, stride=s, padding=k // 2) * scale.double().view(1, -1, 1, 1)
    gy = torch.randn(y.shape, generator=g)
    (y * gy.double()).sum().backward()
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, -1)                       # packed (kh, kw, cin)
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(gpu)
    sc = scale.to(gpu)
    tol = 2e-4 * ref.abs().max().item()
    first = None
    for ks in (None, 1, 3):
        got = ops.conv_wgrad(xd, gd, k, s, k // 2, relu=relu, rowscale=sc, ksplit=ks)
        torch.cuda.synchronize()
        assert (got.cpu().double() - ref).abs().max().item() < tol, (ks, (got.cpu().double() - ref).abs().max().item(), tol)
        again = ops.conv_wgrad(xd, gd, k, s, k // 2, relu=relu, rowscale=sc, ksplit=ks)
        assert torch.equal(got, again)
        if first is None:
            first = got
    acc = first.clone()
    ops.conv_wgrad(xd, gd, k, s, k // 2, relu=relu, rowscale=sc, out=acc, accumulate=True)
    assert (acc.cpu().double() - 2 * ref).abs().max().item() < 2 * tol
    # the slices finished inside the launch (vfn_wgrad_desc.tile_counters; off by default: measured slower on the training step):
    # the same sums in the same order as the reduce launch, bit for bit, and the counters back at zero
    for ks in (3, 7):
        want = ops.conv_wgrad(xd, gd, k, s, k // 2, relu=relu, rowscale=sc, ksplit=ks)
        ops._WGRAD_INLAUNCH = True
        try:
            got = ops.conv_wgrad(xd, gd, k, s, k // 2, relu=relu, rowscale=sc, ksplit=ks)
            acc2 = want.clone()
            ops.conv_wgrad(xd, gd, k, s, k // 2, relu=relu, rowscale=sc, ksplit=ks, out=acc2, accumulate=True)
        finally:
            ops._WGRAD_INLAUNCH = False
        assert torch.equal(got, want), ks
        assert (acc2.cpu().double() - 2 * ref).abs().max().item() < 2 * tol
        counters = [v[1] for v in ops._wgrad_ws.values()]
        assert counters and all(int(c.abs().sum()) == 0 for c in counters)


@pytest.mark.parametrize('M,C,ld,idn', [(10000, 256, 256, True), (625, 1024, 1024, False), (2500, 64, 64, True), (40000, 64, 256, False),
                                        (157, 2048, 2048, True), (7, 128, 128, False), (3000, 2, 4, False), (1234, 96, 96, True)])
def test_one_launch_column_sums_vs_float64(gpu, M, C, ld, idn):
    """vfn_colsum_acc_f32 / vfn_bn_param_grads_acc_f32 with an arrival counter (one launch; the 64-channel slab form when C % 64 == 0,
    the block-row form otherwise) against float64 sums: strided rows, accumulation into an existing value, the counters back at
    zero, run-to-run bit-reproducibility."""
    from vfloodnet_amd import _lib
    L = _lib.lib()
    ptr, check, stream = _lib.ptr, _lib.check, _lib.stream
    gen = torch.Generator().manual_seed(M * 7 + C)
    x = torch.randn(M, ld, generator=gen).to(gpu)
    nb = 128
    part = torch.empty(2 * nb * C, device=gpu)
    cnt = torch.zeros(64, dtype=torch.int32, device=gpu)
    ref = x[:, :C].double().sum(0).cpu()
    tol = 1e-5 * float(x[:, :C].abs().double().sum(0).max())
    out = torch.full((C,), float('nan'), device=gpu)
    check(L.vfn_colsum_acc_f32(ptr(x), M, C, ld, ptr(part), nb, ptr(out), 0, ptr(cnt), stream()), 'colsum')
    assert (out.cpu().double() - ref).abs().max().item() < tol
    again = torch.empty(C, device=gpu)
    check(L.vfn_colsum_acc_f32(ptr(x), M, C, ld, ptr(part), nb, ptr(again), 0, ptr(cnt), stream()), 'colsum')
    assert torch.equal(out, again)
    check(L.vfn_colsum_acc_f32(ptr(x), M, C, ld, ptr(part), nb, ptr(again), 1, ptr(cnt), stream()), 'colsum')
    assert (again.cpu().double() - 2 * ref).abs().max().item() < 2 * tol
    assert int(cnt.abs().sum()) == 0
    if ld != C:
        return
    # frozen-BatchNorm parameter gradients: dbeta = sum g, dgamma = sum g * ((y - idn) - beta) / gamma
    g = torch.randn(M, C, generator=gen).to(gpu)
    y = x
    i = torch.randn(M, C, generator=gen).to(gpu) if idn else None
    beta = torch.randn(C, generator=gen).to(gpu)
    gamma = (1 + 0.2 * torch.rand(C, generator=gen)).to(gpu)
    yy = (y.double() - (i.double() if idn else 0) - beta.double()) / gamma.double()
    ref_b, ref_g = g.double().sum(0).cpu(), (g.double() * yy).sum(0).cpu()
    tol_g = 1e-5 * float((g.double() * yy).abs().sum(0).max())
    dg, db = torch.empty(C, device=gpu), torch.empty(C, device=gpu)
    check(L.vfn_bn_param_grads_acc_f32(ptr(g), ptr(y), ptr(i), ptr(beta), ptr(gamma), M, C, ptr(part), nb, ptr(dg), ptr(db), 0, ptr(cnt),
                                       stream()), 'bn')
    assert (db.cpu().double() - ref_b).abs().max().item() < 1e-5 * float(g.abs().double().sum(0).max())
    assert (dg.cpu().double() - ref_g).abs().max().item() < tol_g
    dg2, db2 = dg.clone(), db.clone()
    check(L.vfn_bn_param_grads_acc_f32(ptr(g), ptr(y), ptr(i), ptr(beta), ptr(gamma), M, C, ptr(part), nb, ptr(dg2), ptr(db2), 1, ptr(cnt),
                                       stream()), 'bn')
    assert (dg2.cpu().double() - 2 * ref_g).abs().max().item() < 2 * tol_g
    assert torch.equal(db2, 2 * db) and int(cnt.abs().sum()) == 0


@pytest.mark.parametrize('case', [(2, 24, 40, 256, 256, True), (1, 23, 37, 128, 256, False), (2, 6, 10, 256, 128, True), (3, 50, 50, 64, 64, False)])
def test_winograd_weight_gradient_vs_direct_and_autograd(gpu, case):
    """The weight gradient of a 3x3 / stride-1 / pad-1 convolution in the Winograd domain (ops.conv_wgrad_winograd: input transform,
    transform of the gradient tiles, the 36 component sums as one batched launch of the weight-gradient kernel, back-transform)
    against float64 autograd and against the direct implicit-GEMM form: ragged tile edges, ReLU on the input, a row scale,
    accumulation, run-to-run bit-reproducibility."""
    import torch.nn.functional as F
    from vfloodnet_amd import ops
    N, H, W, Cin, Cout, relu = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).double().requires_grad_()
    scale = 1 + 0.1 * torch.randn(Cout, generator=g)
    y = F.conv2d(F.relu(x.double()) if relu else x.double(), w, padding=1) * scale.double().view(1, -1, 1, 1)
    gy = torch.randn(y.shape, generator=g)
    (y * gy.double()).sum().backward()
    ref = w.grad.permute(0, 2, 3, 1).reshape(Cout, -1)                       # packed (kh, kw, cin)
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(gpu)
    sc = scale.to(gpu)
    tol = 5e-4 * ref.abs().max().item()                                      # (the transforms amplify f32 rounding ~10x over the direct sum)
    got = ops.conv_wgrad_winograd(xd, gd, relu=relu, rowscale=sc)
    direct = ops.conv_wgrad(xd, gd, 3, 1, 1, relu=relu, rowscale=sc)
    torch.cuda.synchronize()
    err = (got.cpu().double() - ref).abs().max().item()
    assert err < tol, (err, tol, (direct.cpu().double() - ref).abs().max().item())
    assert torch.equal(got, ops.conv_wgrad_winograd(xd, gd, relu=relu, rowscale=sc))
    acc = got.clone()
    ops.conv_wgrad_winograd(xd, gd, relu=relu, rowscale=sc, out=acc, accumulate=True)
    assert (acc.cpu().double() - 2 * ref).abs().max().item() < 2 * tol


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 96, 160, 3, True), (1, 50, 70, 5, True), (5, 64, 48, 3, False), (2, 33, 39, 5, False), (1, 16, 16, 3, True)])
def test_stem_weight_gradient_vs_autograd_and_the_generic_kernel(gpu, case, monkeypatch):
    """The 7x7 / stride-2 / pad-3 stems' weight gradient (vfn_stem_wgrad_f32: the operand columns are (kw, c) of one filter row, one pixel
    walk for all 49 taps; 3 planes for the query encoder, 5 for the memory encoder) against float64 autograd and against the generic
    implicit-GEMM kernel it replaces on this shape: odd sizes (a last pixel without a partner, segments that end inside a row), the
    frozen-BatchNorm row scale, accumulation, run-to-run bit-reproducibility."""
    import torch.nn.functional as F
    from vfloodnet_amd import ops
    N, H, W, C, with_scale = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, C, H, W, generator=g)
    w = (torch.randn(64, C, 7, 7, generator=g) / (C * 49) ** 0.5).double().requires_grad_()
    scale = 1 + 0.1 * torch.randn(64, generator=g)
    y = F.conv2d(x.double(), w, stride=2, padding=3)
    if with_scale:
        y = y * scale.double().view(1, -1, 1, 1)
    gy = torch.randn(y.shape, generator=g)
    (y * gy.double()).sum().backward()
    ref = w.grad.permute(0, 2, 3, 1).reshape(64, -1)                          # packed (kh, kw, c)
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(gpu)
    sc = scale.to(gpu) if with_scale else None
    assert ops._STEM_WGRAD
    got = ops.conv_wgrad(xd, gd, 7, 2, 3, rowscale=sc)
    monkeypatch.setattr(ops, '_STEM_WGRAD', False)
    generic = ops.conv_wgrad(xd, gd, 7, 2, 3, rowscale=sc)
    monkeypatch.setattr(ops, '_STEM_WGRAD', True)
    torch.cuda.synchronize()
    tol = 2e-5 * ref.abs().max().item() * max(1.0, (N * y.shape[2] * y.shape[3]) ** 0.5 / 30)
    err, err_g = (got.cpu().double() - ref).abs().max().item(), (generic.cpu().double() - ref).abs().max().item()
    assert err < tol and err_g < tol, (err, err_g, tol)
    assert not torch.equal(got, generic) or N * H * W < 2000                   # (two kernels, two summation orders: the new one did run)
    assert torch.equal(got, ops.conv_wgrad(xd, gd, 7, 2, 3, rowscale=sc))
    acc = got.clone()
    ops.conv_wgrad(xd, gd, 7, 2, 3, rowscale=sc, out=acc, accumulate=True)
    assert (acc.cpu().double() - 2 * ref).abs().max().item() < 2 * tol


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 17, 23, 64, True), (1, 8, 8, 4, False), (5, 40, 36, 64, True), (1, 7, 9, 6, True), (1, 2, 3, 8, False)])
def test_maxpool_backward_vs_autograd_and_the_scalar_kernel(gpu, case):
    """MaxPool2d(3, 2, 1) backwards (vfn_maxpool3x3s2_backward_f32): the four-channel kernel (one thread per 2 x 2 input block, the 5 x 5
    patch loaded once) against float64 autograd -- inputs quantised so that windows hold ties (PyTorch routes the gradient to the FIRST
    maximum in row-major order) -- with the second gradient and the ReLU mask of the encoders' r1, odd sizes; and bit-identical to the
    scalar kernel it replaces (reached through a 4-byte-misaligned view; 6 channels take it anyway)."""
    import torch.nn.functional as F
    from vfloodnet_amd import _lib
    from vfloodnet_amd._lib import ptr, stream, check
    N, H, W, C, with_add = case
    gen = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = (torch.randn(N, C, H, W, generator=gen) * 2).round() / 2                 # multiples of 0.5: plenty of equal values
    xr = x.double().requires_grad_()
    y = F.max_pool2d(F.relu(xr), 3, 2, 1)
    g = torch.randn(y.shape, generator=gen)
    add = torch.randn(N, C, H, W, generator=gen) if with_add else None
    (y * g.double()).sum().backward()
    # the kernel's contract: x is the ReLU's OUTPUT (r1), the pooled gradient plus ``add`` is masked where x <= 0
    r1 = F.relu(x)
    ref = xr.grad.clone()
    if add is not None:
        ref = ref + add.double() * (x > 0)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    L = _lib.lib()

    def run(misalign):
        def dev(t):
            if t is None:
                return None
            flat = torch.empty(t.numel() + 4, device=gpu)
            v = flat[1:1 + t.numel()] if misalign else flat[4:4 + t.numel()]
            v.copy_(nhwc(t).reshape(-1))
            return v
        xd, gd, ad = dev(r1), dev(g), dev(add)
        out = torch.empty(N * H * W * C + 4, device=gpu)
        ov = out[1:1 + N * H * W * C] if misalign else out[4:4 + N * H * W * C]
        check(L.vfn_maxpool3x3s2_backward_f32(ptr(xd), ptr(gd), ptr(ov), N, H, W, C, ptr(ad), 1, stream()), 'vfn_maxpool3x3s2_backward_f32')
        torch.cuda.synchronize()
        return ov.view(N, H, W, C).permute(0, 3, 1, 2).cpu()
    fast, scalar = run(False), run(True)
    assert torch.equal(fast, scalar)
    assert (fast.double() - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
