"""HIP implicit-GEMM conv vs torch CPU fp32 ``F.conv2d`` (the op's fp32 reference)."""
import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu

# (N, H, W, Cin, Cout, k, stride, relu_in, relu_out, residual)
CASES = [
    (1, 12, 20, 64, 64, 1, 1, False, True, False),
    (2, 12, 20, 64, 256, 1, 1, False, False, True),
    (1, 13, 19, 128, 128, 3, 2, False, True, False),
    (2, 9, 14, 256, 256, 3, 1, True, False, True),
    (1, 16, 24, 256, 512, 1, 2, False, False, False),
    (2, 7, 11, 1024, 640, 3, 1, False, False, False),
    (2, 30, 40, 32, 32, 3, 1, True, False, True),
    (2, 10, 12, 256, 2, 3, 1, True, False, False),
    (1, 33, 47, 64, 96, 3, 1, False, True, True),
]


@pytest.mark.parametrize('case', CASES)
def test_conv_matches_torch_cpu(gpu, case):
    from vfloodnet_amd import ops, weights
    N, H, W, Cin, Cout, k, s, relu_in, relu_out, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = 1 + 0.1 * torch.randn(Cout, generator=g)
    shift = 0.1 * torch.randn(Cout, generator=g)
    xin = F.relu(x) if relu_in else x
    ref = F.conv2d(xin, w, stride=s, padding=k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res
    if relu_out:
        ref = F.relu(ref)
    scale_d, shift_d = scale.to(gpu), shift.to(gpu)      # descriptors hold raw pointers: keep the tensors alive
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    wp = ops.pad_rows(weights.pack_conv_weight(w)).to(gpu)
    resd = res.permute(0, 2, 3, 1).contiguous().to(gpu) if use_res else None
    for cfg, (bm_, bn_) in enumerate(ops.conv_cfg_tiles()):
        if wp.shape[0] < ((Cout + bn_ - 1) // bn_) * bn_:
            continue
        y = ops.conv2d_nhwc(xd, wp, Cout, k, k, s, k // 2, scale_d, shift_d, resd,
                            relu_in, relu_out, cfg=cfg)
        torch.cuda.synchronize()
        got = y.permute(0, 3, 1, 2).cpu()
        err = (got - ref).abs().max().item()
        assert err < 2e-4 * max(1.0, ref.abs().max().item()), f'cfg {cfg}: max err {err}'
    # split-K variants (partial slabs + fixed-order reduce)
    ws = torch.empty(8 * ref.numel(), device=gpu)
    y = torch.empty(N, ref.shape[2], ref.shape[3], Cout, device=gpu)
    d = ops.make_conv_desc(xd, wp, Cout, k, k, s, k // 2, y, scale_d, shift_d, resd, relu_in, relu_out)
    cnt = torch.zeros(4096, dtype=torch.int32, device=gpu)
    for ks in ops.valid_splits(d, 8)[1:]:
        outs = []
        for counters in (None, cnt if Cout % 64 == 0 else None):       # separate reduce launch / in-launch finish
            y.zero_()
            ops.set_splitk(d, ks, ws, counters=counters)
            for _ in range(2):                                          # twice: the counters must return to rest
                ops.conv2d_launch(d, 3)
            torch.cuda.synchronize()
            err = (y.permute(0, 3, 1, 2).cpu() - ref).abs().max().item()
            assert err < 2e-4 * max(1.0, ref.abs().max().item()), f'split {ks}: max err {err}'
            outs.append(y.clone())
        assert torch.equal(outs[0], outs[1]), 'in-launch finish must be bit-identical to the reduce launch'
        assert int(cnt.abs().sum()) == 0
    # tail split: only the tiles from an n-tile-aligned index on are cut along K
    tiles = ops.conv_cfg_tiles()
    bm, bn = tiles[3]
    n_t = (Cout + bn - 1) // bn
    m_t = (d.M + bm - 1) // bm
    for full_m in {1, m_t // 2, m_t - 1}:
        if 0 < full_m < m_t and len(ops.valid_splits(d, 4)) > 1:
            ks = ops.valid_splits(d, 4)[-1]
            y.zero_()
            ops.set_splitk(d, ks, ws, split_from=full_m * n_t, rows=d.M - full_m * bm)
            ops.conv2d_launch(d, 3)
            torch.cuda.synchronize()
            err = (y.permute(0, 3, 1, 2).cpu() - ref).abs().max().item()
            assert err < 2e-4 * max(1.0, ref.abs().max().item()), f'tail split from {full_m}: max err {err}'


BF16_CASES = [
    (1, 12, 20, 64, 64, 1, 1, False, True, False),
    (2, 12, 20, 64, 256, 1, 1, False, False, True),
    (1, 13, 19, 128, 128, 3, 2, False, True, False),
    (2, 9, 14, 256, 256, 3, 1, True, False, True),
    (2, 7, 11, 1024, 640, 3, 1, False, False, False),
    (1, 33, 47, 64, 96, 3, 1, False, True, True),
]


def _rb(t):
    return t.bfloat16().float()


def _reduced_reference(x, w, stride, pad, mode):
    """What the reduced-precision kernels compute, in f64: mode 1 = product of the bf16-rounded operands;
    mode 2 (bf16x3) = (xh+xl)*(wh+wl) - xl*wl with x = xh + xl + eps split into two bf16."""
    if mode == 1:
        return F.conv2d(_rb(x).double(), _rb(w).double(), stride=stride, padding=pad).float()
    xh, wh = _rb(x), _rb(w)
    xl, wl = _rb(x - xh), _rb(w - wh)
    full = F.conv2d((xh.double() + xl.double()), (wh.double() + wl.double()), stride=stride, padding=pad)
    return (full - F.conv2d(xl.double(), wl.double(), stride=stride, padding=pad)).float()


@pytest.mark.parametrize('mode', [1, 2])
@pytest.mark.parametrize('case', BF16_CASES)
def test_conv_reduced_precision_matches_emulation(gpu, case, mode):
    """bf16 / bf16x3 kernels against an f64 evaluation of exactly the products they form (operands rounded /
    split on the host the same way, nearest-even): only the f32 summation order differs."""
    from vfloodnet_amd import ops, weights
    N, H, W, Cin, Cout, k, s, relu_in, relu_out, use_res = case
    g = torch.Generator().manual_seed(hash(case) & 0xFFFF)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale = 1 + 0.1 * torch.randn(Cout, generator=g)
    shift = 0.1 * torch.randn(Cout, generator=g)
    xin = F.relu(x) if relu_in else x
    ref = _reduced_reference(xin, w, s, k // 2, mode) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    exact = F.conv2d(xin, w, stride=s, padding=k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if use_res:
        ref = ref + res
        exact = exact + res
    if relu_out:
        ref = F.relu(ref)
        exact = F.relu(exact)
    scale_d, shift_d = scale.to(gpu), shift.to(gpu)
    xd = x.permute(0, 2, 3, 1).contiguous().to(gpu)
    wp = ops.pad_rows(weights.pack_conv_weight(w)).to(gpu)
    resd = res.permute(0, 2, 3, 1).contiguous().to(gpu) if use_res else None
    tiles = ops.conv_cfg_tiles()
    tol = 2e-4 * max(1.0, ref.abs().max().item())
    for cfg in ops.BF16_CFGS:
        if wp.shape[0] < ((Cout + tiles[cfg][1] - 1) // tiles[cfg][1]) * tiles[cfg][1]:
            continue
        y = ops.conv2d_nhwc(xd, wp, Cout, k, k, s, k // 2, scale_d, shift_d, resd, relu_in, relu_out, cfg=cfg, mode=mode)
        torch.cuda.synchronize()
        got = y.permute(0, 3, 1, 2).cpu()
        err = (got - ref).abs().max().item()
        assert err < tol, f'cfg {cfg}: max err {err}'
    # filters converted on the host (vfn_conv_desc.w_packed): bit-identical to the on-the-fly conversion
    wlp = ops.pack_weights_lp(wp, mode)
    y2 = torch.empty_like(y)
    d2 = ops.make_conv_desc(xd, wp, Cout, k, k, s, k // 2, y2, scale_d, shift_d, resd, relu_in, relu_out)
    ops.use_packed_weights(d2, wlp)
    ops.conv2d_launch(d2, cfg, mode)
    torch.cuda.synchronize()
    assert torch.equal(y2, y), 'packed-weight path differs from the on-the-fly conversion'
    # distance from the exact f32 convolution: what the mode costs (bf16x3 ~2^-16 per product, bf16 ~2^-9)
    dev_exact = (got - exact).abs().max().item() / max(1.0, exact.abs().max().item())
    assert dev_exact < (2e-2 if mode == 1 else 1e-4), dev_exact
    ws = torch.empty(8 * ref.numel(), device=gpu)
    y = torch.empty(N, ref.shape[2], ref.shape[3], Cout, device=gpu)
    d = ops.make_conv_desc(xd, wp, Cout, k, k, s, k // 2, y, scale_d, shift_d, resd, relu_in, relu_out)
    for ks in ops.valid_splits(d, 8, mode=mode)[1:]:
        y.zero_()
        ops.set_splitk(d, ks, ws)
        ops.conv2d_launch(d, 3, mode=mode)
        torch.cuda.synchronize()
        err = (y.permute(0, 3, 1, 2).cpu() - ref).abs().max().item()
        assert err < tol, f'split {ks}: max err {err}'
