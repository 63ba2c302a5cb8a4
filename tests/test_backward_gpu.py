"""Backward pass, first slice: the decoder's global branch on the HIP path against torch.autograd on
the CPU oracle's decoder (oracle/afb_urr_ref.py::decoder_global, AFB_URR.py:209-212) in float64.

The forward runs through the product (``memorize`` -> bank -> ``segment``); the oracle is fed the very tensors the HIP decoder
saw (memory read-out, query value, r3, r2), so the comparison isolates the backward kernels: data gradients (forward
implicit-GEMM kernel over flipped filters + ReLU mask / skip gradient in the epilogue), weight gradients (GEMM over the
pixels on transposed operands, split along K), bias gradients, and the adjoint of Refine's interpolate-and-add.
Tolerance: 1e-4 of each tensor's largest magnitude (f32 MFMA sums vs float64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 20200212


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(1e-30, b.double().abs().max().item())


@pytest.mark.parametrize('H,W', [(96, 160), (112, 176)])
def test_decoder_global_branch_backward_vs_autograd(gpu, H, W):
    from tools import synth
    from vfloodnet_amd import AFB_URR, FeatureBank
    from vfloodnet_amd.backward import DecoderBackward
    from oracle import afb_urr_ref as O
    sd = synth.make_state_dict(SEED)
    model = AFB_URR(gpu, update_bank=False).to(gpu).eval()
    model.load_state_dict(sd, strict=True)
    frames, m0 = synth.clip(3, 2, H, W)
    oh = synth.onehot(m0).unsqueeze(0)
    k, v = model.memorize(frames[0:1].to(gpu), oh.to(gpu))
    fb = FeatureBank(2, 250000, gpu)
    fb.init_bank(k, v)
    model.segment(frames[1:2].to(gpu), fb)
    eng = model.engine()
    plan, qs, slot = eng.last_query
    K = 2
    g = torch.Generator().manual_seed(H * W)
    grad_p = torch.randn(K, plan.h4, plan.w4, 2, generator=g).to(gpu)          # dL/dp, NHWC like plan.pp
    grads, gin = DecoderBackward(eng).run(plan, grad_p, qs, slot)
    torch.cuda.synchronize()

    # ---- the oracle on the same inputs, float64, autograd
    nchw = lambda t: t.permute(0, 3, 1, 2).contiguous().cpu().double()
    mem = nchw(plan.dec_in).requires_grad_()                                     # [K,512,h16,w16]
    q_out = qs.kv_q[slot, :, 128:].t().reshape(1, 512, plan.h16, plan.w16).cpu().double().requires_grad_()
    r3 = nchw(qs.q['res3']['out'][slot:slot + 1]).requires_grad_()
    r2 = nchw(qs.q['res2']['out'][slot:slot + 1]).requires_grad_()
    sd64 = {n: t.double().clone().requires_grad_() for n, t in sd.items() if n.startswith('decoder.') and t.is_floating_point()}
    patch_match = torch.cat([mem, q_out.expand(K, -1, -1, -1)], dim=1)           # AFB_URR.py:159 per object
    p = O.decoder_global(sd64, patch_match, r3.expand(K, -1, -1, -1), r2.expand(K, -1, -1, -1))     # AFB_URR.py:289-295
    assert _rel(nchw(plan.pp), p.detach()) < 1e-4                                # same forward
    (p * nchw(grad_p)).sum().backward()

    worst = {}
    for name, got in grads.items():
        assert sd64[name].grad is not None, name
        worst[name] = _rel(got.cpu(), sd64[name].grad)
    for name, ref in (('mem', mem.grad), ('q_out', q_out.grad), ('r3', r3.grad), ('r2', r2.grad)):
        worst['input.' + name] = _rel(nchw(gin[name]), ref)
    bad = {n: e for n, e in worst.items() if not e < 1e-4}
    print('backward slice, worst relative errors:', {n: f'{e:.1e}' for n, e in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
    assert not bad, bad
    # every weight and bias of the global branch is covered
    want = {n for n in sd64 if not n.startswith('decoder.local_')}
    assert set(grads) == want, want ^ set(grads)


def test_transposed_im2col_and_adjoint_kernels(gpu):
    """vfn_transpose_taps_f32 against unfold, vfn_colsum_f32 against sum, vfn_upsample2x_add_backward_f32 against autograd
    of interpolate, on ragged shapes."""
    import torch.nn.functional as F
    from vfloodnet_amd import _lib
    from vfloodnet_amd._lib import ptr, stream, check
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    N, H, W, C, ld = 2, 7, 9, 40, 48
    x = torch.randn(N, H, W, ld, generator=g)
    M = N * H * W
    Mpad = (M + 31) // 32 * 32
    for relu in (0, 1):
        out = torch.full((9 * C, Mpad), float('nan'), device=gpu)
        check(L.vfn_transpose_taps_f32(ptr(x.to(gpu)), N, H, W, C, ld, relu, 9, ptr(out), Mpad, stream()), 'taps')
        xc = x[..., :C].permute(0, 3, 1, 2)
        xc = F.relu(xc) if relu else xc
        cols = F.unfold(xc, 3, padding=1).view(N, C, 9, H * W)                  # [N, C, tap, HW]
        want = cols.permute(2, 1, 0, 3).reshape(9 * C, M)
        assert torch.equal(out[:, :M].cpu(), want) and float(out[:, M:].abs().sum()) == 0.0
    y = torch.randn(1000, 24, generator=g)
    part = torch.empty(64 * 20, device=gpu)
    cs = torch.empty(20, device=gpu)
    check(L.vfn_colsum_f32(ptr(y.to(gpu)), 1000, 20, 24, ptr(part), 64, ptr(cs), stream()), 'colsum')
    assert (cs.cpu() - y[:, :20].sum(0)).abs().max() < 1e-3
    for (n, h, w, c) in [(2, 6, 10, 8), (3, 2, 2, 4), (1, 14, 22, 12)]:
        gm = torch.randn(n, h, w, c, generator=g)
        pm = torch.zeros(n, c, h // 2, w // 2, dtype=torch.float64, requires_grad=True)
        up = F.interpolate(pm, scale_factor=2, mode='bilinear', align_corners=False)
        (up * gm.permute(0, 3, 1, 2).double()).sum().backward()
        gs = torch.empty(1, h, w, c, device=gpu)
        gpm = torch.empty(n, h // 2, w // 2, c, device=gpu)
        check(L.vfn_upsample2x_add_backward_f32(ptr(gm.to(gpu)), ptr(gs), ptr(gpm), n, h, w, c, 1, stream()), 'adjoint')
        assert (gpm.cpu().permute(0, 3, 1, 2).double() - pm.grad).abs().max() < 1e-5
        assert (gs.cpu() - gm.sum(0, keepdim=True)).abs().max() < 1e-5
