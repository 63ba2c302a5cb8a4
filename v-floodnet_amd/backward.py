"""Backward pass, first slice: the decoder's global branch.

``train_video_seg.py:65-74`` calls ``loss.backward()`` through ``AFB_URR.segment``; this module is the first piece of that
on the HIP path: given dL/dp for ``p = pred2(relu(RF2(r2, RF3(r3, ResMM(convFM(patch_match))))))`` (``AFB_URR.py:209-212``,
the decoder before the local refinement), it returns dL/d(weights and biases) of ``convFM, ResMM, RF3.*, RF2.*, pred2`` and
dL/d(``patch_match, r3, r2``) -- the gradients that flow on into the encoders and the memory read.

How (all f32, exact MFMA; csrc/backward_ops.hip has the details):
  * data gradients run the forward implicit-GEMM kernel over flipped, transposed filters; the ReLU in front of the forward
    convolution and the ResBlock's skip connection are undone in its epilogue (``vfn_conv_desc.mask`` / ``res``);
  * weight gradients are GEMMs over the pixels: both operands are transposed (``vfn_transpose_taps_f32``: dY^T and the
    transposed im2col image of the layer's input) and the same kernel runs a 1x1 problem cut along K over the whole chip;
  * bias gradients are column sums; ``Refine``'s interpolate-and-add has its adjoint kernel.
The forward's economies are mirrored: branches the objects share (``RF*.convFS / ResFS``, the query half of ``convFM``) are
differentiated once on the gradient summed over the objects -- what autograd does to the reference's ``expand``.

Activations come from the forward's own buffers (``FramePlan`` / ``QuerySet``): call it right after ``segment``.
Not yet covered: the local refinement head, the softmax-of-softmax tail, the memory read and the encoders."""
import torch

from . import _lib, ops, weights as W
from ._lib import ptr, stream, check
from .engine import choose_cfg, apply_choice, DK, DV


def _dgrad_filters(w):
    """Forward filters [Cout,Cin,3,3] -> packed filters of the data-gradient convolution (Cin 'filters' over Cout channels)."""
    wt = w.detach().float().flip(2, 3).transpose(0, 1).contiguous()          # [Cin, Cout, 3, 3]
    return ops.pad_rows(W.pack_conv_weight(wt))


class DecoderBackward:
    NB = 256          # blocks of the column-sum's first stage

    def __init__(self, engine):
        self.eng = engine
        if engine.mode != 0:
            raise RuntimeError('the backward slice is f32 only')
        dev = engine.device
        d = engine.model.decoder
        self.dev = dev
        self.f = {}                                        # name -> packed data-gradient filters
        for name, conv in (('ResMM.conv1', d.ResMM.conv1), ('ResMM.conv2', d.ResMM.conv2),
                           ('RF3.convFS', d.RF3.convFS), ('RF3.ResFS.conv1', d.RF3.ResFS.conv1), ('RF3.ResFS.conv2', d.RF3.ResFS.conv2),
                           ('RF3.ResMM.conv1', d.RF3.ResMM.conv1), ('RF3.ResMM.conv2', d.RF3.ResMM.conv2),
                           ('RF2.convFS', d.RF2.convFS), ('RF2.ResFS.conv1', d.RF2.ResFS.conv1), ('RF2.ResFS.conv2', d.RF2.ResFS.conv2),
                           ('RF2.ResMM.conv1', d.RF2.ResMM.conv1), ('RF2.ResMM.conv2', d.RF2.ResMM.conv2)):
            self.f[name] = (_dgrad_filters(conv.weight).to(dev), conv.weight.shape[1])
        wfm = d.convFM.weight
        self.f['convFM.m'] = (_dgrad_filters(wfm[:, :DV]).to(dev), DV)
        self.f['convFM.q'] = (_dgrad_filters(wfm[:, DV:]).to(dev), DV)
        wp = torch.zeros(32, d.pred2.weight.shape[1], 3, 3)                 # pred2 has 2 filters: its gradient arrives in a
        wp[:2] = d.pred2.weight.detach().float()                           # 32-channel tensor (channels 2.. are zero)
        self.f['pred2'] = (_dgrad_filters(wp).to(dev), d.pred2.weight.shape[1])
        # the local refinement head (AFB_URR.py:231-234)
        for name, conv in (('local_ResMM.conv1', d.local_ResMM.conv1), ('local_ResMM.conv2', d.local_ResMM.conv2)):
            self.f[name] = (_dgrad_filters(conv.weight).to(dev), conv.weight.shape[1])
        wl = d.local_convFM.weight                                           # input = cat([r1, r1_local]): 64 + 64 channels
        self.f['local_convFM.r1'] = (_dgrad_filters(wl[:, :64]).to(dev), 64)
        self.f['local_convFM.loc'] = (_dgrad_filters(wl[:, 64:]).to(dev), 64)
        wp = torch.zeros(32, d.local_pred2.weight.shape[1], 3, 3)
        wp[:2] = d.local_pred2.weight.detach().float()
        self.f['local_pred2'] = (_dgrad_filters(wp).to(dev), d.local_pred2.weight.shape[1])
        self._scratch = {}

    # ------------------------------------------------------------------ pieces
    def _buf(self, key, numel):
        t = self._scratch.get(key)
        if t is None or t.numel() < numel:
            t = torch.empty(numel, device=self.dev, dtype=torch.float32)
            self._scratch[key] = t
        return t[:numel]

    def _launch(self, d, plan):
        choice = choose_cfg(d.M, d.Cout, d.KH * d.KW * d.Cin, 0)
        if choice[1] > 1 and (d.out_ld % 4 or (d.res and d.res_ld % 4) or (d.mask and d.mask_ld % 4)):
            choice = (choice[0], 1, 0)
        cfg = apply_choice(d, choice, plan.ws, None)
        ops.conv2d_launch(d, cfg, 0)

    def dgrad(self, plan, name, gy, N, H, Wd, mask=None, res=None):
        """dL/dx of y = conv3x3(act(x)) given gy = dL/dy [N,H,W,Cout]: conv(gy, flipped filters), then the mask of act = ReLU
        (``mask`` = x) and the gradient arriving over a skip connection (``res``)."""
        wp, cin = self.f[name]
        out = torch.empty(N, H, Wd, cin, device=self.dev)
        d = ops.make_conv_desc(gy, wp, cin, 3, 3, 1, 1, out, None, None, res, False, False, N=N, H=H, W=Wd)
        if mask is not None:
            d.mask, d.mask_ld = ptr(mask), mask.shape[-1]
        self._launch(d, plan)
        return out

    def wgrad(self, plan, x, gy, relu, x_ld=None, x_c=None, gy_c=None):
        """(dL/dW [Cout,Cin,3,3], dL/db [Cout]) of y = conv3x3(act(x)) + b given gy [N,H,W,Cout] (``gy_c``: the first gy_c
        channels of a wider gradient tensor)."""
        L = _lib.lib()
        N, H, Wd = gy.shape[0], gy.shape[1], gy.shape[2]
        cout = gy_c if gy_c is not None else gy.shape[-1]
        cin = x_c if x_c is not None else x.shape[-1]
        ld_x = x_ld if x_ld is not None else x.shape[-1]
        M = N * H * Wd
        Mpad = (M + 31) // 32 * 32
        rows = (9 * cin + 255) // 256 * 256                                  # filter rows padded to the widest tile
        xt = self._buf('xt', rows * Mpad).view(rows, Mpad)
        if rows > 9 * cin:
            xt[9 * cin:].zero_()
        gt = self._buf('gt', max(cout, 32) * Mpad).view(-1, Mpad)[:cout]
        check(L.vfn_transpose_taps_f32(ptr(x), N, H, Wd, cin, ld_x, int(relu), 9, ptr(xt), Mpad, stream()), 'vfn_transpose_taps_f32')
        check(L.vfn_transpose_taps_f32(ptr(gy), N, H, Wd, cout, gy.shape[-1], 0, 1, ptr(gt), Mpad, stream()), 'vfn_transpose_taps_f32')
        dw = torch.empty(cout, 9 * cin, device=self.dev)
        # the forward kernel on a 1x1 problem: 'pixels' = the Cout rows of dY^T, 'channels' = the padded pixel axis, filters =
        # the rows of the transposed im2col image
        d = ops.make_conv_desc(gt.view(1, 1, cout, Mpad), xt, 9 * cin, 1, 1, 1, 0, dw.view(1, 1, cout, 9 * cin), None, None, None,
                               False, False, N=1, H=1, W=cout)
        tiles = ops.conv_cfg_tiles()
        cfg = 0 if cout >= 128 else 4                                         # 128x128 tiles, or 32x64 for pred2's two rows
        bm, bn = tiles[cfg]
        blocks = ((cout + bm - 1) // bm) * ((9 * cin + bn - 1) // bn)
        splits = [k for k in ops.valid_splits(d, 16) if k * cout * 9 * cin <= plan.ws.numel()]
        want = max(1, 512 // max(1, blocks))
        ks = max([k for k in splits if k <= want] or [1])
        apply_choice(d, (cfg, ks, 0), plan.ws, None)
        ops.conv2d_launch(d, cfg, 0)
        db = torch.empty(cout, device=self.dev)
        part = self._buf('colsum', self.NB * cout)
        check(L.vfn_colsum_f32(ptr(gy), M, cout, gy.shape[-1], ptr(part), self.NB, ptr(db), stream()), 'vfn_colsum_f32')
        return dw.view(cout, 3, 3, cin).permute(0, 3, 1, 2), db              # packed (kh,kw,cin) -> torch's [Cout,Cin,kh,kw]

    def resblock(self, plan, grads, name, x, r, gy, N, H, Wd):
        """ResBlock y = x + conv2(relu(conv1(relu(x)))) (AFB_URR.py:23-30): returns dL/dx; weight gradients into ``grads``."""
        g_r = self.dgrad(plan, name + '.conv2', gy, N, H, Wd, mask=r)
        grads[f'decoder.{name}.conv2.weight'], grads[f'decoder.{name}.conv2.bias'] = self.wgrad(plan, r, gy, True)
        g_x = self.dgrad(plan, name + '.conv1', g_r, N, H, Wd, mask=x, res=gy)
        grads[f'decoder.{name}.conv1.weight'], grads[f'decoder.{name}.conv1.bias'] = self.wgrad(plan, x, g_r, True)
        return g_x

    def refine(self, plan, grads, name, f, s, dm, g_out, N, H, Wd):
        """Refine.forward (AFB_URR.py:122-127) backwards.  f: skip feature [1,H,W,Cf]; s = [convFS(f), ResFS.conv1 out, ResFS out]
        (one image, shared by the N objects); dm = [m, ResMM.conv1 out] for the N objects.  Returns (dL/df, dL/dpm)."""
        L = _lib.lib()
        g_m = self.resblock(plan, grads, name + '.ResMM', dm[0], dm[1], g_out, N, H, Wd)
        C = g_m.shape[-1]
        g_s = torch.empty(1, H, Wd, C, device=self.dev)
        g_pm = torch.empty(N, H // 2, Wd // 2, C, device=self.dev)
        check(L.vfn_upsample2x_add_backward_f32(ptr(g_m), ptr(g_s), ptr(g_pm), N, H, Wd, C, 1, stream()), 'vfn_upsample2x_add_backward_f32')
        g_s0 = self.resblock(plan, grads, name + '.ResFS', s[0], s[1], g_s, 1, H, Wd)
        grads[f'decoder.{name}.convFS.weight'], grads[f'decoder.{name}.convFS.bias'] = self.wgrad(plan, f, g_s0, False)
        g_f = self.dgrad(plan, name + '.convFS', g_s0, 1, H, Wd)
        return g_f, g_pm

    # ------------------------------------------------------------------ the slice
    @torch.no_grad()
    def run(self, plan, grad_p, qs=None, slot=0):
        """grad_p: dL/dp, f32 [obj_n, h/4, w/4, 2] (NHWC, the layout of ``plan.pp``) for the frame ``segment`` ran last.
        Returns (dict state-dict name -> gradient, dict input name -> gradient)."""
        p = plan
        qs = qs or p.qsets[0]
        K = p.obj_n
        o = lambda t: t[slot:slot + 1]
        grads = {}
        g32 = torch.zeros(K, p.h4, p.w4, 32, device=self.dev)
        g32[..., :2].copy_(grad_p)
        # pred2(relu(x)), x = RF2's output (AFB_URR.py:212)
        x = p.d4[2]
        g = self.dgrad(p, 'pred2', g32, K, p.h4, p.w4, mask=x)
        grads['decoder.pred2.weight'], grads['decoder.pred2.bias'] = self.wgrad(p, x, grad_p.contiguous(), True)
        # RF2, RF3 (AFB_URR.py:210-211)
        g_r2, g = self.refine(p, grads, 'RF2', o(qs.q['res2']['out']), [o(t) for t in qs.s4], p.d4, g, K, p.h4, p.w4)
        g_r3, g = self.refine(p, grads, 'RF3', o(qs.q['res3']['out']), [o(t) for t in qs.s8], p.d8, g, K, p.h8, p.w8)
        # ResMM(convFM(patch_match)) (AFB_URR.py:209); patch_match = cat([mem_i, q_out]) per object (:159)
        g = self.resblock(p, grads, 'ResMM', p.d16[0], p.d16[1], g, K, p.h16, p.w16)
        g_mem = self.dgrad(p, 'convFM.m', g, K, p.h16, p.w16)
        dw_m, db = self.wgrad(p, p.dec_in, g, False)
        g_q = self._sum_objects(g)                                           # the query half is shared: sum over the objects
        kvq_val = o(qs.kv_q)[:, :, DK:]                                        # [1, HW, 512] view, pixel stride 640
        dw_q, _ = self.wgrad(p, kvq_val, g_q, False, x_ld=DK + DV, x_c=DV)
        g_qv = self.dgrad(p, 'convFM.q', g_q, 1, p.h16, p.w16)
        grads['decoder.convFM.weight'] = torch.cat([dw_m, dw_q], dim=1)
        grads['decoder.convFM.bias'] = db
        # patch_match[i] = cat([mem_i, q_out]) (AFB_URR.py:159): dL/dmem per object, dL/dq_out summed over the objects
        return grads, {'mem': g_mem, 'q_out': g_qv, 'r3': g_r3, 'r2': g_r2}

    @torch.no_grad()
    def run_tail(self, plan, grad_score, qs=None, slot=0):
        """The whole decoder backwards: ``grad_score`` = dL/d(logits ``segment`` returned) f32 [obj_n, H0, W0] for the frame
        ``segment`` ran last (AFB_URR.py:208-239 + :300,309-316).  Runs the tail and the local refinement head, then
        ``run`` for the global branch.  Returns (dict state-dict name -> gradient for every ``decoder.*`` parameter, dict of
        input gradients: mem, q_out, r3, r2, r1)."""
        L = _lib.lib()
        p = plan
        qs = qs or p.qsets[0]
        K, h2, w2 = p.obj_n, p.h2, p.w2
        npix = h2 * w2
        dev = self.dev
        s = stream()
        r1 = qs.q['r1'][slot:slot + 1]                                         # [1,h2,w2,64]
        G = grad_score.contiguous()
        g_o = torch.zeros(K, 2 * h2, 2 * w2, 4, device=dev)
        check(L.vfn_tail_grad_o_f32(ptr(G), ptr(p.p_up), ptr(p.unc), ptr(p.conf), ptr(p.qq), ptr(g_o), K, h2, w2,
                                    p.pad[2], p.pad[0], p.H0, p.W0, s), 'vfn_tail_grad_o_f32')
        g_p2 = torch.empty(K, h2, w2, 4, device=dev)
        check(L.vfn_upsample2x_add_backward_f32(ptr(g_o), None, ptr(g_p2), K, 2 * h2, 2 * w2, 4, 0, s), 'vfn_upsample2x_add_backward_f32')
        g_q = torch.zeros(K, h2, w2, 32, device=dev)
        g_cf = torch.empty(K, h2, w2, device=dev)
        g_u = torch.empty(h2, w2, device=dev)
        check(L.vfn_tail_split_f32(ptr(g_p2), ptr(p.unc), ptr(p.conf), ptr(p.qq), ptr(g_q), ptr(g_cf), ptr(g_u), K, npix, s),
              'vfn_tail_split_f32')
        grads = {}
        # q = conf * local_pred2(relu(local_ResMM(local_convFM(cat([r1, r1_local])))))   (AFB_URR.py:231-234)
        l2 = p.l2
        g = self.dgrad(p, 'local_pred2', g_q, K, h2, w2, mask=l2[2])
        grads['decoder.local_pred2.weight'], grads['decoder.local_pred2.bias'] = self.wgrad(p, l2[2], g_q, True, gy_c=2)
        g = self.resblock(p, grads, 'local_ResMM', l2[0], l2[1], g, K, h2, w2)
        g_lm = self.dgrad(p, 'local_convFM.loc', g, K, h2, w2)
        dw_loc, _ = self.wgrad(p, p.lm, g, False)
        g_lq = self._sum_objects(g)                                            # the r1 half is shared by the objects (:231)
        dw_r1, db = self.wgrad(p, r1, g_lq, False)
        g_r1 = self.dgrad(p, 'local_convFM.r1', g_lq, 1, h2, w2)
        grads['decoder.local_convFM.weight'] = torch.cat([dw_r1, dw_loc], dim=1)
        grads['decoder.local_convFM.bias'] = db
        # r1_local, conf, uncertainty, the two softmaxes -> interpolate(p)
        dA = torch.empty(K, h2, w2, 64, device=dev)
        dBv = torch.empty(K, h2, w2, device=dev)
        amax = torch.empty(K, h2, w2, dtype=torch.int32, device=dev)
        g_pup = torch.empty(K, h2, w2, 4, device=dev)
        check(L.vfn_local_stats_backward_f32(ptr(g_lm), ptr(p.lm), ptr(g_cf), ptr(g_u), ptr(g_p2), ptr(r1), ptr(p.rough), ptr(p.p_up),
                                             ptr(dA), ptr(dBv), ptr(amax), ptr(g_r1), ptr(g_pup), K, h2, w2, 64, s),
              'vfn_local_stats_backward_f32')
        g_p4 = torch.empty(K, p.h4, p.w4, 4, device=dev)
        check(L.vfn_upsample2x_add_backward_f32(ptr(g_pup), None, ptr(g_p4), K, h2, w2, 4, 0, s), 'vfn_upsample2x_add_backward_f32')
        g2, inputs = self.run(p, g_p4[..., :2].contiguous(), qs, slot)
        grads.update(g2)
        inputs['r1'] = g_r1
        return grads, inputs

    def _sum_objects(self, g):
        """[N,h,w,C] -> [1,h,w,C], summed over the objects in index order."""
        out = g[0:1].clone()
        for n in range(1, g.shape[0]):
            out += g[n:n + 1]
        return out
