"""Backward pass of the training step: ``loss.backward()`` of ``train_video_seg.py:65-74`` on the HIP path.

``DecoderBackward``: from dL/dscores through the decoder (logit / clamp tail, both interpolations, the two softmaxes, the top-2
uncertainty, the 7x7 local windows, local head, global branch: ``AFB_URR.py:185-240``) to every ``decoder.*`` parameter and to the
decoder's inputs.  ``ModelBackward``: the memory read (``Matcher.forward``, bank = one frame), ``KeyValue``, both ResNet trunks
with BatchNorm frozen (``train_video_seg.py:103-106``) and the 7x7 stems -- the gradient of EVERY parameter; ``train.py`` drives
it (one sample after the other, the gradients that reach the bank go through ``memorize`` once).

How (all f32, exact MFMA; csrc/backward_ops.hip has the details):
  * data gradients run the forward implicit-GEMM kernel over flipped, transposed filters; the ReLU in front of the forward
    convolution and the skip connection are undone in its epilogue (``vfn_conv_desc.mask`` / ``res`` / ``mask_after``); a stride-2
    convolution differentiates as the stride-1 data gradient of the zero-inserted gradient (``vfn_dilate2_f32``);
  * weight gradients are GEMMs over the pixels: both operands are transposed (``vfn_transpose_taps_f32``: dY^T and the
    transposed im2col image of the layer's input) and the same kernel runs a 1x1 problem cut along K over the whole chip;
  * bias and frozen-BatchNorm (gamma, beta) gradients are two-stage column sums; ``Refine``'s interpolate-and-add, the max-pool
    and the window statistics have adjoint kernels.
The forward's economies are mirrored: branches the objects share (``RF*.convFS / ResFS``, the query half of ``convFM``) are
differentiated once on the gradient summed over the objects -- what autograd does to the reference's ``expand``.

Activations come from the forward's own buffers (the training ``FramePlan`` / ``QuerySet`` keep them): call it right after
``segment``.  Checked against float64 autograd through the oracle (tests/test_backward_gpu.py)."""
import torch

from . import _lib, ops, weights as W
from ._lib import ptr, stream, check
from .refresh import DGRAD
from .engine import choose_cfg, apply_choice, DK, DV


_IMPLICIT_WGRAD = __import__('os').environ.get('VFN_IMPLICIT_WGRAD', '1') == '1'      # 0: the round-3 path (transposed operands)
_SIDE_WGRAD = __import__('os').environ.get('VFN_SIDE_WGRAD', '1') == '1'              # 0: weight gradients on the main stream
_SIDE_PRIORITY = int(__import__('os').environ.get('VFN_SIDE_PRIORITY', 0))               # -1: the side stream's kernels are dispatched first
_WINOGRAD_WGRAD = __import__('os').environ.get('VFN_WINOGRAD_WGRAD', '1') == '1'      # weight gradients of the big 3x3 layers in the Winograd domain
_WINOGRAD_WGRAD_MIN_WORK = int(__import__('os').environ.get('VFN_WINOGRAD_WGRAD_MIN_WORK', 600000))        # pixels x min(cin, cout)
_SIDE_DROP = __import__('os').environ.get('VFN_SIDE_DROP', '0') == '1'
_NO_WAIT_PLAN = __import__('os').environ.get('VFN_UNSAFE_NO_WAIT_PLAN', '0') == '1'   # measurement only (WRONG gradients): what the per-sample wait costs
_SIDE2 = __import__('os').environ.get('VFN_SIDE2', '1') == '1'                       # a second side stream for the memory encoder's weight gradients
_SIDE3 = __import__('os').environ.get('VFN_SIDE3', '0') == '1'                       # ... and a third for the query encoder's (the first keeps the decoder's): measured
#                                                                                      no gain (22.0 / 22.1 against 21.9 / 21.8 ms per step, profiles/r05_train_side_streams.txt): off
_SIDE_GROUP = int(__import__('os').environ.get('VFN_SIDE_GROUP', 8))                   # deferred launches per side-stream hand-over


def _dgrad_filters(w):
    """Forward filters [Cout,Cin,3,3] -> packed filters of the data-gradient convolution (Cin 'filters' over Cout channels)."""
    wt = w.detach().float().flip(2, 3).transpose(0, 1).contiguous()          # [Cin, Cout, 3, 3]
    return ops.pad_rows(W.pack_conv_weight(wt))


class DecoderBackward:
    NB = 256          # blocks of the column-sum's first stage
    NB1 = max(1, min(NB, int(__import__('os').environ.get('VFN_COLSUM_BLOCKS', 128))))     # ... of the one-launch form (vfn_colsum_acc_f32 with a counter); <= NB: the scratch rows are sized by NB

    def __init__(self, engine):
        self.eng = engine
        if engine.mode != 0:
            raise RuntimeError('the backward slice is f32 only')
        dev = engine.device
        d = engine.model.decoder
        self.dev = dev
        self.f = {}                                        # name -> packed data-gradient filters
        self.fw, self._fsrc = {}, {}                       # name -> their Winograd banks (built on first use) / the parameter behind them
        for name, conv in (('ResMM.conv1', d.ResMM.conv1), ('ResMM.conv2', d.ResMM.conv2),
                           ('RF3.convFS', d.RF3.convFS), ('RF3.ResFS.conv1', d.RF3.ResFS.conv1), ('RF3.ResFS.conv2', d.RF3.ResFS.conv2),
                           ('RF3.ResMM.conv1', d.RF3.ResMM.conv1), ('RF3.ResMM.conv2', d.RF3.ResMM.conv2),
                           ('RF2.convFS', d.RF2.convFS), ('RF2.ResFS.conv1', d.RF2.ResFS.conv1), ('RF2.ResFS.conv2', d.RF2.ResFS.conv2),
                           ('RF2.ResMM.conv1', d.RF2.ResMM.conv1), ('RF2.ResMM.conv2', d.RF2.ResMM.conv2)):
            self._filters(name, conv.weight)
        wfm = d.convFM.weight
        self._filters('convFM.m', wfm, 0, DV)
        self._filters('convFM.q', wfm, DV, DV)
        self._filters('pred2', d.pred2.weight, cout_ld=32)                  # pred2 has 2 filters: its gradient arrives in a
        # the local refinement head (AFB_URR.py:231-234)                     # 32-channel tensor (channels 2.. are zero)
        for name, conv in (('local_ResMM.conv1', d.local_ResMM.conv1), ('local_ResMM.conv2', d.local_ResMM.conv2)):
            self._filters(name, conv.weight)
        wl = d.local_convFM.weight                                           # input = cat([r1, r1_local]): 64 + 64 channels
        self._filters('local_convFM.r1', wl, 0, 64)
        self._filters('local_convFM.loc', wl, 64, 64)
        self._filters('local_pred2', d.local_pred2.weight, cout_ld=32)
        self._scratch = {}
        self._ticket = torch.zeros(64, dtype=torch.int32, device=dev)      # arrival counters (VFN_COLSUM_COUNTERS) of the one-launch column sums, zero at rest
        self._ticket_main = torch.zeros(64, dtype=torch.int32, device=dev) # ... of the column sums that stay on the main stream (unnamed / fallback paths)
        self.sink = None                                   # ModelBackward: weight gradients accumulate there, in the kernel

    def _filters(self, name, weight, cin_off=0, cin=None, cout_ld=0):
        """Packed data-gradient filters of (a slice of the input channels of) a 3 x 3 convolution, ``cout_ld`` > Cout: zero
        filters up to that width; registered with the engine's Refresher so that they follow the parameter."""
        cout = weight.shape[0]
        cin = weight.shape[1] if cin is None else cin
        w = weight.detach().float()[:, cin_off:cin_off + cin]
        if cout_ld > cout:
            wp = torch.zeros(cout_ld, cin, 3, 3, device=w.device)
            wp[:cout] = w
            w = wp
        self.f[name] = (_dgrad_filters(w).to(self.dev), cin)
        self._fsrc[name] = (weight, cin_off, max(cout, cout_ld))
        self.eng.refresher.add_filter(weight, self.f[name][0], DGRAD, cin=cin, cin_off=cin_off, dst_ld=9 * max(cout, cout_ld),
                                      cout_ld=max(cout, cout_ld))

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
        cfg = apply_choice(d, choice, plan.ws, plan.cnt)          # (split tiles are finished inside the launch)
        ops.conv2d_launch(d, cfg, 0)

    def dgrad(self, plan, name, gy, N, H, Wd, mask=None, res=None):
        """dL/dx of y = conv3x3(act(x)) given gy = dL/dy [N,H,W,Cout]: conv(gy, flipped filters), then the mask of act = ReLU
        (``mask`` = x) and the gradient arriving over a skip connection (``res``)."""
        wp, cin = self.f[name]
        out = torch.empty(N, H, Wd, cin, device=self.dev)
        if self._fsrc[name][2] == self._fsrc[name][0].shape[0] and self._use_winograd(N * H * Wd, gy.shape[-1], cin):      # (not the zero-padded two-filter heads)
            return self._dgrad_winograd(plan, name, gy, out, N, H, Wd, cin, mask, res)
        d = ops.make_conv_desc(gy, wp, cin, 3, 3, 1, 1, out, None, None, res, False, False, N=N, H=H, W=Wd)
        if mask is not None:
            d.mask, d.mask_ld = ptr(mask), mask.shape[-1]
        self._launch(d, plan)
        return out

    def _use_winograd(self, M, c_in, c_out):
        """Winograd F(4x4, 3x3) for a data-gradient convolution with c_in gradient channels and c_out outputs (engine.Engine.
        use_winograd's rule: the measured table, else >= 128 channels either side and enough pixels)."""
        from . import engine as E
        if E._WINOGRAD == '0' or not E._WINOGRAD_TRAIN or c_in % 32 or c_out % 4 or c_out < 32:
            return False
        if E._WINOGRAD == '2':
            return True
        hit = E._WINO_TABLE.get((M, c_in, c_out))
        if hit is not None:
            return bool(hit)
        return c_in >= 128 and c_out >= 128 and M >= E._WINOGRAD_MIN_M

    def _dgrad_winograd(self, plan, name, gy, out, N, H, Wd, cin, mask, res):
        """The data-gradient convolution in the transform domain (csrc/conv_winograd.hip): input transform of gy, the 36 GEMMs
        over the banks of the flipped, transposed filters, output transform with the mask / skip-connection epilogue."""
        L = _lib.lib()
        U = self.fw.get(name)
        if U is None:
            weight, cin_off, cout_ld = self._fsrc[name]
            w = weight.detach().float()[:, cin_off:cin_off + cin].flip(2, 3).transpose(0, 1)      # [cin, Cout, 3, 3]: the dgrad's filters
            U = ops.pack_winograd_weight(w).to(self.dev)
            self.fw[name] = U
            from .refresh import WINO_DGRAD
            self.eng.refresher.add_filter(weight, U, WINO_DGRAD, cin=cin, cin_off=cin_off, dst_ld=weight.shape[0], cout_ld=U.shape[0] // 36)
            self.eng._settle()
        c_g = gy.shape[-1]
        rows = ops.winograd_rows(N, H, Wd)
        need_v, need_m = 36 * rows * c_g, 36 * rows * cin
        V, Mb = self._scratch.get('wino_v'), self._scratch.get('wino_m')
        if V is None or V.numel() < need_v:
            V = self._scratch['wino_v'] = torch.zeros(need_v, device=self.dev)
        if Mb is None or Mb.numel() < need_m:
            Mb = self._scratch['wino_m'] = torch.empty(need_m, device=self.dev)
        V, Mb = V[:need_v].view(36 * rows, c_g), Mb[:need_m].view(36 * rows, cin)
        ops.winograd_input(gy, V, rows, False, N, H, Wd, c_g, c_g)
        dg = ops.make_winograd_gemm_desc(V, U, Mb, rows, c_g, cin)
        self._launch(dg, plan)
        check(L.vfn_winograd_output_masked_f32(ptr(Mb), rows, N, H, Wd, cin, ptr(res), res.shape[-1] if res is not None else 0, ptr(mask),
                                               mask.shape[-1] if mask is not None else 0, 0, ptr(out), cin, stream()),
              'vfn_winograd_output_masked_f32')
        return out

    def wgrad(self, plan, x, gy, relu, x_ld=None, x_c=None, gy_c=None, name=None, bias=None):
        """(dL/dW [Cout,Cin,3,3], dL/db [Cout]) of y = conv3x3(act(x)) + b given gy [N,H,W,Cout] (``gy_c``: the first gy_c
        channels of a wider gradient tensor).  ``name`` (with a ``sink``): the weight gradient is accumulated into the sink's
        buffer of that parameter by the kernel and None is returned in its place.  ``name`` may be one input-channel half of a
        parameter ('decoder.convFM.weight#m': ModelBackward.SPLIT puts the halves together); ``bias``: the bias parameter's name
        (default: ``name`` with 'weight' replaced), False = this call contributes no bias gradient."""
        L = _lib.lib()
        N, H, Wd = gy.shape[0], gy.shape[1], gy.shape[2]
        cout = gy_c if gy_c is not None else gy.shape[-1]
        cin = x_c if x_c is not None else x.shape[-1]
        ld_x = x_ld if x_ld is not None else x.shape[-1]
        M = N * H * Wd
        if (name is not None and self.sink is not None and cout < 32 and gy.shape[-1] == 32 and
                self.sink.wgrad_into(name, x, gy, 3, 1, 1, cin, 32, ld_x, relu, None, N, H, Wd, rows=cout)):
            # the two-filter heads (pred2 / local_pred2): their gradient tensor is 32 channels wide with zeros beyond the second, so
            # the implicit weight-gradient kernel takes it as it stands (30 zero rows computed, two kept) -- no transposed im2col
            # image (184 MB per call at 1/4 resolution), no GEMM over it, and off the main stream
            db, have = self.sink._small(name[:-len('weight')] + 'bias', cout)
            part = self._buf('colsum', self.NB * cout)
            ticket, nb1 = self._ticket, self.NB1
            self.sink._side_do(lambda: check(L.vfn_colsum_acc_f32(ptr(gy), M, cout, 32, ptr(part), nb1, ptr(db), have, ptr(ticket), stream()),
                                             'vfn_colsum_acc_f32'))
            return None, None
        if name is not None and self.sink is not None and self.sink.wgrad_into(name, x, gy, 3, 1, 1, cin, cout, ld_x, relu, None, N, H, Wd):
            if bias is False:
                return None, None
            # the bias gradient accumulates in the kernel too, beside the data-gradient chain (ModelBackward's side stream)
            db, have = self.sink._small(bias or name[:-len('weight')] + 'bias', cout)
            part = self._buf('colsum', self.NB * cout)
            ld_g, ticket, nb1 = gy.shape[-1], self._ticket, self.NB1
            self.sink._side_do(lambda: check(L.vfn_colsum_acc_f32(ptr(gy), M, cout, ld_g, ptr(part), nb1, ptr(db), have, ptr(ticket), stream()),
                                             'vfn_colsum_acc_f32'))
            return None, None
        if _IMPLICIT_WGRAD and cin % 32 == 0 and cout >= 32:
            # round 4: the reduction over the pixels straight from the NHWC tensors (vfn_conv_wgrad_f32), nothing transposed
            dw = ops.conv_wgrad(x, gy, 3, 1, 1, cin=cin, cout=cout, ld_x=ld_x, relu=relu, N=N, H=H, W=Wd)
            db = torch.empty(cout, device=self.dev)
            part = self._buf('colsum_main', self.NB * cout)        # (main stream: not the side stream's scratch / counters)
            check(L.vfn_colsum_acc_f32(ptr(gy), M, cout, gy.shape[-1], ptr(part), self.NB1, ptr(db), 0, ptr(self._ticket_main), stream()), 'vfn_colsum_acc_f32')
            return dw.view(cout, 3, 3, cin).permute(0, 3, 1, 2), db
        Mpad = (M + 31) // 32 * 32
        rows = (9 * cin + 255) // 256 * 256                                  # filter rows padded to the widest tile
        xt = self._buf('xt', rows * Mpad).view(rows, Mpad)
        if rows > 9 * cin:
            xt[9 * cin:].zero_()
        gt = self._buf('gt', max(cout, 32) * Mpad).view(-1, Mpad)[:cout]
        check(L.vfn_transpose_taps_f32(ptr(x), N, H, Wd, cin, ld_x, int(relu), 3, 1, 1, H, Wd, None, ptr(xt), Mpad, stream()), 'vfn_transpose_taps_f32')
        check(L.vfn_transpose_taps_f32(ptr(gy), N, H, Wd, cout, gy.shape[-1], 0, 1, 1, 0, H, Wd, None, ptr(gt), Mpad, stream()), 'vfn_transpose_taps_f32')
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
        part = self._buf('colsum_main', self.NB * cout)        # (main stream: not the side stream's scratch / counters)
        check(L.vfn_colsum_acc_f32(ptr(gy), M, cout, gy.shape[-1], ptr(part), self.NB1, ptr(db), 0, ptr(self._ticket_main), stream()), 'vfn_colsum_acc_f32')
        return dw.view(cout, 3, 3, cin).permute(0, 3, 1, 2), db              # packed (kh,kw,cin) -> torch's [Cout,Cin,kh,kw]

    def resblock(self, plan, grads, name, x, r, gy, N, H, Wd):
        """ResBlock y = x + conv2(relu(conv1(relu(x)))) (AFB_URR.py:23-30): returns dL/dx; weight gradients into ``grads``."""
        g_r = self.dgrad(plan, name + '.conv2', gy, N, H, Wd, mask=r)
        grads[f'decoder.{name}.conv2.weight'], grads[f'decoder.{name}.conv2.bias'] = self.wgrad(plan, r, gy, True, name=f'decoder.{name}.conv2.weight')
        g_x = self.dgrad(plan, name + '.conv1', g_r, N, H, Wd, mask=x, res=gy)
        grads[f'decoder.{name}.conv1.weight'], grads[f'decoder.{name}.conv1.bias'] = self.wgrad(plan, x, g_r, True, name=f'decoder.{name}.conv1.weight')
        return g_x

    def refine(self, plan, grads, name, f, s, dm, g_out, N, H, Wd, G=1):
        """Refine.forward (AFB_URR.py:122-127) backwards.  f: skip feature [G,H,W,Cf]; s = [convFS(f), ResFS.conv1 out, ResFS out]
        (one image per frame, shared by that frame's objects); dm = [m, ResMM.conv1 out] for the N = G * objects images
        (frame-major).  Returns (dL/df, dL/dpm)."""
        L = _lib.lib()
        K = N // G
        g_m = self.resblock(plan, grads, name + '.ResMM', dm[0], dm[1], g_out, N, H, Wd)
        C = g_m.shape[-1]
        g_s = torch.empty(G, H, Wd, C, device=self.dev)
        g_pm = torch.empty(N, H // 2, Wd // 2, C, device=self.dev)
        for g in range(G):                                                     # (the skip's gradient sums over ONE frame's objects)
            check(L.vfn_upsample2x_add_backward_f32(ptr(g_m[g * K:(g + 1) * K]), ptr(g_s[g:g + 1]), ptr(g_pm[g * K:(g + 1) * K]), K, H, Wd, C, 1,
                                                    stream()), 'vfn_upsample2x_add_backward_f32')
        g_s0 = self.resblock(plan, grads, name + '.ResFS', s[0], s[1], g_s, G, H, Wd)
        grads[f'decoder.{name}.convFS.weight'], grads[f'decoder.{name}.convFS.bias'] = self.wgrad(plan, f, g_s0, False, name=f'decoder.{name}.convFS.weight')
        g_f = self.dgrad(plan, name + '.convFS', g_s0, G, H, Wd)
        return g_f, g_pm

    # ------------------------------------------------------------------ the slice
    @torch.no_grad()
    def run(self, plan, grad_p, qs=None, slot=0, G=1):
        """grad_p: dL/dp, f32 [G * obj_n, h/4, w/4, 2] (NHWC, the layout of ``plan.pp``) for the frame ``segment`` ran last (G = 1:
        ``plan`` is the FramePlan, the frame sits in ``slot`` of the query set) or for the G frames of ``Engine.segment_batch``
        (``plan`` is its DecoderBatch; images frame-major).  Returns (dict state-dict name -> gradient, dict input name -> gradient)."""
        p = plan
        qs = qs or p.qsets[0]
        K = p.obj_n
        N = G * K
        o = (lambda t: t[slot:slot + 1]) if G == 1 else (lambda t: t[0:G])     # the frame-only tensors of the query set
        grads = {}
        g32 = torch.zeros(N, p.h4, p.w4, 32, device=self.dev)
        g32[..., :2].copy_(grad_p)
        # pred2(relu(x)), x = RF2's output (AFB_URR.py:212)
        x = p.d4[2]
        g = self.dgrad(p, 'pred2', g32, N, p.h4, p.w4, mask=x)
        grads['decoder.pred2.weight'], grads['decoder.pred2.bias'] = self.wgrad(p, x, g32 if self.sink is not None else grad_p.contiguous(), True,
                                                                                gy_c=2, name='decoder.pred2.weight')
        # RF2, RF3 (AFB_URR.py:210-211)
        g_r2, g = self.refine(p, grads, 'RF2', o(qs.q['res2']['out']), [o(t) for t in qs.s4], p.d4, g, N, p.h4, p.w4, G)
        g_r3, g = self.refine(p, grads, 'RF3', o(qs.q['res3']['out']), [o(t) for t in qs.s8], p.d8, g, N, p.h8, p.w8, G)
        # ResMM(convFM(patch_match)) (AFB_URR.py:209); patch_match = cat([mem_i, q_out]) per object (:159)
        g = self.resblock(p, grads, 'ResMM', p.d16[0], p.d16[1], g, N, p.h16, p.w16)
        g_mem = self.dgrad(p, 'convFM.m', g, N, p.h16, p.w16)
        # (round 5: the two input halves of convFM / local_convFM accumulate like every other weight gradient -- in the kernel, on
        # the side stream, under half-parameter names that ModelBackward.grads concatenates once per step; they were the last
        # weight-gradient launches on the data-gradient chain, 0.28 ms per sample)
        split = self.sink is not None
        dw_m, db = self.wgrad(p, p.dec_in, g, False, name='decoder.convFM.weight#m' if split else None, bias='decoder.convFM.bias')
        g_q = self._sum_objects(g, G)                                        # the query half is shared: sum over the frame's objects
        kvq_val = o(qs.kv_q)[:, :, DK:]                                        # [G, HW, 512] view, pixel stride 640
        dw_q, _ = self.wgrad(p, kvq_val, g_q, False, x_ld=DK + DV, x_c=DV, name='decoder.convFM.weight#q' if split else None, bias=False)
        g_qv = self.dgrad(p, 'convFM.q', g_q, G, p.h16, p.w16)
        if dw_m is not None or dw_q is not None:
            assert dw_m is not None and dw_q is not None
            grads['decoder.convFM.weight'] = torch.cat([dw_m, dw_q], dim=1)
            grads['decoder.convFM.bias'] = db
        # patch_match[i] = cat([mem_i, q_out]) (AFB_URR.py:159): dL/dmem per object, dL/dq_out summed over the objects
        return grads, {'mem': g_mem, 'q_out': g_qv, 'r3': g_r3, 'r2': g_r2}

    @torch.no_grad()
    def run_tail(self, plan, grad_score, qs=None, slot=0, G=1):
        """The whole decoder backwards: ``grad_score`` = dL/d(logits ``segment`` returned) f32 [obj_n, H0, W0] for the frame
        ``segment`` ran last (AFB_URR.py:208-239 + :300,309-316) -- or, G > 1, [G, obj_n, H0, W0] for the frames of
        ``Engine.segment_batch`` with ``plan`` = its DecoderBatch: the convolutions' data and weight gradients run once over the
        G * obj_n images, the kernels that couple the objects of a frame once per frame.  Runs the tail and the local refinement
        head, then ``run`` for the global branch.  Returns (dict state-dict name -> gradient for every ``decoder.*`` parameter, dict
        of input gradients: mem, q_out, r3, r2, r1 -- the shared ones with a leading axis of G frames)."""
        L = _lib.lib()
        p = plan
        qs = qs or p.qsets[0]
        K, h2, w2 = p.obj_n, p.h2, p.w2
        N = G * K
        npix = h2 * w2
        dev = self.dev
        s = stream()
        grp = lambda t, g: t[g * K:(g + 1) * K]
        r1 = qs.q['r1'][slot:slot + 1] if G == 1 else qs.q['r1'][0:G]           # [G,h2,w2,64]
        unc = p.unc.view(G, h2, w2)
        Gs = grad_score.contiguous().view(G, K, p.H0, p.W0)
        g_o = torch.zeros(N, 2 * h2, 2 * w2, 4, device=dev)
        for g in range(G):
            check(L.vfn_tail_grad_o_f32(ptr(Gs[g]), ptr(grp(p.p_up, g)), ptr(unc[g]), ptr(grp(p.conf, g)), ptr(grp(p.qq, g)), ptr(grp(g_o, g)),
                                        K, h2, w2, p.pad[2], p.pad[0], p.H0, p.W0, s), 'vfn_tail_grad_o_f32')
        g_p2 = torch.empty(N, h2, w2, 4, device=dev)
        check(L.vfn_upsample2x_add_backward_f32(ptr(g_o), None, ptr(g_p2), N, 2 * h2, 2 * w2, 4, 0, s), 'vfn_upsample2x_add_backward_f32')
        g_q = torch.zeros(N, h2, w2, 32, device=dev)
        g_cf = torch.empty(N, h2, w2, device=dev)
        g_u = torch.empty(G, h2, w2, device=dev)
        for g in range(G):
            check(L.vfn_tail_split_f32(ptr(grp(g_p2, g)), ptr(unc[g]), ptr(grp(p.conf, g)), ptr(grp(p.qq, g)), ptr(grp(g_q, g)), ptr(grp(g_cf, g)),
                                       ptr(g_u[g]), K, npix, s), 'vfn_tail_split_f32')
        grads = {}
        # q = conf * local_pred2(relu(local_ResMM(local_convFM(cat([r1, r1_local])))))   (AFB_URR.py:231-234)
        l2 = p.l2
        g = self.dgrad(p, 'local_pred2', g_q, N, h2, w2, mask=l2[2])
        grads['decoder.local_pred2.weight'], grads['decoder.local_pred2.bias'] = self.wgrad(p, l2[2], g_q, True, gy_c=2, name='decoder.local_pred2.weight')
        g = self.resblock(p, grads, 'local_ResMM', l2[0], l2[1], g, N, h2, w2)
        g_lm = self.dgrad(p, 'local_convFM.loc', g, N, h2, w2)
        split = self.sink is not None
        dw_loc, _ = self.wgrad(p, p.lm, g, False, name='decoder.local_convFM.weight#loc' if split else None, bias=False)
        g_lq = self._sum_objects(g, G)                                         # the r1 half is shared by the frame's objects (:231)
        dw_r1, db = self.wgrad(p, r1, g_lq, False, name='decoder.local_convFM.weight#r1' if split else None, bias='decoder.local_convFM.bias')
        g_r1 = self.dgrad(p, 'local_convFM.r1', g_lq, G, h2, w2)
        if dw_loc is not None or dw_r1 is not None:
            assert dw_loc is not None and dw_r1 is not None
            grads['decoder.local_convFM.weight'] = torch.cat([dw_r1, dw_loc], dim=1)
            grads['decoder.local_convFM.bias'] = db
        # r1_local, conf, uncertainty, the two softmaxes -> interpolate(p)
        dA = torch.empty(K, h2, w2, 64, device=dev)                           # (scratch of one frame's launch)
        dBv = torch.empty(K, h2, w2, device=dev)
        amax = torch.empty(K, h2, w2, dtype=torch.int32, device=dev)
        g_pup = torch.empty(N, h2, w2, 4, device=dev)
        for g in range(G):
            check(L.vfn_local_stats_backward_f32(ptr(grp(g_lm, g)), ptr(grp(p.lm, g)), ptr(grp(g_cf, g)), ptr(g_u[g]), ptr(grp(g_p2, g)), ptr(r1[g]),
                                                 ptr(grp(p.rough, g)), ptr(grp(p.p_up, g)), ptr(dA), ptr(dBv), ptr(amax), ptr(g_r1[g]),
                                                 ptr(grp(g_pup, g)), K, h2, w2, 64, s), 'vfn_local_stats_backward_f32')
        g_p4 = torch.empty(N, p.h4, p.w4, 4, device=dev)
        check(L.vfn_upsample2x_add_backward_f32(ptr(g_pup), None, ptr(g_p4), N, h2, w2, 4, 0, s), 'vfn_upsample2x_add_backward_f32')
        g2, inputs = self.run(p, g_p4[..., :2].contiguous(), qs, slot, G)
        grads.update(g2)
        inputs['r1'] = g_r1
        return grads, inputs

    def _sum_objects(self, g, G=1):
        """[G * K, h, w, C] (frame-major) -> [G, h, w, C], summed over each frame's K objects in index order."""
        K = g.shape[0] // G
        g5 = g.view(G, K, *g.shape[1:])
        out = g5[:, 0].clone(memory_format=torch.contiguous_format)
        for k in range(1, K):
            out += g5[:, k]
        return out


# =====================================================================================================================
# The rest of the model: memory read, KeyValue, the two ResNet trunks (BatchNorm frozen, train_video_seg.py:103-106), stems
# =====================================================================================================================
class _ConvBwd:
    """One forward convolution seen from the backward pass: the packed filters of its data-gradient convolution (a frozen
    BatchNorm's scale folded in: d/dx of scale * conv(x) is conv(g * scale, flipped filters)) and its geometry."""

    def __init__(self, weight, stride, pad, dev, bn=None, reg=None):
        """``bn``: the frozen BatchNorm behind the convolution; ``reg`` (refresh.Refresher): filters and scale follow the
        parameters in place after an optimizer step."""
        w = weight.detach().float()
        scale = None if bn is None else _bn_scale(bn)
        self.cout, self.cin, self.k, _ = w.shape
        self.stride, self.pad = int(stride), int(pad)
        ws = w if scale is None else w * scale.detach().float().view(-1, 1, 1, 1).to(w.device)
        self.wp = _dgrad_filters(ws).to(dev)
        self.scale = None if scale is None else scale.detach().float().to(dev).contiguous()
        if reg is not None and isinstance(weight, torch.nn.Parameter):
            reg.add_filter(weight, self.wp, DGRAD, dst_ld=self.k * self.k * self.cout, bn=bn)
            if bn is not None:
                reg.add_epilogue(self.scale, None, bn=bn)


def _bn_scale(bn):
    return bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)


class ModelBackward:

    """dL/dscores -> the gradient of EVERY parameter of AFB_URR for one training sample (train_video_seg.py:65-74):
    decoder (DecoderBackward), memory read, KeyValue, query encoder; ``finish_memorize`` then carries the gradients that
    arrived at the bank's keys / values back through KeyValue and the memory encoder.  Gradients accumulate in ``self.grads``
    (state-dict name -> tensor)."""
    NB = 256
    # parameters whose weight gradient is accumulated per input-channel half (DecoderBackward.wgrad): name -> the halves' suffixes in
    # torch's channel order (AFB_URR.py:159 cat([mem, q_out]); :231 cat([r1, r1_local]))
    SPLIT = {'decoder.convFM.weight': ('#m', '#q'), 'decoder.local_convFM.weight': ('#r1', '#loc')}
    NB1 = max(1, min(NB, int(__import__('os').environ.get('VFN_COLSUM_BLOCKS', 128))))     # blocks of the one-launch column sums (<= NB: scratch rows are sized by NB) (128 / 256: 100.6 / 113.2 ms per step on one box; (the last block adds NB1 partial rows)

    def __init__(self, engine):
        self.eng = engine
        self.dev = engine.device
        self.dec = DecoderBackward(engine)
        m = engine.model
        dev = self.dev
        reg = engine.refresher
        self.cb = {}
        for enc_name, enc in (('encoder_q', m.encoder_q), ('encoder_m', m.encoder_m)):
            for lname in ('res2', 'res3', 'res4'):
                for bi, blk in enumerate(getattr(enc, lname)):
                    pre = f'{enc_name}.{lname}.{bi}'
                    self.cb[pre + '.conv1'] = _ConvBwd(blk.conv1.weight, 1, 0, dev, blk.bn1, reg)
                    self.cb[pre + '.conv2'] = _ConvBwd(blk.conv2.weight, blk.conv2.stride[0], 1, dev, blk.bn2, reg)
                    self.cb[pre + '.conv3'] = _ConvBwd(blk.conv3.weight, 1, 0, dev, blk.bn3, reg)
                    if hasattr(blk, 'downsample'):
                        self.cb[pre + '.downsample.0'] = _ConvBwd(blk.downsample[0].weight, blk.downsample[0].stride[0], 0, dev,
                                                                  blk.downsample[1], reg)
            # the 7x7 stem: only its weight gradient is taken (geometry + the frozen bn1 scale; no data-gradient filters needed)
            stem = _ConvBwd(torch.zeros(64, 3 if enc_name == 'encoder_q' else 5, 7, 7), 2, 3, dev, enc.bn1)
            reg.add_epilogue(stem.scale, None, bn=enc.bn1)
            self.cb[enc_name + '.stem'] = stem
        self._mean = torch.tensor(engine.mean, device=dev).view(1, 3, 1, 1)
        self._std = torch.tensor(engine.std, device=dev).view(1, 3, 1, 1)
        kv = m.keyval_r4
        wkv = torch.cat([kv.Key.weight.detach().float(), kv.Value.weight.detach().float()], 0)      # one 640-filter conv
        self.cb['keyval'] = _ConvBwd(wkv, 1, 1, dev)
        col = 0
        for c in (kv.Key, kv.Value):
            reg.add_filter(c.weight, self.cb['keyval'].wp, DGRAD, dst_ld=9 * wkv.shape[0], dst_col0=col, cout_ld=wkv.shape[0])
            col += c.weight.shape[0]
        self._grads = {}
        self._ticket = torch.zeros(64, dtype=torch.int32, device=dev)      # arrival counters (VFN_COLSUM_COUNTERS) of the one-launch column sums, zero at rest
        # convolution weight gradients accumulate IN THE KERNEL (vfn_conv_wgrad_f32, accumulate = 1) in the packed filter layout
        # [Cout][kh][kw][Cin]; ``grads`` hands them out as [Cout,Cin,kh,kw] views -- no torch add / copy per sample and layer
        self._packed = {}            # name -> (buffer [cout, k*k*cin], cout, k, cin)
        self.dec.sink = self         # the decoder's weight gradients go the same way
        # Weight / bias / BatchNorm-parameter gradients feed nothing but the optimizer, so they run on a side stream beside the
        # data-gradient chain (whose small layers leave most of the chip idle): launches are deferred (_side_do) and issued in
        # groups (_flush: one event per group) after the main stream has produced their operands; the main stream joins the
        # side stream before the next forward overwrites the activations and before the gradients are read (join).
        # (round 6: every side stream is PICKED so that it shares no hardware queue with the main stream, the forward's look-ahead
        # stream or another side stream -- _lib.independent_stream; streams that share one run in order whatever their events say)
        from ._lib import independent_stream
        fwd_side = engine.side_stream()
        self.side = independent_stream(dev, beside=[fwd_side], priority=_SIDE_PRIORITY) if _SIDE_WGRAD else None
        self._pending, self._inflight, self._by_plan = [], [], {}
        # round 5: with the decoder batched the side stream IS the step's critical path (12.8 ms of weight-gradient launches that cannot
        # start before the backward pass does, one after the other).  The memory encoder's (finish_memorize: the last to be produced,
        # parameters no other launch touches) go to a second side stream and run beside the query encoder's.  ``_lane`` = where
        # _side_do queues; per-lane scratch for the BatchNorm sums (the convolution kernels' workspaces are keyed by stream already)
        self.side2 = independent_stream(dev, beside=[self.side], priority=_SIDE_PRIORITY) if (_SIDE_WGRAD and _SIDE2) else None
        # (four hardware queues: the main stream and three side streams; the third shares the forward look-ahead stream's queue,
        # which is idle during a backward pass)
        self.side3 = independent_stream(dev, beside=[self.side, self.side2], priority=_SIDE_PRIORITY) if (_SIDE_WGRAD and _SIDE2 and _SIDE3) else None
        self._pending2, self._pending3, self._lane = [], [], 0        # lane 0: decoder + KeyValue, 1: memory encoder, 2: query encoder
        self._ticket2 = torch.zeros(64, dtype=torch.int32, device=dev)
        self._ticket3 = torch.zeros(64, dtype=torch.int32, device=dev)
        self._pad = {}               # zero-padded operand images of the memory read's small GEMMs (_gemm_nt)
        self._query_grads = {}       # slot -> gradients entering the memory read / the query encoder (batched samples: finish_query)
        self._batch_fb = None        # ... and the bank they were segmented against
        self._ticket_main = torch.zeros(64, dtype=torch.int32, device=dev)  # (the column sums that stay on the main stream)

    def reset(self):
        """Forget the gradients of the previous step (their tensors belong to whoever took them from ``grads``)."""
        self.join()
        self._grads = {}
        self._packed = {}
        self._query_grads = {}

    # ------------------------------------------------------------------ side stream
    def _side_do(self, fn):
        """Defer a launch whose result only the optimizer reads (the closure keeps its operand tensors alive)."""
        if _SIDE_DROP:                             # (measurement only: the step without its weight gradients = the main chain alone)
            return
        if self.side is None:
            fn()
        elif self._lane == 1 and self.side2 is not None:
            self._pending2.append(fn)
            if len(self._pending2) >= _SIDE_GROUP:
                self._flush()
        elif self._lane == 2 and self.side3 is not None:
            self._pending3.append(fn)
            if len(self._pending3) >= _SIDE_GROUP:
                self._flush()
        else:
            self._pending.append(fn)
            if len(self._pending) >= _SIDE_GROUP:
                self._flush()

    def _flush(self):
        """Issue the deferred launches on the side stream, behind everything the main stream has enqueued so far."""
        if not self._pending and not self._pending2 and not self._pending3:
            return
        ev = torch.cuda.current_stream().record_event()
        for side, pend in ((self.side, self._pending), (self.side2, self._pending2), (self.side3, self._pending3)):
            if pend:
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    for fn in pend:
                        fn()
                self._inflight += pend             # (operands stay referenced until the main stream has joined the side streams)
        self._pending, self._pending2, self._pending3 = [], [], []

    def join(self):
        """The main stream waits for the side stream: before the gradients are handed out (and before a forward overwrites
        activations whose readers are not tracked per plan)."""
        self._flush()
        if self._inflight or self._by_plan:
            torch.cuda.current_stream().wait_stream(self.side)
            for extra in (self.side2, self.side3):
                if extra is not None:
                    torch.cuda.current_stream().wait_stream(extra)
            self._inflight = []
            self._by_plan = {}

    def _end_sample(self, plan):
        """Everything deferred for the sample that used ``plan`` is on the side stream now: remember where it ends."""
        self._flush()
        if self._inflight:
            evs = [st.record_event() for st in (self.side, self.side2, self.side3) if st is not None]
            self._by_plan[id(plan)] = (evs, self._inflight)
            self._inflight = []

    def wait_plan(self, plan):
        """The main stream is about to overwrite ``plan``'s activations: wait for the side-stream launches that read them."""
        hit = self._by_plan.pop(id(plan), None)
        if hit is not None and not _NO_WAIT_PLAN:
            for ev in hit[0]:
                torch.cuda.current_stream().wait_event(ev)

    @property
    def grads(self):
        """state-dict name -> gradient, everything accumulated so far."""
        self.join()
        for name, (buf, cout, k, cin, rows) in list(self._packed.items()):
            g = buf.view(cout, k, k, cin)[:rows].permute(0, 3, 1, 2)
            if name in self._grads:
                self._grads[name] += g
            else:
                self._grads[name] = g
            del self._packed[name]
        for full, parts in self.SPLIT.items():                     # input-channel halves accumulated under their own names
            if all(full + s_ in self._grads for s_ in parts):
                g = torch.cat([self._grads.pop(full + s_) for s_ in parts], dim=1)
                self._grads[full] = self._grads[full] + g if full in self._grads else g
        return self._grads

    def wgrad_into(self, name, x, gy, k, stride, pad, cin, cout, ld_x, relu, rowscale, N, H, Wd, rows=None):
        """Accumulate dL/dW of one convolution for parameter ``name`` (see vfn_conv_wgrad_f32); False if the shapes need the
        round-3 path (channel counts that are not multiples of 32).  ``rows``: the parameter has only that many filters (the
        two-filter heads: their gradient arrives in a 32-channel tensor whose other channels are zero)."""
        if not (_IMPLICIT_WGRAD and cout >= 32 and (cin % 32 == 0 or cin < 32)) or name in self._grads:
            return False
        have = self._packed.get(name)
        if have is None:
            buf = torch.empty(cout, k * k * cin, device=self.dev)
            self._packed[name] = (buf, cout, k, cin, cout if rows is None else rows)
        else:
            buf = have[0]
        acc = have is not None
        if (_WINOGRAD_WGRAD and k == 3 and stride == 1 and pad == 1 and min(cin, cout) >= 32 and cin % 4 == 0 and cout % 4 == 0 and
                N * H * Wd * min(cin, cout) >= _WINOGRAD_WGRAD_MIN_WORK and gy.shape[-1] == cout):
            # the 3x3 layers with enough pixels x channels: the weight gradient in the Winograd domain (a quarter of the multiplies;
            # ops.conv_wgrad_winograd).  scripts/bench_wgrad_winograd.py: 2.2x at 20 000 pixels x 256 channels, 1.5x at 5 000 x 256 or
            # 50 000 x 64, break-even near 2 500 x 256; slower below
            self._side_do(lambda: ops.conv_wgrad_winograd(x, gy, cin=cin, cout=cout, ld_x=ld_x, relu=relu, rowscale=rowscale, out=buf,
                                                          accumulate=acc, N=N, H=H, W=Wd))
            return True
        self._side_do(lambda: ops.conv_wgrad(x, gy, k, stride, pad, cin=cin, cout=cout, ld_x=ld_x, relu=relu, rowscale=rowscale, out=buf,
                                             accumulate=acc, N=N, H=H, W=Wd))
        return True

    # ------------------------------------------------------------------ generic pieces
    def _acc(self, name, g):
        grads = self.grads if name in self._packed else self._grads         # (``grads`` joins the side stream first)
        g = g.contiguous()
        if name in grads:
            grads[name] += g
        else:
            grads[name] = g.clone()

    def _wacc(self, name, plan, x, gy, cb, N, H, Wd, relu=False):
        """dL/dW of the convolution ``cb`` into parameter ``name``."""
        if not self.wgrad_into(name, x, gy, cb.k, cb.stride, cb.pad, cb.cin, gy.shape[-1], x.shape[-1], relu, cb.scale, N, H, Wd):
            self._acc(name, self._wgrad(plan, x, gy, cb, N, H, Wd, relu=relu))

    def _dgrad(self, plan, cb, gy, N, H, Wd, mask=None, res=None, mask_after=False):
        """Data gradient of a k x k convolution at input resolution H x W (``gy`` already zero-inserted for stride 2)."""
        out = torch.empty(N, H, Wd, cb.cin, device=self.dev)
        d = ops.make_conv_desc(gy, cb.wp, cb.cin, cb.k, cb.k, 1, cb.k // 2, out, None, None, res, False, False, N=N, H=H, W=Wd)
        if mask is not None:
            d.mask, d.mask_ld, d.mask_after = ptr(mask), mask.shape[-1], int(mask_after)
        self.dec._launch(d, plan)
        return out

    def _dilate(self, g, H, Wd):
        N, Ho, Wo, C = g.shape
        out = torch.empty(N, H, Wd, C, device=self.dev)
        check(_lib.lib().vfn_dilate2_f32(ptr(g), ptr(out), N, Ho, Wo, H, Wd, C, stream()), 'vfn_dilate2_f32')
        return out

    def _wgrad(self, plan, x, gy, cb, N, H, Wd, relu=False, ld_x=None, cin=None):
        """dL/dW [Cout,Cin,k,k] of y = conv(act(x)) (k, stride, pad of ``cb``; x [N,H,W,*]) given gy [N,Ho,Wo,Cout]; a frozen
        BatchNorm's scale is applied to gy on the way (cb.scale)."""
        L = _lib.lib()
        k, s, pad = cb.k, cb.stride, cb.pad
        cin = cin if cin is not None else cb.cin
        cout = gy.shape[-1]
        Ho, Wo = gy.shape[1], gy.shape[2]
        M = N * Ho * Wo
        if _IMPLICIT_WGRAD and cout >= 32 and (cin % 32 == 0 or cin < 32):        # (cin < 32: the stems' 3 / 5 planes)
            dw = ops.conv_wgrad(x, gy, k, s, pad, cin=cin, cout=cout, ld_x=ld_x, relu=relu, rowscale=cb.scale, N=N, H=H, W=Wd)
            return dw.view(cout, k, k, cin).permute(0, 3, 1, 2)
        Mpad = (M + 31) // 32 * 32
        kk = k * k * cin
        rows = (kk + 255) // 256 * 256
        xt = self.dec._buf('xt', rows * Mpad).view(rows, Mpad)
        if rows > kk:
            xt[kk:].zero_()
        gt = self.dec._buf('gt', max(cout, 32) * Mpad).view(-1, Mpad)[:cout]
        check(L.vfn_transpose_taps_f32(ptr(x), N, H, Wd, cin, ld_x if ld_x is not None else x.shape[-1], int(relu), k, s, pad, Ho, Wo,
                                       None, ptr(xt), Mpad, stream()), 'vfn_transpose_taps_f32')
        check(L.vfn_transpose_taps_f32(ptr(gy), N, Ho, Wo, cout, gy.shape[-1], 0, 1, 1, 0, Ho, Wo, ptr(cb.scale), ptr(gt), Mpad, stream()),
              'vfn_transpose_taps_f32')
        dw = torch.empty(cout, kk, device=self.dev)
        d = ops.make_conv_desc(gt.view(1, 1, cout, Mpad), xt, kk, 1, 1, 1, 0, dw.view(1, 1, cout, kk), None, None, None, False, False,
                               N=1, H=1, W=cout)
        tiles = ops.conv_cfg_tiles()
        cfg = 0 if cout >= 128 else (3 if cout >= 64 else 4)                   # 128x128 / 64x64 / 32x64 tiles
        bm, bn = tiles[cfg]
        blocks = ((cout + bm - 1) // bm) * ((kk + bn - 1) // bn)
        splits = [q for q in ops.valid_splits(d, 16) if q * cout * kk <= plan.ws.numel()]
        want = max(1, 512 // max(1, blocks))
        ks = max([q for q in splits if q <= want] or [1])
        if kk % 4:
            ks = 1
        apply_choice(d, (cfg, ks, 0), plan.ws, None)
        ops.conv2d_launch(d, cfg, 0)
        return dw.view(cout, k, k, cin).permute(0, 3, 1, 2)

    def _small(self, name, C):
        """(running-gradient buffer of a bias / BatchNorm parameter, does it hold a value already)"""
        have = name in self._grads
        if not have:
            self._grads[name] = torch.empty(C, device=self.dev)
        return self._grads[name], int(have)

    def _colsum(self, g, C=None):
        C = C if C is not None else g.shape[-1]
        M = g.numel() // g.shape[-1]
        out = torch.empty(C, device=self.dev)
        part = self.dec._buf('colsum_main', self.NB * C)
        check(_lib.lib().vfn_colsum_acc_f32(ptr(g), M, C, g.shape[-1], ptr(part), self.NB1, ptr(out), 0, ptr(self._ticket_main), stream()),
              'vfn_colsum_acc_f32')
        return out

    def _bn_grads(self, name, bn, g, y, idn=None):
        """Gradients of a frozen BatchNorm's weight / bias (they stay trainable: only the statistics are frozen)."""
        C = g.shape[-1]
        M = g.numel() // C
        dg, acc_g = self._small(name + '.weight', C)                          # accumulated by the kernel: no torch add per sample
        db, acc_b = self._small(name + '.bias', C)
        assert acc_g == acc_b
        lane = self._lane if self._lane and (self.side2, self.side3)[self._lane - 1] is not None else 0
        part = self.dec._buf(('bn', 'bn2', 'bn3')[lane], 2 * self.NB1 * C)
        beta, gamma, ticket, nb1 = bn.bias.detach(), bn.weight.detach(), (self._ticket, self._ticket2, self._ticket3)[lane], self.NB1
        self._side_do(lambda: check(_lib.lib().vfn_bn_param_grads_acc_f32(ptr(g), ptr(y), ptr(idn), ptr(beta), ptr(gamma), M, C, ptr(part), nb1,
                                                                          ptr(dg), ptr(db), acc_g, ptr(ticket), stream()),
                                    'vfn_bn_param_grads_acc_f32'))

    # ------------------------------------------------------------------ ResNet bottleneck / trunk
    def _bottleneck(self, plan, pre, blk, a, N, g_pre, extra, mask_x):
        """torchvision Bottleneck (v1.5: the stride sits on conv2) backwards.  ``g_pre``: dL/d(out), already zero where the final
        ReLU clipped.  ``extra``: a further gradient arriving at the block's input (decoder skip connections).  Returns dL/dx,
        masked by x > 0 when x is itself a ReLU output (``mask_x``)."""
        x, t1, t2, out, s, H, Wd = a['x'], a['t1'], a['t2'], a['out'], a['stride'], a['H'], a['W']
        c1, c2, c3 = self.cb[pre + '.conv1'], self.cb[pre + '.conv2'], self.cb[pre + '.conv3']
        Ho, Wo = H // s, Wd // s
        idn = a['ds'] if a['ds'] is not None else x
        # conv3 + bn3 (+ idn) + relu
        self._bn_grads(pre + '.bn3', blk.bn3, g_pre, out, idn)
        self._wacc(pre + '.conv3.weight', plan, t2, g_pre, c3, N, Ho, Wo)
        g_t2 = self._dgrad(plan, c3, g_pre, N, Ho, Wo, mask=t2)
        # conv2 + bn2 + relu
        self._bn_grads(pre + '.bn2', blk.bn2, g_t2, t2)
        self._wacc(pre + '.conv2.weight', plan, t1, g_t2, c2, N, H, Wd)
        g_t1 = self._dgrad(plan, c2, self._dilate(g_t2, H, Wd) if s == 2 else g_t2, N, H, Wd, mask=t1)
        # identity branch
        if a['ds'] is not None:
            cd = self.cb[pre + '.downsample.0']
            self._bn_grads(pre + '.downsample.1', blk.downsample[1], g_pre, a['ds'])
            self._wacc(pre + '.downsample.0.weight', plan, x, g_pre, cd, N, H, Wd)
            side = self._dgrad(plan, cd, self._dilate(g_pre, H, Wd) if s == 2 else g_pre, N, H, Wd, res=extra)
        else:
            side = g_pre if extra is None else g_pre + extra
        # conv1 + bn1 + relu, joined with the identity branch; then the ReLU that produced x
        self._bn_grads(pre + '.bn1', blk.bn1, g_t1, t1)
        self._wacc(pre + '.conv1.weight', plan, x, g_t1, c1, N, H, Wd)
        return self._dgrad(plan, c1, g_t1, N, H, Wd, mask=x if mask_x else None, res=side, mask_after=True)

    def _trunk(self, plan, enc_name, enc, acts, bufs, N, g_r4, extras):
        """res4 .. res2, max-pool, back to dL/d(bn1 output) of the stem.  ``g_r4`` is masked by r4 > 0 already; ``extras``:
        gradients the decoder sends to r3 / r2 / r1 ([1,...] tensors or None)."""
        g = g_r4
        prev = {'res4': 'res3', 'res3': 'res2', 'res2': None}
        for lname in ('res4', 'res3', 'res2'):
            blocks = getattr(enc, lname)
            for bi in reversed(range(len(blocks))):
                extra = extras.get(prev[lname]) if bi == 0 and prev[lname] else None
                g = self._bottleneck(plan, f'{enc_name}.{lname}.{bi}', blocks[bi], acts[(lname, bi)], N, g, extra,
                                     mask_x=not (lname == 'res2' and bi == 0))
        r1 = bufs['r1']
        g_r1 = torch.empty_like(r1)
        check(_lib.lib().vfn_maxpool3x3s2_backward_f32(ptr(r1), ptr(g), ptr(g_r1), N, plan.h2, plan.w2, 64, ptr(extras.get('r1')), 1,
                                                       stream()), 'vfn_maxpool3x3s2_backward_f32')
        return g_r1

    def _stem(self, plan, enc_name, enc, xn, g_c1, r1, N, names):
        """conv1 (7x7, stride 2, pad 3; + conv1_m / conv1_o for the memory encoder: one convolution over the concatenated
        planes) + bn1.  xn: the padded, normalised input planes [N,Hp,Wp,C]; g_c1: dL/d(bn1 output), masked."""
        self._bn_grads(enc_name + '.bn1', enc.bn1, g_c1, r1)
        C = xn.shape[-1]
        cb = self.cb[enc_name + '.stem']                                       # (geometry + the frozen bn1 scale; built once)
        assert cb.cin == C
        dw = self._wgrad(plan, xn, g_c1, cb, N, plan.Hp, plan.Wp)              # [64, C, 7, 7]
        c0 = 0
        for name, nc in names:
            self._acc(name, dw[:, c0:c0 + nc])
            c0 += nc

    def _normalised_input(self, plan, frame, mask=None):
        """pad_divide_by (myutils/data.py:132-149: zeros, BEFORE the normalisation) + (x - mean) / std, NHWC.  Data preparation
        for the stem's weight gradient; plain tensor ops."""
        lw, uw, lh, uh = plan.pad
        e = self.eng
        mean, std = self._mean, self._std                                      # (device constants: no blocking copy per sample)
        f = torch.nn.functional.pad(frame, (lw, uw, lh, uh))
        f = (f - mean) / std
        if mask is None:
            return f.permute(0, 2, 3, 1).contiguous()
        K = mask.shape[1]
        mk = torch.nn.functional.pad(mask.float(), (lw, uw, lh, uh))[0]        # [K,Hp,Wp], zero in the padding
        inv = (1.0 - mk).clamp(0, 1)                                           # AFB_URR.py:262-264 (one in the padding)
        planes = [torch.cat([f[0], mk[k:k + 1], inv[k:k + 1]], 0) for k in range(K)]
        return torch.stack(planes, 0).permute(0, 2, 3, 1).contiguous()

    # ------------------------------------------------------------------ memory read (the bank = one frame, materialised P)
    def _padded(self, key, rows, cols):
        t = self._pad.get(key)
        if t is None:
            if len(self._pad) >= 32:
                self._pad.clear()
            t = self._pad[key] = torch.zeros(rows, cols, device=self.dev)
        return t

    def _gemm_nt(self, plan, A, Bm, M, Nn, Kd):
        """C [M,Nn] = A [M,Kd] @ Bm [Nn,Kd]^T through the forward kernel (a 1x1 problem); operands zero-padded to its granules."""
        Kp = (Kd + 31) // 32 * 32
        rows = (Nn + 255) // 256 * 256
        # padded operand images live across calls (zeroed once; only the valid region is rewritten, keyed by it), and an operand
        # that already has the kernel's layout is used where it lies: 100 fill + 50 copy launches per step less
        if Kp == Kd and A.is_contiguous():
            Ap = A
        else:
            Ap = self._padded(('A', M, Kd), M, Kp)
            Ap[:, :Kd] = A
        Bp = self._padded(('B', Nn, Kd), rows, Kp)
        Bp[:Nn, :Kd] = Bm
        ldo = (Nn + 3) // 4 * 4
        Cm = torch.empty(M, ldo, device=self.dev)
        d = ops.make_conv_desc(Ap.view(1, 1, M, Kp), Bp, Nn, 1, 1, 1, 0, Cm.view(1, 1, M, ldo), None, None, None, False, False, N=1, H=1, W=M)
        d.out_ld = ldo
        cfg = 3
        apply_choice(d, (cfg, 1, 0), plan.ws, None)
        ops.conv2d_launch(d, cfg, 0)
        return Cm[:, :Nn]

    def memory_read(self, plan, fb, q, g_mem):
        """Matcher.forward backwards (AFB_URR.py:136-146) for a bank that holds one frame (training: init_bank only).
        q [HW,128] query keys, g_mem [K,HW,512] = dL/d(mem).  Returns dL/dq [HW,128], dL/d(bank keys) [K,B,128],
        dL/d(bank values) [K,B,512]."""
        L = _lib.lib()
        K = fb.obj_n
        lens = fb._sync_len()
        HW = q.shape[0]
        scale = 1.0 / (DK ** 0.5)
        g_q = torch.zeros(HW, DK, device=self.dev)
        g_k, g_v = [], []
        for k in range(K):
            B = lens[k]
            Kb, Vb = fb._kbuf[k, :B], fb._vbuf[k, :B]
            g = g_mem[k]                                                       # [HW,512]
            S = self._gemm_nt(plan, Kb, q, B, HW, DK).contiguous()             # [B,HW]
            P = torch.empty_like(S)
            check(L.vfn_softmax_cols_f32(ptr(S), B, HW, HW, scale, ptr(P), stream()), 'vfn_softmax_cols_f32')
            g_v.append(self._gemm_nt(plan, P, g.t(), B, DV, HW))               # dV = P dO
            dP = self._gemm_nt(plan, Vb, g, B, HW, DV).contiguous()            # dP = V dO^T
            dS = torch.empty_like(S)
            check(L.vfn_softmax_cols_backward_f32(ptr(P), ptr(dP), B, HW, HW, scale, ptr(dS), stream()), 'vfn_softmax_cols_backward_f32')
            g_q += self._gemm_nt(plan, dS.t(), Kb.t(), HW, DK, B)              # dq = dS^T K
            g_k.append(self._gemm_nt(plan, dS, q.t(), B, DK, HW))              # dK = dS q
        return g_q, g_k, g_v

    # ------------------------------------------------------------------ one sample
    @torch.no_grad()
    def segment_sample(self, fb, grad_score, query=None):
        """Backward of the sample ``segment`` ran last (its activations are in the training plan; ``query``: the (plan, query
        set, slot) triple of another segment call, ``engine.last_query`` right after it).  grad_score = dloss/dscores
        [obj_n,H0,W0].  Accumulates parameter gradients; returns (dL/d bank keys, dL/d bank values) for ``finish_memorize``.
        When the sample's frames went through ``Engine.query_batch``, the query encoder's part is left to ``finish_query`` (the
        whole batch at once) and only the decoder / memory-read part runs here."""
        plan, qs, slot = query if query is not None else self.eng.last_query
        if not plan.keep_acts:
            raise RuntimeError('backward needs the training plan: call model.train() before memorize / segment')
        m = self.eng.model
        K = plan.obj_n
        g_dec, gin = self.dec.run_tail(plan, grad_score, qs, slot)
        for n_, g_ in g_dec.items():
            if g_ is not None:                                                # (None: accumulated by the kernel, wgrad_into)
                self._acc(n_, g_)
        batch = self.eng._batch
        if batch is not None and batch[1] is qs and qs.nq == qs.n:
            # gradients that enter the memory read and the query encoder: collected, differentiated for all frames of the sample in
            # finish_query (which also returns the bank's gradients: the bank is the same for every frame of the sample)
            self._query_grads[slot] = (gin['mem'].reshape(K, plan.HW, DV), gin['q_out'].reshape(1, plan.HW, DV), gin['r3'], gin['r2'], gin['r1'])
            self._batch_fb = fb
            self._end_sample(plan)
            return None, None
        kvq = qs.kv_q[slot]                                                   # [HW,640]
        g_qk, g_bk, g_bv = self.memory_read(plan, fb, kvq[:, :DK].contiguous(), gin['mem'].reshape(K, plan.HW, DV))
        # KeyValue on the query side: dL/d[key | value]
        g_kv = torch.cat([g_qk, gin['q_out'].reshape(plan.HW, DV)], dim=1).view(1, plan.h16, plan.w16, DK + DV).contiguous()
        if slot != 0 or qs.stage != 0:
            raise RuntimeError('backward expects the frame-only part of segment to have run in place (no look-ahead in training)')
        acts = qs.acts[1]
        bufs = {'r1': qs.q['r1'][0:1]}
        r4 = acts[('res4', len(m.encoder_q.res4) - 1)]['out']
        self._keyval(plan, r4, g_kv, 1)
        g_r4 = self._dgrad(plan, self.cb['keyval'], g_kv, 1, plan.h16, plan.w16, mask=r4)
        g_c1 = self._trunk(plan, 'encoder_q', m.encoder_q, acts, bufs, 1, g_r4, {'res3': gin['r3'], 'res2': gin['r2'], 'r1': gin['r1']})
        xn = self._normalised_input(plan, qs.frames[slot:slot + 1])
        self._stem(plan, 'encoder_q', m.encoder_q, xn, g_c1, bufs['r1'], 1, [('encoder_q.conv1.weight', 3)])
        self._end_sample(plan)
        return g_bk, g_bv

    @torch.no_grad()
    def segment_batch(self, fb, grad_scores):
        """Backward of ``Engine.segment_batch`` (all frames of the sample through the decoder at once): grad_scores = dloss/dscores
        [n, obj_n, H0, W0].  The decoder's data and weight gradients run over n * obj_n images (DecoderBackward.run_tail with G = n);
        what enters the memory read and the query encoder is left to ``finish_query``, as for ``segment_sample`` on a batched sample."""
        b, qs = self.eng.last_batch
        plan = b.plan
        G, K = qs.n, plan.obj_n
        g_dec, gin = self.dec.run_tail(b, grad_scores, qs, 0, G)
        for n_, g_ in g_dec.items():
            if g_ is not None:                                                # (None: accumulated by the kernel, wgrad_into)
                self._acc(n_, g_)
        # dL/dmem arrives frame-major [n * K, HW, 512]; the memory read differentiates object by object over all frames' queries
        g_mem = gin['mem'].view(G, K, plan.HW, DV).permute(1, 0, 2, 3).reshape(K, G * plan.HW, DV)
        self._query_grads = {'batch': (g_mem, gin['q_out'].view(G, plan.HW, DV), gin['r3'], gin['r2'], gin['r1'])}
        self._batch_fb = fb
        self._end_sample(b)

    @torch.no_grad()
    def finish_query(self):
        """The query encoder (KeyValue, res4 .. res2, stem) backwards for ALL frames of the sample that ``Engine.query_batch`` ran
        in one pass: the per-sample gradients ``segment_sample`` collected are stacked along the batch axis, so every data- and
        weight-gradient launch sees n times the pixels (the 1/16-resolution layers have 625 of them per frame at 400 x 400).  The
        memory read is differentiated here too, for all frames at once; returns (dL/d bank keys, dL/d bank values) of the sample
        (``segment_sample`` returned (None, None) for its frames), or (None, None) when nothing was batched."""
        if not self._query_grads:
            return None, None
        plan, qs = self.eng._batch
        n = qs.n
        m = self.eng.model
        if 'batch' in self._query_grads:                                       # (segment_batch: already stacked)
            g_mem, g_qv, g3, g2, g1 = self._query_grads['batch']
        else:
            if sorted(self._query_grads) != list(range(n)):
                raise RuntimeError(f'finish_query: gradients for slots {sorted(self._query_grads)} of a batch of {n}')
            parts = [self._query_grads[i] for i in range(n)]
            g_qv, g3, g2, g1 = (torch.cat([t[j] for t in parts], dim=0) for j in range(1, 5))
            g_mem = torch.cat([t[0] for t in parts], dim=1)                   # [K, n*HW, 512]
        # the memory read backwards for the n frames at once: n * HW query columns against the one bank -- six GEMMs and two softmax
        # kernels per object instead of per object and frame (0.48 ms per frame of 625 pixels, launch-bound)
        fb, self._batch_fb = self._batch_fb, None
        q_keys = qs.kv_q[0:n].reshape(n * plan.HW, DK + DV)[:, :DK].contiguous()
        g_qk, g_bk, g_bv = self.memory_read(plan, fb, q_keys, g_mem)
        g_kv = torch.cat([g_qk.view(n, plan.HW, DK), g_qv], dim=2).view(n, plan.h16, plan.w16, DK + DV)
        self._query_grads = {}
        acts = qs.acts[n]
        bufs = {'r1': qs.q['r1'][0:n]}
        r4 = acts[('res4', len(m.encoder_q.res4) - 1)]['out']
        self._keyval(plan, r4, g_kv, n)              # (KeyValue: first side stream, with the memory side's)
        g_r4 = self._dgrad(plan, self.cb['keyval'], g_kv, n, plan.h16, plan.w16, mask=r4)
        self._flush()
        self._lane = 2                               # the query encoder's own parameters: third side stream
        try:
            g_c1 = self._trunk(plan, 'encoder_q', m.encoder_q, acts, bufs, n, g_r4, {'res3': g3, 'res2': g2, 'r1': g1})
            xn = self._normalised_input(plan, qs.frames[0:n])
            self._stem(plan, 'encoder_q', m.encoder_q, xn, g_c1, bufs['r1'], n, [('encoder_q.conv1.weight', 3)])
        finally:
            self._lane = 0
        self._flush()
        return g_bk, g_bv

    def _keyval(self, plan, r4, g_kv, N):
        m = self.eng.model
        db = self._colsum(g_kv)
        ok_k = self.wgrad_into('keyval_r4.Key.weight', r4, g_kv[..., :DK], 3, 1, 1, r4.shape[-1], DK, r4.shape[-1], False, None, N, plan.h16, plan.w16)
        ok_v = self.wgrad_into('keyval_r4.Value.weight', r4, g_kv[..., DK:], 3, 1, 1, r4.shape[-1], DV, r4.shape[-1], False, None, N, plan.h16, plan.w16)
        if not (ok_k and ok_v):
            assert not ok_k and not ok_v
            dw = self._wgrad(plan, r4, g_kv, self.cb['keyval'], N, plan.h16, plan.w16)            # [640,1024,3,3]
            self._acc('keyval_r4.Key.weight', dw[:DK])
            self._acc('keyval_r4.Value.weight', dw[DK:])
        self._acc('keyval_r4.Key.bias', db[:DK])
        self._acc('keyval_r4.Value.bias', db[DK:])

    @torch.no_grad()
    def finish_memorize(self, frame, mask, g_bank_k, g_bank_v):
        """The gradients that reached the bank's keys / values (summed over the samples) back through ``memorize``
        (AFB_URR.py:255-272): KeyValue and the memory encoder.  frame [1,3,H0,W0], mask [1,K,H0,W0] as given to memorize."""
        plan = self.eng.last_memorize
        m = self.eng.model
        K = plan.obj_n
        g_kv = torch.stack([torch.cat([g_bank_k[k], g_bank_v[k]], dim=1) for k in range(K)], 0).view(K, plan.h16, plan.w16, DK + DV).contiguous()
        acts = plan.acts_m
        r4 = acts[('res4', len(m.encoder_m.res4) - 1)]['out']
        self._keyval(plan, r4, g_kv, K)              # (KeyValue is shared with the query side: its gradients stay on the first side stream)
        g_r4 = self._dgrad(plan, self.cb['keyval'], g_kv, K, plan.h16, plan.w16, mask=r4)
        self._flush()
        self._lane = 1                               # the memory encoder's own parameters: second side stream
        try:
            g_c1 = self._trunk(plan, 'encoder_m', m.encoder_m, acts, plan.m, K, g_r4, {})
            xn = self._normalised_input(plan, frame, mask)
            self._stem(plan, 'encoder_m', m.encoder_m, xn, g_c1, plan.m['r1'], K,
                       [('encoder_m.conv1.weight', 3), ('encoder_m.conv1_m.weight', 1), ('encoder_m.conv1_o.weight', 1)])
        finally:
            self._lane = 0
