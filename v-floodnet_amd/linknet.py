"""First-frame bootstrap model on the HIP path: ``smp.Linknet(encoder_name='efficientnet-b4', classes=1, activation='sigmoid')``
(train_image_seg.py:82-89), the model ``test_image_seg.test_waterseg`` unpickles (test_image_seg.py:133) when
``test_video_seg.py:67-69`` finds no first-frame mask.

    model = LinknetB4.from_checkpoint('records/link_efficientb4_model.pth', device)   # or LinknetB4(device); load_state_dict(sd)
    prob = model.predict(x)            # x f32 [1,3,416,416] ImageNet-normalised -> f32 [1,1,416,416] in [0,1]  (the smp API)

The module is a parameter container with the state-dict names of segmentation_models_pytorch 0.2.0 / efficientnet-pytorch 0.6.3
(``encoder._blocks.7._se_reduce.weight``, ``decoder.blocks.2.block.1.0.bias``, ``segmentation_head.0.weight`` ...), so the state dict
of the reference's pickled model loads as it is; ``predict`` runs entirely in HIP kernels:

* every 1x1 convolution (expand / project / decoder) and the decoder's ConvTranspose2d(4, 2, 1) -- a stride-1 4x4 convolution of the
  zero-inserted input with flipped filters -- through the implicit-GEMM kernel (``vfn_conv2d_nhwc_f32``) on NHWC tensors whose
  channel counts are padded to multiples of 32 (zero filters, scale and shift in the padding, so padded channels stay 0);
* eval-mode BatchNorm in the convolutions' epilogues; swish is applied by the consumer of an expand convolution (the depthwise
  kernel reads ``swish(x)``), so no activation pass exists;
* the squeeze-excite gate multiplies the project convolution's input channels = the columns of its filter matrix
  (``vfn_ln_scale_cols_f32``) instead of making a pass over the activation tensor;
* stem, depthwise convolutions, gate, skip adds, head: ``csrc/linknet_ops.hip``.

**Parity unpinned**: both packages and the trained weights are absent here and the reference holds no golden output for this model;
the architecture is restated from the packages' published definitions (``oracle/linknet_ref.py`` is the same restatement in torch
ops, which ``tests/test_linknet*.py`` compare this path with, on synthetic weights).  There is no CPU fallback.
"""
import math

import torch
from torch import nn

from . import _lib, ops, weights as W
from ._lib import ptr, stream, check
from .engine import choose_cfg, apply_choice, WS_FLOATS

BN_EPS_ENC, BN_EPS_DEC = 1e-3, 1e-5
# (repeats, kernel, stride, expand, in, out): EfficientNet-B4 = B0's stages under width 1.4 / depth 1.8 (efficientnet_pytorch/utils.py)
STAGES = ((2, 3, 1, 1, 48, 24), (4, 3, 2, 6, 24, 32), (4, 5, 2, 6, 32, 56), (6, 3, 2, 6, 56, 112),
          (6, 5, 1, 6, 112, 160), (8, 5, 2, 6, 160, 272), (2, 3, 1, 6, 272, 448))
STAGE_IDXS = (6, 10, 22, 32)                     # smp encoders/efficientnet.py: features after these many blocks
DEC_CHANNELS = (448, 160, 56, 32, 48, 32)        # linknet/decoder.py: encoder channels deepest first + prefinal_channels


def _blocks():
    out = []
    for rep, k, s, e, cin, cout in STAGES:
        for r in range(rep):
            ci = cin if r == 0 else cout
            out.append(dict(k=k, s=s if r == 0 else 1, e=e, cin=ci, cout=cout, sq=max(1, int(ci * 0.25))))
    return out


def _same_pad_before(k, s):
    """Zeros before the image of efficientnet-pytorch 0.6.3's static "same" padding (computed for the native 380-pixel input at
    every layer): (k-1)/2 for stride 1; stride 2: 0 for k = 3, 1 for k = 5 (the larger half falls after the image)."""
    ih = 380
    p = max((math.ceil(ih / s) - 1) * s + k - ih, 0)
    return p // 2


def _cp(c):
    return (c + 31) // 32 * 32


class _MBConv(nn.Module):
    def __init__(self, b):
        super().__init__()
        oup = b['cin'] * b['e']
        if b['e'] != 1:
            self._expand_conv = nn.Conv2d(b['cin'], oup, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(oup, eps=BN_EPS_ENC)
        self._depthwise_conv = nn.Conv2d(oup, oup, b['k'], stride=b['s'], groups=oup, bias=False)
        self._bn1 = nn.BatchNorm2d(oup, eps=BN_EPS_ENC)
        self._se_reduce = nn.Conv2d(oup, b['sq'], 1)
        self._se_expand = nn.Conv2d(b['sq'], oup, 1)
        self._project_conv = nn.Conv2d(oup, b['cout'], 1, bias=False)
        self._bn2 = nn.BatchNorm2d(b['cout'], eps=BN_EPS_ENC)


class _Encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self._conv_stem = nn.Conv2d(3, 48, 3, stride=2, bias=False)
        self._bn0 = nn.BatchNorm2d(48, eps=BN_EPS_ENC)
        self._blocks = nn.ModuleList([_MBConv(b) for b in _blocks()])
        self._conv_head = nn.Conv2d(448, 1792, 1, bias=False)      # in smp's state dict, not in its encoder's forward
        self._bn1 = nn.BatchNorm2d(1792, eps=BN_EPS_ENC)


class _DecoderBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        m = cin // 4
        self.block = nn.Sequential(
            nn.Sequential(nn.Conv2d(cin, m, 1, bias=False), nn.BatchNorm2d(m), nn.ReLU()),
            nn.Sequential(nn.ConvTranspose2d(m, m, 4, stride=2, padding=1), nn.BatchNorm2d(m), nn.ReLU()),
            nn.Sequential(nn.Conv2d(m, cout, 1, bias=False), nn.BatchNorm2d(cout), nn.ReLU()))


class _Decoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.blocks = nn.ModuleList([_DecoderBlock(DEC_CHANNELS[j], DEC_CHANNELS[j + 1]) for j in range(5)])


def _bn_consts(bn, cpad, dev, conv_bias=None):
    """(scale, shift) of an eval-mode BatchNorm over ``cpad`` channels (zeros in the padding); a bias of the convolution in
    front of it is folded into the shift."""
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)      # (the encoder's eps is 1e-3)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    if conv_bias is not None:
        shift = shift + scale * conv_bias.detach().float().to(scale.device)
    s = torch.zeros(cpad, device=dev)
    h = torch.zeros(cpad, device=dev)
    s[:scale.numel()] = scale.to(dev)
    h[:shift.numel()] = shift.to(dev)
    return s, h


def _pack_1x1(w, cin_p, cout_p, dev):
    """[cout, cin, 1, 1] -> packed [rows, cin_p] with zero rows / columns in the padding."""
    cout, cin = w.shape[0], w.shape[1]
    full = torch.zeros(cout_p, cin_p, device=dev)
    full[:cout, :cin] = w.detach().float().view(cout, cin).to(dev)
    return ops.pad_rows(full)


class LinknetB4(nn.Module):
    def __init__(self, device=None):
        super().__init__()
        self.encoder = _Encoder()
        self.decoder = _Decoder()
        self.segmentation_head = nn.Sequential(nn.Conv2d(32, 1, 1), nn.Identity(), nn.Identity())
        self._packed = None
        self._graphs, self._graph_runs = {}, {}        # (H, W, logits) -> (captured graph, static input, static output) / eager calls so far
        if device is not None:
            self.to(device)
        self.eval()

    # -- weight lifecycle ---------------------------------------------------
    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        self._packed = None
        self._graphs, self._graph_runs = {}, {}        # (the captured launches point at the packed weights)
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._packed = None
        self.__dict__['_graphs'], self.__dict__['_graph_runs'] = {}, {}
        return out

    @classmethod
    def from_checkpoint(cls, path_or_obj, device):
        """The reference's ``torch.load(model_path)`` (test_image_seg.py:133) returns the pickled smp module (its package must be
        importable to unpickle it); a saved state dict works as well.  Either way the parameters are read by name."""
        obj = path_or_obj
        if isinstance(path_or_obj, (str, bytes)) or hasattr(path_or_obj, 'read'):
            try:                                   # a plain state dict loads without executing anything from the file
                obj = torch.load(path_or_obj, map_location='cpu', weights_only=True)
            except Exception:
                if hasattr(path_or_obj, 'seek'):
                    path_or_obj.seek(0)
                try:                               # the reference's file: a whole pickled module (test_image_seg.py:133) -- as unsafe
                    obj = torch.load(path_or_obj, map_location='cpu', weights_only=False)        # on untrusted files as it is there
                except ModuleNotFoundError as e:
                    raise RuntimeError(f'{path_or_obj}: the file is a pickled module of a package that is not installed ({e.name}); '
                                       'install it to unpickle, or save model.state_dict() and pass that file') from e
        sd = obj.state_dict() if hasattr(obj, 'state_dict') else obj
        if isinstance(sd, dict) and 'model' in sd and 'encoder._conv_stem.weight' not in sd:
            sd = sd['model']
        m = cls()
        m.load_state_dict(sd, strict=True)
        return m.to(device)

    # -- packing --------------------------------------------------------------
    def _pack(self):
        dev = next(self.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('LinknetB4 runs on hand-written HIP kernels only: move the model to the GPU (no CPU fallback)')
        _lib.lib()
        e = self.encoder
        P = dict(dev=dev)
        P['stem_w'] = e._conv_stem.weight.detach().float().contiguous().to(dev)
        P['stem_sc'], P['stem_sh'] = _bn_consts(e._bn0, 64, dev)
        blocks = []
        for b, m in zip(_blocks(), e._blocks):
            oup = b['cin'] * b['e']
            cin_p, oup_p, cout_p = _cp(b['cin']), _cp(oup), _cp(b['cout'])
            q = dict(b, oup=oup, cin_p=cin_p, oup_p=oup_p, cout_p=cout_p)
            if b['e'] != 1:
                q['exp_w'] = _pack_1x1(m._expand_conv.weight, cin_p, oup_p, dev)
                q['exp_sc'], q['exp_sh'] = _bn_consts(m._bn0, oup_p, dev)
            k = b['k']
            dw = torch.zeros(k * k, oup_p, device=dev)
            dw[:, :oup] = m._depthwise_conv.weight.detach().float().view(oup, k * k).t().to(dev)
            q['dw_w'] = dw.contiguous()
            q['dw_sc'], q['dw_sh'] = _bn_consts(m._bn1, oup_p, dev)
            q['se_w1'] = m._se_reduce.weight.detach().float().view(b['sq'], oup).contiguous().to(dev)
            q['se_b1'] = m._se_reduce.bias.detach().float().contiguous().to(dev)
            q['se_w2'] = m._se_expand.weight.detach().float().view(oup, b['sq']).contiguous().to(dev)
            q['se_b2'] = m._se_expand.bias.detach().float().contiguous().to(dev)
            q['proj_w'] = _pack_1x1(m._project_conv.weight, oup_p, cout_p, dev)
            q['proj_sc'], q['proj_sh'] = _bn_consts(m._bn2, cout_p, dev)
            q['pad_b'] = _same_pad_before(k, b['s'])
            blocks.append(q)
        P['blocks'] = blocks
        dec = []
        for j, blk in enumerate(self.decoder.blocks):
            cin, cout = DEC_CHANNELS[j], DEC_CHANNELS[j + 1]
            mid = cin // 4
            cin_p, mid_p, cout_p = _cp(cin), _cp(mid), _cp(cout)
            q = dict(cin_p=cin_p, mid_p=mid_p, cout_p=cout_p)
            q['a_w'] = _pack_1x1(blk.block[0][0].weight, cin_p, mid_p, dev)
            q['a_sc'], q['a_sh'] = _bn_consts(blk.block[0][1], mid_p, dev)
            # ConvTranspose2d weight [in, out, 4, 4]: the equivalent convolution over the zero-inserted input has filters
            # Wc[o][i][kh][kw] = Wt[i][o][3-kh][3-kw]
            wt = blk.block[1][0].weight.detach().float().to(dev)
            wc = torch.zeros(mid_p, mid_p, 4, 4, device=dev)
            wc[:mid, :mid] = wt.flip(2, 3).transpose(0, 1)
            q['t_w'] = ops.pad_rows(W.pack_conv_weight(wc))
            q['t_sc'], q['t_sh'] = _bn_consts(blk.block[1][1], mid_p, dev, conv_bias=blk.block[1][0].bias)
            q['c_w'] = _pack_1x1(blk.block[2][0].weight, mid_p, cout_p, dev)
            q['c_sc'], q['c_sh'] = _bn_consts(blk.block[2][1], cout_p, dev)
            dec.append(q)
        P['dec'] = dec
        hw = torch.zeros(32, device=dev)
        hw[:32] = self.segmentation_head[0].weight.detach().float().view(32).to(dev)
        P['head_w'] = hw
        P['head_b'] = float(self.segmentation_head[0].bias.detach().float().cpu())
        P['ws'] = torch.empty(WS_FLOATS, device=dev)
        self._packed = P
        return P

    # -- launches ---------------------------------------------------------------
    @staticmethod
    def _conv(P, x, wp, cout_p, k, pad, out, scale, shift, N, H, Wd, res=None, relu_out=False, Ho=None, Wo=None):
        d = ops.make_conv_desc(x, wp, cout_p, k, k, 1, pad, out, scale, shift, res, False, relu_out, N=N, H=H, W=Wd)
        if Ho is not None:
            d.Ho, d.Wo, d.M = Ho, Wo, N * Ho * Wo
        choice = choose_cfg(d.M, d.Cout, d.KH * d.KW * d.Cin, 0)
        if choice[1] > 1 and (d.out_ld % 4 or (d.res and d.res_ld % 4)):
            choice = (choice[0], 1, 0)
        cfg = apply_choice(d, choice, P['ws'], None)
        ops.conv2d_launch(d, cfg, 0)

    @torch.no_grad()
    def predict(self, x, logits=False):
        """x f32 [1,3,H,W] (H, W multiples of 32) on the GPU -> probabilities f32 [1,1,H,W] (``smp`` ``model.predict``)."""
        _lib.require_gpu(x, 'x')
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f'expected a normalised RGB batch [N,3,H,W], got {tuple(x.shape)}')
        if x.shape[2] % 32 or x.shape[3] % 32:
            raise RuntimeError(f'input height and width must be divisible by 32 (5 stride-2 stages), got {tuple(x.shape[2:])}')
        if x.shape[0] != 1:
            return torch.cat([self.predict(x[i:i + 1], logits) for i in range(x.shape[0])], 0)
        P = self._packed or self._pack()
        # ≈ 230 launches of a few microseconds each, paced by the host (3.3 ms per 416 x 416 call): from the third call at one input size
        # on they are ONE captured HIP graph per (size, logits) -- static input / output buffers, the intermediates in the graph's pool
        # (round 5; VFN_GRAPHS=0 or a failed capture: the launch-by-launch path below)
        from .engine import _GRAPHS
        key = (x.shape[2], x.shape[3], bool(logits))
        g = self._graphs.get(key) if _GRAPHS else None
        if g is not None:
            g[1].copy_(x)
            g[0].replay()
            return g[2].clone()
        if _GRAPHS:
            n = self._graph_runs.get(key, 0)
            self._graph_runs[key] = n + 1
            if n == 2:
                try:
                    xs = x.float().contiguous().clone()
                    graph = torch.cuda.CUDAGraph()
                    cur = torch.cuda.current_stream()
                    cap = torch.cuda.Stream(device=cur.device)
                    cap.wait_stream(cur)
                    with torch.cuda.graph(graph, stream=cap, capture_error_mode='thread_local'):
                        out_s = self._predict_eager(P, xs, logits)
                    cur.wait_stream(cap)
                    self._graphs[key] = (graph, xs, out_s)
                    graph.replay()
                    return out_s.clone()
                except Exception:
                    import warnings
                    warnings.warn('vfloodnet_amd: HIP graph capture of LinknetB4.predict failed; running it launch by launch')
                    self._graph_runs[key] = -10 ** 9
        return self._predict_eager(P, x, logits)

    def _predict_eager(self, P, x, logits):
        L = _lib.lib()
        dev = P['dev']
        x = x.float().contiguous()
        N, H, Wd = 1, x.shape[2], x.shape[3]
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        h, w = H // 2, Wd // 2
        cur = f(N, h, w, 64)
        check(L.vfn_ln_stem_f32(ptr(x), ptr(P['stem_w']), ptr(P['stem_sc']), ptr(P['stem_sh']), ptr(cur), N, H, Wd, h, w, 64,
                                _same_pad_before(3, 2), stream()), 'vfn_ln_stem_f32')
        feats = [cur]
        NB = 256
        for i, q in enumerate(P['blocks']):
            inp = cur
            if q['e'] != 1:
                ex = f(N, h, w, q['oup_p'])
                self._conv(P, cur, q['exp_w'], q['oup_p'], 1, 0, ex, q['exp_sc'], q['exp_sh'], N, h, w)
            else:
                ex = cur
            ho, wo = h // q['s'], w // q['s']
            dwo = f(N, ho, wo, q['oup_p'])
            check(L.vfn_ln_dwconv_f32(ptr(ex), ptr(q['dw_w']), ptr(q['dw_sc']), ptr(q['dw_sh']), ptr(dwo), N, h, w, q['oup_p'],
                                      q['oup_p'], q['oup_p'], q['k'], q['s'], q['pad_b'], ho, wo, int(q['e'] != 1), stream()),
                  'vfn_ln_dwconv_f32')
            M = N * ho * wo
            sums, part, gate = f(q['oup_p']), f(NB * q['oup_p']), f(q['oup_p'])
            check(L.vfn_colsum_f32(ptr(dwo), M, q['oup_p'], q['oup_p'], ptr(part), NB, ptr(sums), stream()), 'vfn_colsum_f32')
            check(L.vfn_ln_se_gate_f32(ptr(sums), 1.0 / M, ptr(q['se_w1']), ptr(q['se_b1']), ptr(q['se_w2']), ptr(q['se_b2']), ptr(gate),
                                       q['oup'], q['sq'], q['oup_p'], stream()), 'vfn_ln_se_gate_f32')
            # se_w1 is [sq][oup] over the UNPADDED channels while the sums are padded: gather is by index < oup, same layout
            wg = torch.empty_like(q['proj_w'])
            check(L.vfn_ln_scale_cols_f32(ptr(q['proj_w']), ptr(gate), ptr(wg), q['proj_w'].shape[0], q['oup_p'], stream()),
                  'vfn_ln_scale_cols_f32')
            out = f(N, ho, wo, q['cout_p'])
            skip = inp if (q['s'] == 1 and q['cin'] == q['cout']) else None
            self._conv(P, dwo, wg, q['cout_p'], 1, 0, out, q['proj_sc'], q['proj_sh'], N, ho, wo, res=skip)
            cur, h, w = out, ho, wo
            if i + 1 in STAGE_IDXS:
                feats.append(cur)
        feats = feats[::-1]                                   # deepest first: 448, 160, 56, 32, 48 channels
        xd = feats[0]
        for j, q in enumerate(P['dec']):
            a = f(N, h, w, q['mid_p'])
            self._conv(P, xd, q['a_w'], q['mid_p'], 1, 0, a, q['a_sc'], q['a_sh'], N, h, w, relu_out=True)
            z = f(N, 2 * h, 2 * w, q['mid_p'])
            check(L.vfn_dilate2_f32(ptr(a), ptr(z), N, h, w, 2 * h, 2 * w, q['mid_p'], stream()), 'vfn_dilate2_f32')
            t = f(N, 2 * h, 2 * w, q['mid_p'])
            self._conv(P, z, q['t_w'], q['mid_p'], 4, 2, t, q['t_sc'], q['t_sh'], N, 2 * h, 2 * w, relu_out=True, Ho=2 * h, Wo=2 * w)
            h, w = 2 * h, 2 * w
            c = f(N, h, w, q['cout_p'])
            self._conv(P, t, q['c_w'], q['cout_p'], 1, 0, c, q['c_sc'], q['c_sh'], N, h, w, relu_out=True)
            if j + 1 < len(feats):
                check(L.vfn_ln_add_f32(ptr(c), ptr(feats[j + 1]), ptr(c), c.numel(), stream()), 'vfn_ln_add_f32')
            xd = c
        out = f(N, 1, h, w)
        check(L.vfn_ln_head_f32(ptr(xd), ptr(P['head_w']), P['head_b'], ptr(out), N * h * w, 32, xd.shape[-1], int(not logits), stream()),
              'vfn_ln_head_f32')
        return out

    def forward(self, x):
        return self.predict(x)
