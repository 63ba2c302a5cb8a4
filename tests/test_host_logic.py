"""Host-side logic that needs no GPU: padding, resize rule, tiling choices, weight repacking,
dataset / palette plumbing, the host connected-components routine."""
import os

import numpy as np
import pytest
import torch
from torch.nn import functional as F


def test_pad_divide_by_matches_oracle():
    from vfloodnet_amd.engine import pad_divide_by
    from oracle import afb_urr_ref as O
    for h, w in [(480, 854), (480, 853), (1080, 1920), (96, 160), (90, 150), (481, 17)]:
        pad, nh, nw = pad_divide_by(h, w)
        (x,), pad_ref = O.pad_divide_by([torch.zeros(1, 1, h, w)], 16, (h, w))
        assert pad == tuple(pad_ref) and (nh, nw) == tuple(x.shape[-2:])
    assert pad_divide_by(480, 854)[0] == (5, 5, 0, 0)


def test_resized_hw_rule():
    from vfloodnet_amd.video_seg import resized_hw
    assert resized_hw(480, 854, 480) == (480, 854)
    assert resized_hw(720, 1280, 480) == (480, 853)
    assert resized_hw(1080, 1920, 480) == (480, 853)
    assert resized_hw(1920, 1080, 480) == (853, 480)
    assert resized_hw(64, 96, 128) == (128, 192)


def test_conv_weight_packing_is_a_gemm():
    """packed[c, (kh,kw,cin)] . im2col(NHWC) == conv2d."""
    from vfloodnet_amd import weights as W
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 8, 6, 7, generator=g)
    w = torch.randn(5, 8, 3, 3, generator=g)
    ref = F.conv2d(x, w, padding=1)
    wp = W.pack_conv_weight(w)                                   # [5, 72]
    xp = F.pad(x, (1, 1, 1, 1)).permute(0, 2, 3, 1)              # NHWC
    cols = torch.stack([xp[0, i:i + 3, j:j + 3, :].reshape(-1) for i in range(6) for j in range(7)])
    out = (cols @ wp.t()).t().reshape(1, 5, 6, 7)
    assert (out - ref).abs().max() < 1e-5


def test_bn_scale_shift_equals_eval_batchnorm():
    from vfloodnet_amd import weights as W
    bn = torch.nn.BatchNorm2d(6).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
    x = torch.randn(2, 6, 4, 4)
    sc, sh = W.bn_scale_shift(bn)
    assert (bn(x) - (x * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))).abs().max() < 1e-5


def test_state_dict_names_match_reference_schema():
    """562 entries with the reference's names (checkpoint schema)."""
    import vfloodnet_amd
    m = vfloodnet_amd.AFB_URR('cpu', update_bank=True, _allow_cpu_container=True)
    sd = m.state_dict()
    assert len(sd) == 562
    n_param = sum(v.numel() for k, v in sd.items() if v.dtype.is_floating_point and 'running' not in k and not k.endswith(('.mean', '.std')))
    assert n_param == 33082660            # 33,082,660 parameters
    for k in ['encoder_m.conv1_m.weight', 'encoder_m.res4.5.bn3.running_var', 'encoder_q.res3.0.downsample.1.weight',
              'keyval_r4.Key.bias', 'decoder.RF2.ResFS.conv2.weight', 'decoder.local_pred2.bias', 'encoder_q.mean']:
        assert k in sd
    golden = os.path.join(os.path.dirname(__file__), 'golden', 'state_dict_names.txt')
    if os.path.exists(golden):
        names = [l.split()[0] for l in open(golden).read().splitlines() if l.strip()]
        assert names == list(sd.keys())


def test_nsplit_and_cfg_choices():
    from vfloodnet_amd.feature_bank import pick_nsplit
    from vfloodnet_amd import engine
    assert pick_nsplit(1620, 2, 100) == 1                   # two 64-entry chunks are not worth splitting
    s = pick_nsplit(1620, 2, 100000)
    assert 1 <= s <= 20 and (26 * 2 * s) % 256 <= 256
    engine._CFG_TILES = [(128, 128), (128, 64), (64, 128), (64, 64), (32, 64), (64, 32), (128, 32), (256, 128),
                         (128, 128), (128, 128), (64, 128), (128, 128), (64, 128), (64, 64), (256, 128), (128, 32), (64, 128), (128, 256), (128, 256), (64, 256)]
    engine._TUNED.clear()                                  # exercise the heuristic, not the measured table
    c, ks, sf = engine.choose_cfg(51840, 256, 2304)
    assert engine._CFG_TILES[c][1] >= 64 and ks == 1
    c, ks, sf = engine.choose_cfg(207360, 2, 288)
    assert engine._CFG_TILES[c][1] == 32 and ks == 1
    c, ks, sf = engine.choose_cfg(1620, 256, 2304)          # 1/16-resolution layer: too few tiles -> split K
    assert ks > 1 and sf == 0 and ((72 + ks - 1) // ks) * (ks - 1) < 72
    engine._load_tuned()


def test_persistent_choices_fall_back_for_descriptors_they_do_not_take():
    """Round 5: a shape's measured choice may be the persistent 1x1 kernel (configuration ids >= 2000), which does not implement the
    backward passes' ReLU masks, operand images or taps -- and the table is keyed by shape alone.  ``apply_choice`` must hand such a
    descriptor the shape's second entry (or the heuristic), never the persistent id; every persistent 1x1 entry of the shipped table
    carries that second entry; the ids' encoding round-trips."""
    import json
    from vfloodnet_amd import engine, ops
    from vfloodnet_amd._lib import ConvDesc
    d = ConvDesc()
    d.KH = d.KW = d.stride = 1
    d.pad, d.Cin, d.in_ld, d.Cout, d.cout_pad, d.out_ld, d.M, d.N, d.H, d.W = 0, 256, 256, 64, 256, 64, 51840, 2, 120, 216
    assert ops.pconv_eligible(d, 0) and not ops.pconv_eligible(d, 1)
    pc = ops.pconv_cfg(5, 512)
    assert pc == 2000 + 5 + 8 * 4 and ops.conv_cfg_name(pc) == 'wino_gemm_kernel<64, 128, 2, 4, 2, true, false>'
    assert ops.conv_cfg_name(ops.wino_gemm_cfg(2, 768), 1) == 'wino_gemm_kernel<128, 64, 4, 2, 1, false, true>'
    assert engine.apply_choice(d, (pc, 1, 0, 10, 1, 0), None) == pc             # eligible: the persistent kernel
    d.mask, d.mask_ld = 4096, 256                                              # a masked data gradient of the same shape
    assert not ops.pconv_eligible(d, 0)
    assert engine.apply_choice(d, (pc, 1, 0, 10, 1, 0), None) == 10            # the shape's fallback entry
    assert engine.apply_choice(d, (pc, 1, 0), None) < ops.WINO_GEMM_CFG0       # none recorded: the heuristic
    d.mask, d.KH, d.KW, d.pad = None, 3, 3, 1
    assert not ops.pconv_eligible(d, 0)
    table = json.load(open(engine._TUNED_PATH))
    n = 0
    for k, v in table.items():
        if v[0] >= ops.PCONV_CFG0:
            n += 1
            assert len(v) == 6 and v[3] < ops.WINO_GEMM_CFG0, (k, v)
        elif v[0] >= ops.WINO_GEMM_CFG0:
            assert int(k.split(',')[0]) % 36 == 0, (k, v)                       # transform-domain GEMM shapes only (36 components)
    assert n >= 20, n
    for k, v in json.load(open(engine._TUNED_BF16_PATH)).items():
        assert v[0] < ops.PCONV_CFG0                                           # (the bf16 table: persistent ids for the Winograd GEMMs only)


def test_video_ds_and_palette(tmp_path):
    from PIL import Image
    from vfloodnet_amd.dataset import Video_DS, to_onehot
    from vfloodnet_amd.data import save_seg_mask, load_image_in_PIL, color_palette
    rng = np.random.RandomState(0)
    paths = []
    for i in range(3):
        p = str(tmp_path / f'{i:05d}.png')
        Image.fromarray(rng.randint(0, 255, (20, 30, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    mask = np.zeros((20, 30), np.uint8)
    mask[5:15, 8:20] = 1
    mp = str(tmp_path / 'm.png')
    save_seg_mask(mask, mp, color_palette)
    im = Image.open(mp)
    assert im.mode == 'P' and im.getpalette()[:12] == [0, 0, 0, 0, 0, 128, 0, 128, 0, 128, 0, 0]
    ds = Video_DS(paths, load_image_in_PIL(paths[0]), load_image_in_PIL(mp, 'P'))
    assert ds.obj_n == 2 and len(ds) == 2
    assert ds.first_mask.shape == (2, 20, 30) and ds.first_mask.dtype == torch.uint8
    assert torch.equal(ds.first_mask[1], torch.from_numpy(mask)) and torch.equal(ds.first_mask[0], 1 - ds.first_mask[1])
    f, name = ds[0]
    assert f.shape == (3, 20, 30) and f.dtype == torch.float32 and name == '00001' and 0 <= f.min() and f.max() <= 1
    oh, objs = to_onehot(np.array([[0, 2], [1, 2]]), 3)
    assert objs == [1, 2] and oh[0].tolist() == [[1, 0], [0, 0]]


def test_postprocess_cases():
    """vfn_postprocess_pred_u8 is a host routine of the HIP library (the reference runs cv2 on the CPU)."""
    from vfloodnet_amd.data import postprocessing_pred
    from oracle import afb_urr_ref as O
    a = np.zeros((20, 30), np.uint8)
    assert postprocessing_pred(a).min() == 1                     # all background -> all ones (reference quirk)
    assert postprocessing_pred(np.ones((20, 30), np.uint8)).min() == 1
    a[2:6, 3:9] = 1
    a[10:18, 12:28] = 1
    a[7, 9] = 1
    assert np.array_equal(postprocessing_pred(a), O.postprocessing_pred(a))
    b = np.zeros((9, 9), np.uint8)
    b[1:4, 1:4] = 1
    assert np.array_equal(postprocessing_pred(b), b)             # single blob: identity
    rng = np.random.RandomState(0)
    for thr in (0.45, 0.55, 0.7):
        c = (rng.rand(64, 80) > thr).astype(np.uint8)
        assert np.array_equal(postprocessing_pred(c), O.postprocessing_pred(c))


def test_overlay_matches_reference_outputs():
    """data.add_overlay == the REFERENCE's myutils.add_overlay (outputs generated by oracle/gen_image_seg_golden.py):
    two / three labels, no background label, a single label, a gap in the label ids."""
    import os
    from vfloodnet_amd.data import add_overlay, color_palette
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'overlay_cases.npz'))
    img = (np.transpose(g['frame'], (1, 2, 0)) * 255).astype(np.uint8)
    bgr = np.ascontiguousarray(img[..., ::-1])
    cases = [k[5:] for k in g.files if k.startswith('mask_')]
    assert len(cases) == 5
    for n in cases:
        assert np.array_equal(add_overlay(bgr, g['mask_' + n], color_palette), g['bgr_out_' + n]), n


def test_bench_and_entry_modules_import_on_cpu():
    """bench.py / __graft_entry__.py only touch the GPU inside main() / smoke(): importing them here catches syntax
    and contract regressions (the driver's JSON fields) without a device."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    bench = importlib.import_module('bench')
    assert set(bench.WORKLOADS) == {'C2', 'C3', 'C5'} and bench.WORKLOADS['C2'] == (480, 854, 1)
    assert bench.PEAKS['fp32'] == 157.3 and abs(bench.PEAKS['bf16x3'] * 3 - bench.PEAKS['bf16']) < 1e-9
    entry = importlib.import_module('__graft_entry__')
    assert callable(entry.build) and callable(entry.smoke)


def test_video_ds_device_decode_sniffs_content_and_defers_bad_files_to_pil(tmp_path):
    """The reference opens every frame with PIL, which looks at the bytes (Water_DS.py:105-109): a PNG saved as .jpg (or the
    reverse) works there and must work on the default decode path; a file the host-side decoders reject goes to PIL,
    which either tolerates it or raises ITS error."""
    import numpy as np
    from PIL import Image
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd.dataset import Video_DS
    rng = np.random.RandomState(5)
    a = rng.randint(0, 256, (24, 40, 3)).astype(np.uint8)
    first = tmp_path / '00000.jpg'
    Image.fromarray(a).save(first)
    png_as_jpg, jpg_as_png, cut = tmp_path / '00001.jpg', tmp_path / '00002.png', tmp_path / '00003.jpg'
    Image.fromarray(a).save(png_as_jpg, format='PNG')
    Image.fromarray(a).save(jpg_as_png, format='JPEG', quality=90)
    import io
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format='JPEG', quality=90)
    cut.write_bytes(buf.getvalue()[:120])                                    # truncated inside the tables
    ds = Video_DS([str(first), str(png_as_jpg), str(jpg_as_png), str(cut)], Image.fromarray(a),
                  Image.fromarray((a[:, :, 0] > 100).astype(np.uint8)), decode='device')
    item, name = ds[0]
    assert name == '00001' and set(item) == {'png'}
    item, _ = ds[1]
    assert set(item) == {'jpeg'}
    import pytest
    with pytest.raises(OSError):                                             # PIL's own verdict on the truncated file
        ds[2]


def test_hard_task_generators_are_deterministic_and_colour_free():
    """tools/synth.frame0_hard / noisy_labels / hard_step (round 6: the task whose margins do not saturate): reproducible bit for bit,
    water and land share their colour statistics (no tint cue beyond 0.04 in any channel), ~10 % of the noisy labels disagree with the
    image, and enlarged clips move by steps that are whole pixels at the network's resolution."""
    import torch
    from tools import synth
    a, ma = synth.frame0_hard(7, 200, 320)
    b, mb = synth.frame0_hard(7, 200, 320)
    assert torch.equal(a, b) and torch.equal(ma, mb)
    w = ma.bool()
    assert 0.3 < float(ma.float().mean()) < 0.7
    # (a single small frame carries a random offset from its low-frequency field; the SYSTEMATIC water - land difference is the 0.02-0.03 tint)
    diffs = []
    for sd_ in range(6):
        x_, m_ = synth.frame0_hard(100 + sd_, 200, 320)
        diffs.append([float(x_[c][m_.bool()].mean()) - float(x_[c][~m_.bool()].mean()) for c in range(3)])
    for c in range(3):
        assert abs(sum(d_[c] for d_ in diffs) / len(diffs)) < 0.05, diffs          # (frame0: 0.05-0.20 per channel, every frame)
    assert max(abs(float(x[c][wx].mean()) - float(x[c][~wx].mean())) for x, wx in [(synth.frame0(7, 200, 320)[0], synth.frame0(7, 200, 320)[1].bool())]
               for c in range(3)) > 0.1                                             # ... which the tinted task has
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    n1, n2 = synth.noisy_labels(ma.long(), g1), synth.noisy_labels(ma.long(), g2)
    assert torch.equal(n1, n2)
    fr = [float((synth.noisy_labels(ma.long(), g1) != ma.long()).float().mean()) for _ in range(8)]
    assert 0.03 < sum(fr) / len(fr) < 0.25, fr
    assert synth.hard_step(480, 854) == (2, 5) and synth.hard_step(720, 1280) == (3, 6) and synth.hard_step(1080, 1920) == (9, 18)
    f, m = synth.clip_hard(3, 3, 720, 1280)
    assert f.shape == (3, 3, 720, 1280) and m.shape == (720, 1280)
    assert torch.equal(f[1], torch.roll(f[0], (3, 6), (1, 2)))
    f2, _ = synth.clip_hard(3, 3, 720, 1280, in_place=True)
    assert torch.equal(f, f2)


def test_bank_snapshot_is_validated_before_any_device_work():
    """FeatureBank.state_dict / load_state_dict (round 6, SURVEY section 5 "bank snapshot for long streams"): an empty bank has no
    snapshot; a snapshot of another version, object count or with entries that do not match their recorded lengths is refused by
    the host checks -- before the loader allocates on the device (which this CPU container does not have)."""
    from vfloodnet_amd.feature_bank import FeatureBank, DK, DV
    fb = FeatureBank(2, 1000, 'cpu')
    with pytest.raises(RuntimeError, match='empty'):
        fb.state_dict()
    good = dict(version=1, obj_n=2, hw=60, dk=DK, dv=DV, class_budget=400.0, update_rate=0.1, thres_close=0.95, precision='fp32',
                lens=[3, 2], n_updates=1, hist={0: [3, 2]}, peak_n=[3.0, 2.0], replace_n=[0.0, 0.0],
                keys=[torch.zeros(3, DK), torch.zeros(2, DK)], values=[torch.zeros(3, DV), torch.zeros(2, DV)],
                info=[torch.zeros(3, 2), torch.zeros(2, 2)])
    for edit in (dict(version=2), dict(obj_n=3), dict(dk=64), dict(lens=[3, 3]), dict(info=[torch.zeros(3, 2), torch.zeros(2, 3)])):
        with pytest.raises(ValueError):
            fb.load_state_dict({**good, **edit})
    with pytest.raises(RuntimeError, match='GPU'):               # a valid snapshot gets as far as the device allocation
        fb.load_state_dict(good)


def test_group_partition_ends_on_memorised_frames():
    """ClipRunner.group_len (round 6: the frames between two memorize calls as one batched pass): groups end on the frames the loop
    memorises (t % mem_every == 0), the first group of a resumed or unaligned stream is shorter, the last one takes what is left."""
    import types
    from vfloodnet_amd.video_seg import ClipRunner
    r = ClipRunner(types.SimpleNamespace(device=torch.device('cpu'), precision='fp32'), 2, 1000, mem_every=5)
    got, t, T = [], 0, 23
    while t < T:
        r.t = t
        g = r.group_len(T - t)
        got.append(g)
        assert all((t + 1 + i) % 5 != 0 for i in range(g - 1))          # only the last frame of a group may be memorised
        t += g
    assert got == [5, 5, 5, 5, 3]
    r.t = 7                                                            # e.g. resumed from a snapshot after frame 7
    assert r.group_len(100) == 3 and r.group_len(2) == 2 and r.group_len(0) == 0
    r1 = ClipRunner(types.SimpleNamespace(device=torch.device('cpu'), precision='fp32'), 2, 1000, mem_every=1)
    r1.t = 11
    assert r1.group_len(4) == 1
