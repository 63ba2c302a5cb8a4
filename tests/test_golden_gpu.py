"""The HIP path against the reference's golden vectors directly (not through the oracle)."""
import numpy as np
import pytest
import torch

from golden_util import load, state_dict, t, close_logits, miou

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def model(gpu):
    from vfloodnet_amd import AFB_URR
    m = AFB_URR(gpu, update_bank=True).to(gpu).eval()
    m.load_state_dict(state_dict(), strict=True)
    return m


@pytest.mark.parametrize('wino', ['table', 'all'])
@pytest.mark.parametrize('tag', ['96x160', '90x150'])
def test_blocks(gpu, model, tag, wino, monkeypatch):
    """``wino='all'``: every eligible 3x3 / stride-1 layer (KeyValue, the bottlenecks' conv2, the whole decoder but the heads) runs as
    Winograd F(4x4, 3x3) (VFN_WINOGRAD=2) -- at these frame sizes the measured table keeps them all direct, so this is where the
    transform-domain path meets the reference's block goldens at the same 5e-4 / 1e-3."""
    from vfloodnet_amd import FeatureBank, ops
    if wino == 'all':
        from vfloodnet_amd import AFB_URR, engine as E
        monkeypatch.setattr(E, '_WINOGRAD', '2')
        model = AFB_URR(gpu, update_bank=True).to(gpu).eval()
        model.load_state_dict(state_dict(), strict=True)
    g = load(f'blocks_{tag}.npz')
    frames, m0 = t(g['frames']).to(gpu), t(g['mask'])
    H, W = frames.shape[-2:]
    oh = torch.stack([1 - m0, m0], 0).unsqueeze(0).to(gpu)
    k, v = model.memorize(frames[0:1], oh)
    assert (torch.stack([x.cpu() for x in k]) - t(g['key0'])).abs().max() < 5e-4
    assert (torch.stack([x.cpu() for x in v]) - t(g['val0'])).abs().max() < 5e-4
    fb = FeatureBank(2, 250000, gpu)
    fb.init_bank(k, v)
    score, _ = model.segment(frames[1:2], fb)
    ok, dl, dp = close_logits(score.cpu(), t(g['score']), 1e-3)
    assert ok, (dl, dp)
    p, qs, slot = model.engine().last_query                 # the query set / slot this frame's features live in
    n_wino = sum(1 for lst in p.all_lists() for l in lst if 'wino_gemm' in (l.name or ''))
    if wino == 'all':
        assert n_wino >= 30, n_wino                          # (KeyValue x2, the bottlenecks, the decoder; query-side lists exist per batch size)
    else:
        assert n_wino == 0, n_wino                           # (below the table's / heuristic's size threshold: all direct)
    for n, buf in dict(r1=qs.q['r1'], r2=qs.q['res2']['out'], r3=qs.q['res3']['out'], r4=qs.q['res4']['out']).items():
        x = buf[slot:slot + 1].permute(0, 3, 1, 2).contiguous().cpu()
        assert (x.flatten()[t(g[n + '_idx'])] - t(g[n + '_val'])).abs().max() < 5e-4, n
    # info[:,1] = log(hit count + 1) (AFB_URR.py:161-174): the counts are integers, so an entry either agrees to the
    # rounding of log() or one probability sat on the 1e-3 threshold and its count moved by one -- at most a couple of
    # entries may do that, every other one must agree to 1e-3 (measured: identical)
    got, ref = fb.info[0].cpu(), t(g['info0'])
    d = (got - ref).abs().max(dim=1).values
    flipped = d >= 1e-3
    assert int(flipped.sum()) <= 2, int(flipped.sum())
    dc = (torch.exp(got[flipped, 1]) - torch.exp(ref[flipped, 1])).abs()
    assert bool((dc < 1.5).all()), dc
    assert torch.equal(got[:, 0], ref[:, 0])
    pm = ops.softmax_objects(score)
    k2, v2 = model.memorize(frames[1:2], pm)
    assert (torch.stack([x.cpu() for x in k2]) - t(g['key1'])).abs().max() < 2e-3
    ref_lab = t(g['score'])[0].argmax(0)
    margin = (t(g['score'])[0, 1] - t(g['score'])[0, 0]).abs()
    assert torch.equal(score.cpu()[0].argmax(0)[margin > 1e-2], ref_lab[margin > 1e-2])


@pytest.mark.parametrize('regime', ['append', 'merge', 'mixed', 'evict'])
def test_bank_update(gpu, regime):
    from vfloodnet_amd import FeatureBank
    g = load(f'bank_{regime}.npz')
    k0, v0 = t(g['k0']).to(gpu), t(g['v0']).to(gpu)
    fb = FeatureBank(2, int(g['budget']), gpu, 0.1, 0.95)
    fb.init_bank([k0[i] for i in range(2)], [v0[i] for i in range(2)])
    for step in range(1, 5):
        for i in range(2):
            fb.info[i][:, 1] += t(g[f'bump_{step}_{i}']).to(gpu)
        k1, v1 = t(g[f'k1_{step}']).to(gpu), t(g[f'v1_{step}']).to(gpu)
        fb.update([k1[i] for i in range(2)], [v1[i] for i in range(2)], step)
        for i in range(2):
            assert tuple(fb.info[i].shape) == g[f'info_{step}_{i}'].shape, (regime, step)
            assert (fb.info[i].cpu() - t(g[f'info_{step}_{i}'])).abs().max() < 1e-5
    for i in range(2):
        assert (fb.keys[i].cpu() - t(g[f'keys_4_{i}'])).abs().max() < 1e-4
        assert (fb.values[i].cpu() - t(g[f'values_4_{i}'])).abs().max() < 1e-4
    assert np.array_equal(fb.peak_n, g['peak_n']) and np.array_equal(fb.replace_n, g['replace_n'])


def test_main_loop_pngs(gpu, tmp_path, monkeypatch):
    """vfloodnet_amd.video_seg.main on PNG frames == the PNGs the reference's main() wrote."""
    import argparse
    from PIL import Image
    from vfloodnet_amd import video_seg
    from tools import synth
    from vfloodnet_amd.data import save_seg_mask, color_palette
    g = load('main_loop_120x200.npz')
    H, W = [int(x) for x in g['shape']]
    labels = np.unpackbits(g['labels'], axis=-1)[..., :W]
    fdir = tmp_path / 'frames'
    fdir.mkdir()
    for i, fr in enumerate(g['frames_u8']):
        Image.fromarray(fr.transpose(1, 2, 0)).save(str(fdir / f'{i:05d}.png'))
    ckpt = str(tmp_path / 'ckpt.pth')
    torch.save({'epoch': 0, 'model': state_dict(), 'loss': 0.0, 'seed': 20200212}, ckpt)
    monkeypatch.chdir(tmp_path)
    (tmp_path / 'output' / 'segs' / 'clip' / 'mask').mkdir(parents=True)
    save_seg_mask(g['mask'], str(tmp_path / 'output' / 'segs' / 'clip' / 'mask' / '00000.png'), color_palette)
    args = argparse.Namespace(gpu=0, budget=250000, viz=True, model_path=ckpt, update_rate=0.1, merge_thres=0.95,
                              test_path=str(fdir), test_name='clip')
    video_seg.main(args, gpu)
    for i in range(len(labels)):
        im = Image.open(str(tmp_path / 'output' / 'segs' / 'clip' / 'mask' / f'{i:05d}.png'))
        assert im.mode == 'P' and im.getpalette()[:12] == [int(x) for x in g['palette'][:12]]
        assert miou(torch.from_numpy(np.array(im)), torch.from_numpy(labels[i])) > 0.995, i
    ov = np.array(Image.open(str(tmp_path / 'output' / 'segs' / 'clip' / 'overlay' / '00001.png')))
    assert ov.shape == g['overlay1'].shape
    assert (np.abs(ov.astype(int) - g['overlay1'].astype(int)) > 1).mean() < 0.01


def test_full_size_samples(gpu, model):
    from vfloodnet_amd import FeatureBank
    from tools import synth
    g = load('full_480x854.npz')
    frames, m0 = synth.clip(1, 2, 480, 854)
    oh = synth.onehot(m0).unsqueeze(0)
    k, v = model.memorize(frames[0:1].to(gpu), oh.to(gpu))
    for i in range(2):
        assert (k[i].cpu().flatten()[t(g['key_idx'])] - t(g['key_val'][i])).abs().max() < 1e-3
        assert (v[i].cpu().flatten()[t(g['val_idx'])] - t(g['val_val'][i])).abs().max() < 1e-3
    fb = FeatureBank(2, 250000, gpu)
    fb.init_bank(k, v)
    score, _ = model.segment(frames[1:2].to(gpu), fb)
    ok, dl, dp = close_logits(score.cpu().flatten()[t(g['score_idx'])], t(g['score_val']), 2e-3)
    assert ok, (dl, dp)
    assert abs(float((score[0, 1] > score[0, 0]).float().mean()) - float(g['label_water_frac'])) < 2e-3


def test_c2_full_clip_vs_reference(gpu, model):
    """BASELINE config C2: the whole 100-frame 480x854 fp32 clip, free-running (segment -> softmax -> memorize ->
    bank update every frame, eviction at the budget), against the label maps produced by the reference's own
    model + FeatureBank on CPU (oracle/gen_c2_golden.py).  Target: mIoU >= 0.99 on every frame."""
    import os
    from golden_util import GOLDEN
    from tools import synth
    from vfloodnet_amd.video_seg import run_clip
    path = os.path.join(GOLDEN, 'c2_480x854_100.npz')
    if not os.path.exists(path):
        pytest.skip('c2 golden not generated')
    g = np.load(path)
    H, W = [int(x) for x in g['shape']]
    ref = np.unpackbits(g['labels'], axis=-1)[..., :W]
    T = ref.shape[0]
    frames, m0 = synth.clip(int(g['seed']), T, H, W)
    out = run_clip(model, frames.to(gpu), m0)
    lab = out['labels'].numpy()
    assert np.array_equal(lab[0], ref[0])
    ious = [miou(torch.from_numpy(lab[t]), torch.from_numpy(ref[t])) for t in range(1, T)]
    sizes = np.array(out['bank_sizes'])
    drift = np.abs(sizes - g['bank_sizes']).max()
    print('C2 mIoU min %.5f mean %.5f; bank size max drift %d; peak %s replace %s (ref %s %s)' % (
        min(ious), sum(ious) / len(ious), drift, out['fb'].peak_n, out['fb'].replace_n, g['peak_n'], g['replace_n']))
    assert min(ious) >= 0.99, (min(ious), int(np.argmin(ious)) + 1)
    assert drift <= max(8, 0.001 * sizes.max())        # merge/append decisions sit on a float threshold


def test_c2_clip_bf16x3_against_reference_labels(gpu):
    import os
    """The reduced-precision configuration this package recommends (precision='bf16x3': every matrix operand split
    into two bf16, three bf16 MFMAs per product, f32 accumulation) on the full C2 clip against the labels of the
    reference's own f32 CPU run.  Tolerance: mIoU >= 0.985 on every frame and >= 0.995 on average (measured: min
    0.994-0.996, mean 0.998-0.999 depending on the split-K choices of the tuned table; the exact-f32 path gives min
    0.998).  Plain 'bf16' is not asserted: with these random synthetic weights (margin-free logits)
    it reaches mIoU ~0.79 -- see DESIGN.md."""
    import numpy as np
    from golden_util import GOLDEN
    from vfloodnet_amd import AFB_URR
    from tools import synth
    from vfloodnet_amd.video_seg import run_clip
    path = os.path.join(GOLDEN, 'c2_480x854_100.npz')
    if not os.path.exists(path):
        pytest.skip('c2 golden not generated')
    g = np.load(path)
    H, W = [int(x) for x in g['shape']]
    ref = np.unpackbits(g['labels'], axis=-1)[..., :W]
    T = ref.shape[0]
    model = AFB_URR(gpu, update_bank=True, precision='bf16x3').to(gpu).eval()
    model.load_state_dict(synth.make_state_dict(20200212))
    frames, m0 = synth.clip(int(g['seed']), T, H, W)
    out = run_clip(model, frames.to(gpu), m0)
    lab = out['labels'].numpy()
    ious = [miou(torch.from_numpy(lab[t]), torch.from_numpy(ref[t])) for t in range(1, T)]
    print('C2 bf16x3 mIoU min %.5f mean %.5f' % (min(ious), sum(ious) / len(ious)))
    assert min(ious) >= 0.985, (min(ious), int(np.argmin(ious)) + 1)
    assert sum(ious) / len(ious) >= 0.995


@pytest.mark.parametrize('tag,H,W,training', [('eval_90x150', 90, 150, False), ('train_96x160', 96, 160, True)])
def test_segment_batch_and_training_branch(gpu, tag, H, W, training):
    """SURVEY.md 8(f) row 4, first step: ``segment`` with a batch of frames (train_video_seg.py:69 passes bs = clip_n - 1)
    in eval mode and in the training branch (no padding AFB_URR.py:278, scalar uncertainty :302-305, BatchNorm frozen as
    train_video_seg.py:103-106), forward only, against the reference's own outputs (oracle/gen_bs2_golden.py)."""
    from vfloodnet_amd import AFB_URR, FeatureBank
    from tools import synth
    g = load('segment_bs2.npz')
    model = AFB_URR(gpu, update_bank=not training).to(gpu)
    model.load_state_dict(state_dict(), strict=True)
    model.train() if training else model.eval()
    frames, m0 = synth.clip(6, 3, H, W)
    oh = synth.onehot(m0).unsqueeze(0)
    k, v = model.memorize(frames[0:1].to(gpu), oh.to(gpu))
    fb = FeatureBank(2, 250000, gpu)
    fb.init_bank(k, v)
    score, unc = model.segment(frames[1:3].to(gpu), fb)
    assert tuple(score.shape) == (2, 2, H, W)
    ok, dl, dp = close_logits(score.detach().cpu(), t(g[f'{tag}_score']), 1e-3)      # (training mode with autograd on: a graph node)
    assert ok, (dl, dp)
    for i in range(2):
        assert (fb.info[i][:, 1].cpu() - t(g[f'{tag}_info1'][i])).abs().max() < 1e-3      # hit counts: sample 0 only
    if training:
        assert unc.dim() == 0 and abs(float(unc.detach()) - float(g[f'{tag}_uncertainty'])) < 1e-5
        with pytest.raises(RuntimeError):                       # no padding in this branch (AFB_URR.py:278)
            model.segment(frames[1:3, :, :90, :150].contiguous().to(gpu), fb)
    else:
        assert unc is None
        # a batch is its samples one by one (hit counts apart)
        fb2 = FeatureBank(2, 250000, gpu)
        fb2.init_bank(k, v)
        one = model.segment(frames[2:3].to(gpu), fb2)[0].clone()
        assert torch.equal(one[0], score[1])
