"""The CPU oracle (oracle/afb_urr_ref.py) against golden vectors produced by the reference itself
(oracle/gen_golden.py: /root/reference imported under stubs).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from golden_util import load, meta, state_dict, t, close_logits, checksum, miou
from oracle import afb_urr_ref as O


@pytest.fixture(scope='module')
def sd():
    torch.set_num_threads(8)
    return state_dict()


@pytest.mark.parametrize('tag', ['96x160', '90x150'])
def test_blocks(sd, tag):
    g = load(f'blocks_{tag}.npz')
    frames, m0 = t(g['frames']), t(g['mask'])
    H, W = frames.shape[-2:]
    oh = torch.stack([1 - m0, m0], 0).unsqueeze(0)
    k, v = O.memorize(sd, frames[0:1], oh)
    assert (torch.stack(k) - t(g['key0'])).abs().max() < 2e-4
    assert (torch.stack(v) - t(g['val0'])).abs().max() < 2e-4
    [fp], pad = O.pad_divide_by([frames[1:2]], 16, (H, W))
    assert list(pad) == list(g['pad'])
    r4, r3, r2, r1 = O.encoder_q(sd, fp)
    for n, x in dict(r1=r1, r2=r2, r3=r3, r4=r4).items():
        assert (x.flatten()[t(g[n + '_idx'])] - t(g[n + '_val'])).abs().max() < 2e-4, n
        assert np.allclose(checksum(x), g[n + '_sum'], rtol=1e-4), n
    fb = O.FeatureBankRef(2, 250000)
    fb.init_bank(k, v)
    score, unc = O.segment(sd, frames[1:2], fb)
    ok, dl, dp = close_logits(score, t(g['score']), 1e-3)
    assert ok and unc is None, (dl, dp)
    assert (fb.info[0] - t(g['info0'])).abs().max() < 1e-3 and (fb.info[1] - t(g['info1'])).abs().max() < 1e-3
    k2, v2 = O.memorize(sd, frames[1:2], torch.softmax(score, dim=1))
    assert (torch.stack(k2) - t(g['key1'])).abs().max() < 2e-4
    assert (torch.stack(v2) - t(g['val1'])).abs().max() < 2e-4


@pytest.mark.parametrize('regime', ['append', 'merge', 'mixed', 'evict'])
def test_bank_update(regime):
    g = load(f'bank_{regime}.npz')
    k0, v0 = t(g['k0']), t(g['v0'])
    fb = O.FeatureBankRef(2, int(g['budget']), 'cpu', 0.1, 0.95)
    fb.init_bank([k0[i].clone() for i in range(2)], [v0[i].clone() for i in range(2)])
    for step in range(1, 5):
        for i in range(2):
            fb.info[i][:, 1] += t(g[f'bump_{step}_{i}'])
        k1, v1 = t(g[f'k1_{step}']), t(g[f'v1_{step}'])
        fb.update([k1[i].clone() for i in range(2)], [v1[i].clone() for i in range(2)], step)
        for i in range(2):
            assert tuple(fb.info[i].shape) == g[f'info_{step}_{i}'].shape
            assert (fb.info[i] - t(g[f'info_{step}_{i}'])).abs().max() < 1e-5
            assert np.allclose(checksum(fb.keys[i]), g[f'keysum_{step}_{i}'], rtol=1e-5)
            assert np.allclose(checksum(fb.values[i]), g[f'valsum_{step}_{i}'], rtol=1e-5)
    for i in range(2):
        assert (fb.keys[i] - t(g[f'keys_4_{i}'])).abs().max() < 1e-5
        assert (fb.values[i] - t(g[f'values_4_{i}'])).abs().max() < 1e-5
    assert np.array_equal(fb.peak_n, g['peak_n']) and np.array_equal(fb.replace_n, g['replace_n'])
    if regime == 'evict':
        assert g['replace_n'].sum() > 0
    if regime == 'append':
        assert fb.keys[0].shape[1] == 24 * 5


def test_postprocess_and_pad():
    g = load('postprocess.npz')
    for n in [k[3:] for k in g.files if k.startswith('in_')]:
        assert np.array_equal(O.postprocessing_pred(g['in_' + n].copy()), g['out_' + n]), n
    assert g['out_zeros'].min() == 1                                  # the all-background quirk
    for key, val in meta()['pad_divide_by'].items():
        h, w = [int(x) for x in key.split('x')]
        outs, pad = O.pad_divide_by([torch.zeros(1, 1, h, w)], 16, (h, w))
        assert list(pad) == val['pad'] and list(outs[0].shape[-2:]) == val['shape']


def test_main_loop_labels(sd):
    """oracle.run_clip == the label PNGs written by the reference's test_video_seg.main (6 frames, 120x200
    -> 480x800 inside the loop), including the largest-component post-processing."""
    g = load('main_loop_120x200.npz')
    H, W = [int(x) for x in g['shape']]
    labels = np.unpackbits(g['labels'], axis=-1)[..., :W]
    frames = t(g['frames_u8']).float().div(255)
    out = O.run_clip(sd, frames, t(g['mask']))
    assert np.array_equal(labels[0], g['mask'])
    for i in range(1, frames.shape[0]):
        post = O.postprocessing_pred(out['labels'][i].numpy())
        assert miou(torch.from_numpy(post), torch.from_numpy(labels[i])) > 0.999, i
    assert list(g['palette'][:12]) == [0, 0, 0, 0, 0, 128, 0, 128, 0, 128, 0, 0]


def test_full_size_samples(sd):
    """480x854 (the benchmark shape): sparse samples + checksums of the reference's outputs."""
    from tools import synth
    g = load('full_480x854.npz')
    frames, m0 = synth.clip(1, 2, 480, 854)
    assert np.allclose(checksum(frames), g['frames_sum'], rtol=1e-6)
    oh = synth.onehot(m0).unsqueeze(0)
    k, v = O.memorize(sd, frames[0:1], oh)
    for i in range(2):
        assert (k[i].flatten()[t(g['key_idx'])] - t(g['key_val'][i])).abs().max() < 2e-4
        assert (v[i].flatten()[t(g['val_idx'])] - t(g['val_val'][i])).abs().max() < 2e-4
    fb = O.FeatureBankRef(2, 250000)
    fb.init_bank(k, v)
    score, _ = O.segment(sd, frames[1:2], fb)
    ok, dl, dp = close_logits(score.flatten()[t(g['score_idx'])], t(g['score_val']))
    assert ok, (dl, dp)
    assert abs(float((score[0, 1] > score[0, 0]).float().mean()) - float(g['label_water_frac'])) < 1e-3


def test_segment_batch_and_training_branch_vs_reference():
    """oracle.segment with bs = 2 (eval, padded 90x150) and in the training branch (no pad, scalar uncertainty,
    frozen BatchNorm) against the reference's own AFB_URR.segment (oracle/gen_bs2_golden.py)."""
    from tools import synth
    from golden_util import load, state_dict, t
    sd = state_dict()
    g = load('segment_bs2.npz')
    torch.set_num_threads(8)
    for tag, H, W, training in (('eval_90x150', 90, 150, False), ('train_96x160', 96, 160, True)):
        frames, m0 = synth.clip(6, 3, H, W)
        oh = synth.onehot(m0).unsqueeze(0)
        k, v = O.memorize(sd, frames[0:1], oh)
        fb = O.FeatureBankRef(2, 250000)
        fb.init_bank(k, v)
        score, unc = O.segment(sd, frames[1:3], fb, update_bank=not training, training=training)
        assert tuple(score.shape) == (2, 2, H, W)
        assert (score - t(g[f'{tag}_score'])).abs().max() < 1e-4
        for i in range(2):
            assert (fb.info[i][:, 1] - t(g[f'{tag}_info1'][i])).abs().max() < 1e-5
        if training:
            assert abs(float(unc) - float(g[f'{tag}_uncertainty'])) < 1e-6
        else:
            assert unc is None


def test_oracle_training_step_vs_reference(sd):
    """One training step of THE REFERENCE (train_video_seg.py:56-74 on its own AFB_URR / FeatureBank, loss.backward(),
    AdamW.step(); oracle/gen_train_golden.py) against autograd through the oracle's memorize / segment / loss in the same
    float32 arithmetic, with NOTHING adapted to a device under test: loss, uncertainty, logits, the gradient of all 300
    parameters and the parameters after the optimiser step.  This pins the oracle's backward; tests/test_backward_gpu.py then
    holds the HIP path against the same fixture."""
    import torch.nn.functional as F
    from golden_util import train_sample, train_names, train_positions, compare_grads_with_reference, TRAIN_K, TRAIN_LU, TRAIN_LR
    g = load('train_step_96x160.npz')
    frames, masks, lab = train_sample()
    names = train_names()
    assert len(names) == 300
    leaf = {n: (v.clone().requires_grad_() if n in set(names) else v) for n, v in sd.items()}
    k_ref, v_ref = O.memorize(leaf, frames[0:1], masks[0:1])
    fb = O.FeatureBankRef(TRAIN_K, 300000)
    fb.init_bank(k_ref, v_ref)
    loss, unc, scs = 0.0, 0.0, []
    for i in range(2):                      # (the batch's samples are independent given the bank: batch means = means of the samples')
        sc, un = O.segment(leaf, frames[1 + i:2 + i], fb, update_bank=False, training=True)
        loss = loss + (F.cross_entropy(sc, lab[1 + i:2 + i]) + TRAIN_LU * un) / 2
        unc += un.item() / 2
        scs.append(sc.detach())
    loss.backward()
    tot = loss.item()
    scores = torch.cat(scs, 0)
    assert abs(tot - float(g['loss'])) < 2e-6 * abs(float(g['loss'])), (tot, float(g['loss']))
    assert abs(unc - float(g['uncertainty'])) < 2e-6
    pos = torch.from_numpy(train_positions(scores.numel(), 'scores'))
    assert (scores.flatten()[pos] - t(g['scores_sample'])).abs().max() < 1e-3
    assert abs(float(scores.double().abs().sum()) - float(g['scores_abs_sum'])) < 1e-5 * float(g['scores_abs_sum'])
    worst = compare_grads_with_reference({n: leaf[n].grad for n in names}, g, names)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:5]
    print('oracle backward vs the reference\'s, worst relative errors:', {n: f'{e:.1e}' for n, e in top})
    assert max(worst.values()) < 1e-3, top
    # the optimiser step: torch.optim.AdamW on the oracle's gradients against the reference's parameters after ITS step
    params = [torch.nn.Parameter(sd[n].clone()) for n in names]
    for p_, n in zip(params, names):
        p_.grad = leaf[n].grad.clone()
    torch.optim.AdamW(params, TRAIN_LR).step()
    for i, (p_, n) in enumerate(zip(params, names)):
        d = p_.detach().double() - sd[n].double()
        assert abs(float(d.norm()) - g['step_stats'][i][0]) <= 2e-3 * g['step_stats'][i][0] + 1e-12, n
