"""TEST INFRASTRUCTURE (build container only): ONE TRAINING STEP OF THE REFERENCE ITSELF -> tests/golden/train_step_96x160.npz.

The body of ``train_model`` in /root/reference/train_video_seg.py:56-74 -- ``memorize(frames[0:1], masks[0:1])``, ``init_bank``,
``segment(frames[1:], fb)``, ``CrossEntropyLoss + lu * uncertainty``, ``loss.backward()``, ``AdamW(lr=1e-5).step()`` -- is run on
the reference's own ``AFB_URR`` / ``FeatureBank`` (imported under oracle/refstubs.py) with the model prepared as
train_video_seg.py:103-109 prepares it (``model.train()``, ``model.apply(set_bn_eval)``, ``update_bank=False``,
``torch.optim.AdamW(params, lr)``), on a 3-frame 96x160 sample (reference frame + a batch of two), lu = 0.5, float32 on the CPU,
8 threads.  The inputs are the ones tests/test_backward_gpu.py::test_train_step_vs_reference_loop builds from the seed.

Stored (the fixture is data: inputs are regenerated from the seed by the tests, outputs are the reference's numbers):
  loss, cross entropy, uncertainty, the scores' checksum and 64 sampled logits;
  for EVERY trainable parameter (300 tensors, state-dict order): L2 norm, sum and largest magnitude of its gradient (float64)
  and 16 elements of it at fixed positions; the full gradient of 24 small tensors (biases, BatchNorm weights, a stem);
  after ``optimizer.step()``: per parameter the L2 norm / sum of (after - before) in float64 and the same 16 elements of it.

    python oracle/gen_train_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 20200212
H, W, K, LU, LR = 96, 160, 2, 0.5, 1e-5
N_SAMPLE = 16
FULL = ['keyval_r4.Key.bias', 'keyval_r4.Value.bias', 'decoder.convFM.bias', 'decoder.ResMM.conv1.bias', 'decoder.ResMM.conv2.bias',
        'decoder.RF3.convFS.bias', 'decoder.RF3.ResMM.conv2.bias', 'decoder.RF2.convFS.bias', 'decoder.RF2.ResFS.conv1.bias',
        'decoder.RF2.ResMM.conv2.bias', 'decoder.pred2.weight', 'decoder.pred2.bias', 'decoder.local_convFM.bias',
        'decoder.local_ResMM.conv1.bias', 'decoder.local_pred2.weight', 'decoder.local_pred2.bias',
        'encoder_q.bn1.weight', 'encoder_q.bn1.bias', 'encoder_m.bn1.weight', 'encoder_m.conv1_m.weight', 'encoder_m.conv1_o.weight',
        'encoder_q.res4.5.bn3.weight', 'encoder_m.res2.0.downsample.1.bias', 'encoder_m.res3.1.bn2.weight']


def sample_inputs():
    """The 3-frame sample of tests/test_backward_gpu.py::test_train_step_vs_reference_loop (clip seed 6, 5 % label noise)."""
    from tools import synth
    frames, m0 = synth.clip(6, 3, H, W)
    gen = torch.Generator().manual_seed(11)
    lab = torch.stack([m0.long()] + [torch.roll(m0.long(), (2 * t, 5 * t), (0, 1)) for t in (1, 2)], 0)       # [3,H,W]
    flip = torch.rand(3, H, W, generator=gen) < 0.05
    lab = torch.where(flip, 1 - lab, lab)
    masks = torch.nn.functional.one_hot(lab, K).permute(0, 3, 1, 2).float()                                    # [3,K,H,W]
    return frames, masks, lab


def sample_positions(numel, name):
    """16 fixed positions inside a tensor of ``numel`` elements (name-keyed, so every tensor is probed elsewhere)."""
    import zlib
    g = np.random.default_rng(zlib.crc32(name.encode()))
    return np.sort(g.integers(0, numel, size=N_SAMPLE)).astype(np.int64)


def main():
    from tools import synth
    from oracle import refstubs
    ref = refstubs.import_reference()
    torch.set_grad_enabled(True)          # (importing test_video_seg.py switches autograd off process-wide, test_video_seg.py:17;
                                          # train_video_seg.py runs with it on)
    torch.set_num_threads(8)
    cpu = torch.device('cpu')
    sd = synth.make_state_dict(SEED)
    model = ref.AFB_URR(cpu, update_bank=False, load_imagenet_params=False)          # train_video_seg.py:101
    model.load_state_dict(sd, strict=True)
    model.train()                                                                    # :102
    model.apply(ref.myutils.set_bn_eval)                                             # :103 turn-off BN
    params = model.parameters()
    optimizer = torch.optim.AdamW(filter(lambda x: x.requires_grad, params), LR)     # :105-106
    criterion = torch.nn.CrossEntropyLoss()                                          # :141

    frames, masks, lab = sample_inputs()
    obj_n = K
    # ---- train_video_seg.py:65-74, verbatim semantics
    fb_global = ref.FeatureBank(obj_n, 300000, cpu)
    k4_list, v4_list = model.memorize(frames[0:1], masks[0:1])
    fb_global.init_bank(k4_list, v4_list)
    scores, uncertainty = model.segment(frames[1:], fb_global)
    label = torch.argmax(masks[1:], dim=1).long()
    optimizer.zero_grad()
    ce = criterion(scores, label)
    loss = ce + LU * uncertainty
    loss.backward()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    optimizer.step()

    out = {'loss': np.float64(loss.item()), 'cross_entropy': np.float64(ce.item()), 'uncertainty': np.float64(uncertainty.item()),
           'scores_sum': np.float64(scores.detach().double().sum().item()),
           'scores_abs_sum': np.float64(scores.detach().double().abs().sum().item())}
    pos = sample_positions(scores.numel(), 'scores')[:N_SAMPLE]
    out['scores_sample'] = scores.detach().flatten()[torch.from_numpy(pos)].numpy()
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert len(names) == 300, len(names)
    gstat = np.zeros((len(names), 3), np.float64)
    gsamp = np.zeros((len(names), N_SAMPLE), np.float32)
    dstat = np.zeros((len(names), 2), np.float64)
    dsamp = np.zeros((len(names), N_SAMPLE), np.float64)
    for i, (n, p) in enumerate((n_, p_) for n_, p_ in model.named_parameters() if p_.requires_grad):
        g = p.grad.detach()
        assert g is not None, n
        g64 = g.double()
        gstat[i] = (g64.norm().item(), g64.sum().item(), g64.abs().max().item())
        idx = torch.from_numpy(sample_positions(g.numel(), n))
        gsamp[i] = g.flatten()[idx].numpy()
        d = p.detach().double() - before[n].double()
        dstat[i] = (d.norm().item(), d.sum().item())
        dsamp[i] = d.flatten()[idx].numpy()
    out.update(grad_stats=gstat, grad_samples=gsamp, step_stats=dstat, step_samples=dsamp)
    for n in FULL:
        out['full_grad.' + n] = dict(model.named_parameters())[n].grad.detach().numpy()
    path = os.path.join(ROOT, 'tests', 'golden', 'train_step_96x160.npz')
    np.savez_compressed(path, **out)
    with open(os.path.join(ROOT, 'tests', 'golden', 'train_step_names.txt'), 'w') as f:
        f.write('\n'.join(names) + '\n')
    print(f'loss {loss.item():.6f} (ce {ce.item():.6f}, uncertainty {uncertainty.item():.6f}); {len(names)} gradients; '
          f'largest |g| {gstat[:, 2].max():.3e}; written {os.path.getsize(path) / 1e3:.1f} kB')


if __name__ == '__main__':
    main()
