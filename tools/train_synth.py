"""A margin-bearing checkpoint without a network or a dataset: the synthetic recipe's weights (tools/synth.make_state_dict) trained
on synthetic clips with the HIP training step (vfloodnet_amd.train.train_step = train_video_seg.py:56-76).

Test / measurement infrastructure (like tools/synth.py): the reference ships neither its trained checkpoint
(test_video_seg.py:28) nor data, and random weights have no logit margin -- which is what made plain bf16 look unusable
(DESIGN.md 3.4).  ``train_checkpoint`` gives the state dict after ``steps`` optimizer steps on samples shaped like
``Water_Image_Train_DS`` output (6 frames of 400 x 400, ground-truth masks of every frame, two objects): a pool frame
(tools/synth.frame0: tinted, textured water under a sinusoidal shoreline) at a random offset, translated by a random (dy, dx)
per frame, upside down half of the time.  Loss: CrossEntropy + 0.5 * uncertainty (train_video_seg.py:73-74); AdamW, BatchNorm
frozen at the calibrated statistics (train_video_seg.py:103-106).  Bit-reproducible for a given (seed, steps, lr, size)."""
import math
import time

import torch

from tools import synth


def train_checkpoint(device, steps=3000, lr=2e-5, size=400, pool=48, seed=20200212, log=None, task='easy'):
    """-> (state_dict on the CPU, info dict).  ``log``: callable for progress lines (None: silent).
    ``task``: 'easy' = tools/synth.frame0 (tinted water, clean labels: trains to saturated logits); 'hard' (round 6) =
    tools/synth.frame0_hard (water differs from land by texture only) with annotation noise along the shoreline on every frame's
    label (tools/synth.noisy_labels: ~10 % of the labels disagree with the image) -- a loss that plateaus well above zero and
    logits whose margins do not sit at the clamp."""
    from vfloodnet_amd import AFB_URR, train as T
    g_host = torch.Generator().manual_seed(12345)
    g_noise = torch.Generator().manual_seed(777)
    threads = torch.get_num_threads()
    make = synth.frame0 if task == 'easy' else synth.frame0_hard
    pf, pm = zip(*[make(100000 + i, size, size) for i in range(pool)])     # (multi-octave texture: ~50 ms each on the host)
    pool_f, pool_m = torch.stack(pf, 0).to(device), torch.stack(pm, 0).to(device).long()
    torch.set_num_threads(1)       # (an idle OpenMP pool spinning on the host's cores starves the launch thread)

    def sample(Tn=6):
        r = torch.randint(0, 1 << 30, (6,), generator=g_host).tolist()
        i, oy, ox = r[0] % pool, r[1] % size, r[2] % size
        dy, dx = r[3] % 19 - 9, r[4] % 19 - 9
        fr = torch.stack([torch.roll(pool_f[i], (oy + dy * t, ox + dx * t), (1, 2)) for t in range(Tn)], 0)
        lab = torch.stack([torch.roll(pool_m[i], (oy + dy * t, ox + dx * t), (0, 1)) for t in range(Tn)], 0)
        if task != 'easy':
            lab = torch.stack([synth.noisy_labels(lab[t], g_noise) for t in range(Tn)], 0)
        if r[5] & 1:
            fr, lab = fr.flip(2), lab.flip(1)
        return fr.contiguous(), torch.nn.functional.one_hot(lab, 2).permute(0, 3, 1, 2).float().contiguous()

    model = AFB_URR(device, update_bank=False).to(device)
    model.load_state_dict(synth.make_state_dict(seed), strict=True)
    model.train()
    opt = T.AdamW(model.named_parameters(), lr=lr)
    losses, uncs, t0 = [], [], time.perf_counter()
    good, good_step, restores, lr_now, s = None, 0, 0, lr, 0
    while s < steps:
        opt.lr = lr_now * min(1.0, (s - good_step + 1) / 100.0)                    # warm-up (again after a restore)
        fr, mk = sample()
        loss, unc = T.train_step(model, opt, fr, mk, 0.5)
        losses.append(loss)
        uncs.append(unc)
        # a collapse (every pixel 0.5 / 0.5: uncertainty 1, loss ln 2 + 0.5) is what lr = 1e-4 did to the random trunk within 100 steps
        collapsed = (not math.isfinite(loss)) or (len(uncs) >= 20 and min(uncs[-20:]) > 0.995)
        if collapsed and good is not None and restores < 6:
            restores += 1
            lr_now *= 0.5
            if log:
                log(f'step {s}: collapsed (loss {loss:.4f}, uncertainty {unc:.4f}) - back to the snapshot of step {good_step}, lr -> {lr_now:g}')
            model.load_state_dict(good, strict=True)
            model.train()
            opt = T.AdamW(model.named_parameters(), lr=lr_now)
            s, losses, uncs = good_step, losses[:good_step], uncs[:good_step]
            continue
        if s % 100 == 0 and not collapsed:
            good, good_step = {k: v.detach().clone() for k, v in model.state_dict().items()}, s
        if log and (s % 100 == 0 or s == steps - 1):
            log(f'step {s}: loss {loss:.4f} (uncertainty {unc:.4f}), mean of last 50 {sum(losses[-50:]) / len(losses[-50:]):.4f}, lr {opt.lr:g}, '
                f'{time.perf_counter() - t0:.0f} s')
        s += 1
    train_s = time.perf_counter() - t0
    model.eval()
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    mean = lambda x: round(float(sum(x) / max(1, len(x))), 4)
    info = {'task': task, 'steps_run': len(losses), 'lr': lr, 'lr_final': lr_now, 'restores_after_collapse': restores,
            'sample': f'6 frames of {size}x{size}, 2 objects, synthetic (a pool of {pool} tools/synth.{make.__name__} images at a random offset, rolled by a '
                      f'random step per frame' + ('' if task == 'easy' else '; every label displaced and flipped along the shoreline, tools/synth.noisy_labels') + ')',
            'seconds': round(train_s, 1), 'ms_per_step_incl_host_data': round(1e3 * train_s / max(1, len(losses)), 2),
            'loss_first_50': mean(losses[:50]), 'loss_last_50': mean(losses[-50:]),
            'loss_every_250': [mean(losses[i:i + 50]) for i in range(0, len(losses), 250)]}
    del model, opt
    torch.cuda.empty_cache()
    torch.set_num_threads(threads)
    return sd, info
