"""The per-frame inference loop: drop-in for ``test_video_seg.py`` (reference lines cited inline).

``main(args, device)`` keeps the reference's contract (``test_video_seg.py:41-123``): a directory
of ``*.jpg`` / ``*.png`` frames in, ``./output/segs/<name>/{mask,overlay}/<stem>.png`` out, the
first-frame mask read from ``output/segs/<name>/mask/<first>.png``.  ``ClipRunner`` is the same
loop on tensors already resident in HBM (what ``bench.py`` times and what the clip-sharded
multi-GPU driver in ``dist.py`` calls).

Every per-frame operator is a HIP kernel: bicubic / nearest resize, segment, object softmax,
memorize, bank update, resize + arg-max.  One host synchronisation per frame (the label D2H the
reference also has at ``.cpu()``, ``test_video_seg.py:115``) -- the bank bookkeeping rides on it.
"""
import argparse
import os
import time
from glob import glob

import numpy as np
import torch

from . import ops
from .data import AsyncWriter, color_palette, load_image_in_PIL, save_overlay, save_seg_mask
from .png_device import PngSink
from . import jpeg_device
from . import png_decode
from .dataset import Video_DS
from .feature_bank import FeatureBank
from .model import AFB_URR


def gct():
    """myutils/system.py:56-65 log timestamp."""
    return time.strftime('%m/%d %H:%M:%S', time.localtime(time.time()))


def get_args(argv=None):
    """test_video_seg.py:20-38 plus the harness options SURVEY.md section 5 lists."""
    parser = argparse.ArgumentParser(description='V-FloodNet: Water Video Segmentation (MI355X-native)')
    parser.add_argument('--gpu', type=int, default=0, help='GPU card id.')
    parser.add_argument('--budget', type=int, default='250000',
                        help='Max number of features that feature bank can store. Default: 300000')
    parser.add_argument('--viz', action='store_true', default=True, help='Visualize data.')
    parser.add_argument('--no-viz', dest='viz', action='store_false', help='Skip the overlay PNGs.')
    parser.add_argument('--load-workers', type=int, default=4, help='DataLoader worker processes (frame decoding).')
    parser.add_argument('--png-workers', type=int, default=4, help='Threads that frame and write the PNG files (compressed on the GPU).')
    parser.add_argument('--model-path', type=str, default='records/video_seg_checkpoint_20200212-001734.pth',
                        help='Path to the checkpoint (default: none)')
    parser.add_argument('--update-rate', type=float, default=0.1, help='Update Rate. Impact of merging new features.')
    parser.add_argument('--merge-thres', type=float, default=0.95,
                        help='Merging Rate. If similarity higher than this, then merge, else append.')
    parser.add_argument('--test-path', type=str, required=True, help='Video Path')
    parser.add_argument('--test-name', type=str, required=True, help='Video Name')
    parser.add_argument('--decode', choices=['device', 'pil'], default='device',
                        help='device: JPEG frames are entropy-decoded by the loader workers and reconstructed on the GPU; '
                             'pil: every frame is decoded by PIL (the reference path).')
    parser.add_argument('--size', type=int, default=480, help='Short-edge size the network runs at (reference: 480).')
    parser.add_argument('--mem-every', type=int, default=1, help='Memorise every n-th frame (reference: 1).')
    return parser.parse_args(argv)


def resized_hw(h, w, size):
    """torchvision 0.9.2 ``TF.resize(img, int)``: short edge -> size, long edge ``int(size*long/short)``."""
    short, long = (w, h) if w <= h else (h, w)
    if short == size:
        return h, w
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


class ClipRunner:
    """test_video_seg.py:83-121 on device tensors."""

    def __init__(self, model, obj_n=2, budget=250000, update_rate=0.1, thres_close=0.95, size=480, mem_every=1,
                 postprocess=False, autotune=False, capture_graphs=False):
        self.model = model
        self.group_capture = 0                # (with capture_graphs: also capture launch_group's lists for groups of this many frames in start())
        self.capture_graphs = capture_graphs  # capture the launch lists' HIP graphs in start(), before the loop (long loops: video_seg.main, bench.py)
        self.postprocess = postprocess       # run postprocessing_pred (:116) on the device before the D2H
        self.autotune = autotune             # measure tile / split-K choices for conv shapes the shipped tables lack
        self.device = model.device
        self.obj_n = obj_n
        self.size = size
        self.mem_every = mem_every
        prec = getattr(model, 'precision', None)
        if getattr(model, 'precision_map', None) or os.environ.get('VFN_PRECISION_MAP'):
            prec = model.engine().layer_precision('bank_update')      # (per-layer arithmetic: engine.Engine.layer_mode)
        self.fb = FeatureBank(obj_n, budget, self.device, update_rate=update_rate, thres_close=thres_close, precision=prec)
        self.t = 0
        self._pinned = None
        self._stats_pinned = None
        self._net_cache_cap = 8
        self._net_cache = {}                 # source frame (data_ptr, version, shape) -> (the source frame, its network-resolution tensor)
        self.lookahead = int(os.environ.get('VFN_LOOKAHEAD', 3))    # frames of look-ahead the query side may use (0..3)

    def _net_frame(self, frame):
        """TF.resize(ori_frame, 480, BICUBIC) (:88,:107); identity when the short edge already matches."""
        H0, W0 = frame.shape[-2:]
        h, w = resized_hw(H0, W0, self.size)
        if (h, w) == (H0, W0):
            return frame
        return ops.resize_bicubic(frame.contiguous(), h, w)

    def start(self, first_frame, first_mask_onehot):
        """first_frame f32[1,3,H0,W0]; first_mask_onehot u8/f32[1,obj_n,H0,W0] (:85-101)."""
        H0, W0 = first_frame.shape[-2:]
        self.ori_size = (H0, W0)
        f = self._net_frame(first_frame)
        h, w = f.shape[-2:]
        if self.group_capture > 1:                                 # launch_group's lists exist before the tuner / the capture look at the plan
            self.model.engine().plan(h, w, self.obj_n).batch_set(self.group_capture).dec_batch()
        if self.autotune and os.environ.get('VFN_AUTOTUNE', '1') != '0':    # (VFN_AUTOTUNE=0: the heuristic choices for unlisted shapes)
            self.model.engine().autotune(h, w, self.obj_n, only_missing=True)
        m = first_mask_onehot.to(torch.float32).contiguous()
        if (h, w) != (H0, W0):
            m = ops.resize_nearest(m, h, w)                      # TF.resize(mask, 480, NEAREST) (:89)
        k, v = self.model.memorize(f, m)
        self.fb.init_bank(k, v)
        if self.capture_graphs and os.environ.get('VFN_CAPTURE_AT_START', '1') == '1':
            # the launch lists' HIP graphs are captured here, before the loop (engine.Engine.capture), not on their third run inside it
            # (a capture enters with a device-wide synchronize + gc.collect + empty_cache: ADVICE r5); the first frame's keys / values
            # are already in the bank (init_bank copies them out of the plan's buffers).  Opt-in: a loop of a few frames (most tests)
            # is better off never capturing at all
            self.graph_captures_at_start = self.model.engine().capture(h, w, self.obj_n, group=self.group_capture)
        self.t = 0
        self._alloc_outputs(H0, W0)

    def _alloc_outputs(self, H0, W0):
        if self.lookahead > 0:                 # the look-ahead's stream is probed for a hardware queue of its own (~10 ms): here, not in the loop
            self.model.engine().side_stream()
        self._net_cache = {}                                     # no look-ahead carried over from a previous clip
        # two sets of per-frame outputs: frame t+1 may be enqueued (launch) before the host has looked at frame t
        # (collect), and a side stream may still be compressing frame t's label map while frame t+1 runs
        self._bufs = [dict(label=torch.empty(H0, W0, dtype=torch.uint8, device=self.device),
                           post=torch.empty(H0, W0, dtype=torch.uint8, device=self.device),
                           pinned=torch.empty(H0, W0, dtype=torch.uint8).pin_memory(),
                           stats=torch.zeros(self.obj_n, 4, dtype=torch.int32).pin_memory(),
                           done=torch.cuda.Event()) for _ in range(2)]
        self._cur = self._bufs[0]
        self._label_dev, self._post_dev, self._pinned = self._cur['label'], self._cur['post'], self._cur['pinned']
        self._ccl_scratch = torch.empty(2 * H0 * W0 + 8, dtype=torch.int32, device=self.device)
        self._pending = []
        self._gsets, self._gturn, self._gpending, self._glast = {}, {}, [], None      # launch_group
        if self.group_capture > 1:             # (pinned allocations cost milliseconds each: not inside the loop)
            self._group_set(self.group_capture)
        self.size_log = [list(self.fb._len_host)]                 # live entries per object: after init_bank, then after every collected frame

    def snapshot(self, device='cpu'):
        """Everything frame t + 1 depends on besides the weights, as a dictionary ``torch.save`` takes: the bank
        (``FeatureBank.state_dict``), the frame counter (the bank's birth frames and the ``mem_every`` rhythm are counted
        from it) and the clip geometry.  A long stream stops here and ``resume`` continues it -- bit-identically to the
        uninterrupted loop when both look equally far ahead (the reference has no inference-state checkpoint: SURVEY §5
        lists it as optional).  No frame may be in flight."""
        if self._pending or self._gpending:
            raise RuntimeError('ClipRunner.snapshot: frames in flight; collect() first')
        return dict(version=1, t=int(self.t), ori_size=[int(x) for x in self.ori_size], obj_n=int(self.obj_n), size=int(self.size),
                    mem_every=int(self.mem_every), bank=self.fb.state_dict(device))

    def resume(self, snap):
        """Take up the stream ``snapshot`` left: the next ``launch`` / ``step`` is frame ``snap['t'] + 1``."""
        if snap.get('version') != 1 or snap['obj_n'] != self.obj_n:
            raise ValueError(f'ClipRunner.resume: snapshot version {snap.get("version")!r} with {snap.get("obj_n")} objects, runner with {self.obj_n}')
        if snap['size'] != self.size:
            raise ValueError(f'ClipRunner.resume: snapshot taken at network size {snap["size"]}, runner at {self.size}: the bank holds one frame geometry')
        H0, W0 = snap['ori_size']
        self.ori_size = (H0, W0)
        self.fb.load_state_dict(snap['bank'])
        h, w = resized_hw(H0, W0, self.size)
        if self.autotune and os.environ.get('VFN_AUTOTUNE', '1') != '0':
            self.model.engine().autotune(h, w, self.obj_n, only_missing=True)
        if self.capture_graphs and os.environ.get('VFN_CAPTURE_AT_START', '1') == '1':
            self.graph_captures_at_start = self.model.engine().capture(h, w, self.obj_n)
        self.t = int(snap['t'])
        self._alloc_outputs(H0, W0)

    def _net_cached(self, frame):
        """The network-resolution tensor of ``frame``, resized once however often the look-ahead sees it.  The key is an
        allocator address, so an entry OWNS its source tensor: while the key is live the block cannot be handed out again
        for a later frame (video_seg.main decodes every frame into a fresh ``torch.empty`` with ``_version`` 0; without
        the reference a block recycled 6-7 frames later hit the stale entry of an older frame)."""
        key = (frame.data_ptr(), frame._version, tuple(frame.shape))
        hit = self._net_cache.get(key)
        if hit is None:
            hit = (frame, self._net_frame(frame))
            if len(self._net_cache) >= self._net_cache_cap:
                self._net_cache.pop(next(iter(self._net_cache)))
            self._net_cache[key] = hit
        return hit[1]

    def _look_ahead(self, nxt):
        """The query side of the coming frames (``nxt``: network-resolution tensors of frames t+1, t+2, ...), which depends
        on nothing but those frames: batched over TWO frames and spread over the side stream underneath memorize / update
        of two loop iterations (Engine.prefetch_begin / prefetch_finish)."""
        eng = self.model.engine()
        p = eng.plan(nxt[0].shape[2], nxt[0].shape[3], self.obj_n)
        have = set(eng.prefetched_keys(p))
        keys = [eng._key(f) for f in nxt]
        if keys[0] not in have:
            # frame t+1 is needed next: everything for it (and, batched with it, for t+2) goes out now
            if eng.prefetch_pending(p):
                eng.prefetch_finish(p)
                have = set(eng.prefetched_keys(p))
            if keys[0] not in have:
                pair = [f for f, k in zip(nxt[:2], keys[:2]) if k not in have]
                eng.prefetch_begin(pair, self.obj_n, full=True)
        elif eng.prefetch_pending(p):
            eng.prefetch_finish(p)
        elif len(nxt) >= 2 and keys[1] not in have:
            # t+1 is ready; start on (t+2, t+3): first half now, the rest after the next frame's decoder is on its way
            pair = [f for f, k in zip(nxt[1:3], keys[1:3]) if k not in have]
            eng.prefetch_begin(pair, self.obj_n, full=False)

    def launch(self, frame, next_frame=None, want_label=True, next_frames=None):
        """Enqueue one iteration of the hot loop (:105-116) for ``frame`` f32[1,3,H0,W0] (on the GPU) and return
        without waiting.  ``collect()`` later waits for it and absorbs the bank bookkeeping.  At most two launches may
        be outstanding (the kernels read the true bank lengths from device memory; the host only needs upper bounds
        to size the grids, ``FeatureBank.len_upper``).

        ``next_frames`` (optional, already on the GPU: frames t+1, t+2, t+3; ``next_frame`` = just t+1): lets the query
        side of the coming frames -- which depends on nothing but those frames -- run on a side stream underneath
        memorize / update, two frames per pass.  The labels do not depend on how far the loop looks ahead up to the
        summation order inside the split-K convolutions (the batch of two picks other tile shapes than a single
        frame)."""
        if len(self._pending) >= 2:
            raise RuntimeError('ClipRunner: two frames already in flight; collect() first')
        if self._gpending:
            raise RuntimeError('ClipRunner: a group of frames in flight; collect_group() first')
        self.t += 1
        buf = self._bufs[self.t & 1]
        if buf.get('reader_done') is not None:                    # e.g. the PNG side stream still reading frame t-2
            torch.cuda.current_stream().wait_event(buf['reader_done'])
            buf['reader_done'] = None
        self._cur = buf
        self._label_dev, self._post_dev, self._pinned = buf['label'], buf['post'], buf['pinned']
        f = self._net_cached(frame)                               # (resized when it was prefetched)
        score, _ = self.model.segment(f, self.fb)                 # :108
        pred_mask = ops.softmax_objects(score)                    # :109
        nxt = list(next_frames) if next_frames is not None else ([next_frame] if next_frame is not None else [])
        nxt = [x for x in nxt if x is not None and x.shape[0] == 1][:max(0, self.lookahead)]
        if nxt:
            self._look_ahead([self._net_cached(x) for x in nxt])
        if self.t % self.mem_every == 0:
            k, v = self.model.memorize(f, pred_mask)              # :111 (soft masks are memorised)
            self.fb.update(k, v, self.t)                          # :112
        H0, W0 = self.ori_size
        ops.resize_argmax(pred_mask, H0, W0, out=buf['label'])    # :114-115
        src = buf['label']
        if self.postprocess:                                       # :116, largest 8-connected water blob
            src = ops.postprocess_pred_device(buf['label'], buf['post'], self._ccl_scratch)
        # the frame's single host hand-over: labels + bank bookkeeping
        if want_label:
            buf['pinned'].copy_(src, non_blocking=True)
        buf['stats'].copy_(self.fb.stats_device(), non_blocking=True)
        buf['done'].record()
        buf['t'] = self.t
        self._pending.append(buf)
        return buf

    def collect(self):
        """Wait for the oldest outstanding ``launch`` and take over its bank bookkeeping; returns its uint8 label map
        [H0,W0] as a pinned host tensor (valid until the launch after next)."""
        buf = self._pending.pop(0)
        buf['done'].synchronize()
        self.fb.absorb_stats(buf['stats'], in_flight=len(self._pending))
        self.size_log.append(list(self.fb._len_host))
        return buf['pinned']

    def step(self, frame, want_label=True, next_frame=None, next_frames=None):
        """``launch`` + ``collect``: one iteration of the hot loop, synchronised (one host synchronisation per frame,
        the one the reference also has at ``.cpu()``, test_video_seg.py:115)."""
        self.launch(frame, next_frame, want_label, next_frames)
        lab = self.collect()
        return lab if want_label else None

    # ------------------------------------------------------------------ groups of frames between two memorize calls (mem_every > 1)
    def _group_set(self, G):
        """Two alternating sets of G per-frame outputs (one group may be in flight while the host reads the one before)."""
        G = max(G, self.group_capture)        # (the short group at a clip's end uses the full-size sets: no pinned allocation in the loop)
        sets = self._gsets.get(G)
        if sets is None:
            H0, W0 = self.ori_size
            mk = lambda: dict(label=torch.empty(H0, W0, dtype=torch.uint8, device=self.device),
                              post=torch.empty(H0, W0, dtype=torch.uint8, device=self.device),
                              pinned=torch.empty(H0, W0, dtype=torch.uint8).pin_memory())
            sets = self._gsets[G] = [dict(frames=[mk() for _ in range(G)], prob=None, done=torch.cuda.Event(),
                                          stats=torch.zeros(self.obj_n, 4, dtype=torch.int32).pin_memory()) for _ in range(2)]
            self._gturn[G] = 0
        self._gturn[G] ^= 1
        return sets[self._gturn[G]]

    def group_len(self, available):
        """How many of the next ``available`` frames form the next group: up to and including the next memorised frame."""
        return max(0, min(self.mem_every - self.t % self.mem_every, available))

    def launch_group(self, frames, want_label=True, next_frames=None):
        """Enqueue the hot loop for the G frames ``frames`` (a list of f32[1,3,H0,W0] on the GPU: frames t+1 .. t+G) in ONE batched
        pass (``AFB_URR.segment_group``): they must all see the same bank, i.e. only the LAST of them may be a frame the loop
        memorises (``t % mem_every == 0``) -- with a key-frame interval of n that is groups of up to n frames ending on a key frame
        (BASELINE config C3: n = 5).  Same labels as G ``launch`` calls up to the summation order of the larger GEMMs; the bank
        is updated once, behind the group, exactly as the frame-by-frame loop would have.  ``collect_group()`` returns the G label
        maps.  At most two groups may be outstanding; do not mix with ``launch`` while one is.  ``next_frames``: the frames of the
        group after this one (already on the GPU) -- their frame-only side then runs on the side stream underneath this group's
        memorize / update (``Engine.prefetch_group``)."""
        G = len(frames)
        if G < 1:
            raise ValueError('launch_group: no frames')
        if self._pending:
            raise RuntimeError('ClipRunner: single frames in flight; collect() first')
        if len(self._gpending) >= 2:
            raise RuntimeError('ClipRunner: two groups already in flight; collect_group() first')
        for g in range(G - 1):
            if (self.t + 1 + g) % self.mem_every == 0:
                raise ValueError(f'launch_group: frame {self.t + 1 + g} is memorised (mem_every = {self.mem_every}) but is not the last '
                                 f'of the group: the frames behind it would not see its update')
        self._net_cache_cap = max(self._net_cache_cap, 2 * G + 2)    # this group's frames and the next one's stay resident
        gs = self._group_set(G)
        outs = gs['frames'][:G]
        for fo in outs:
            if fo.get('reader_done') is not None:
                torch.cuda.current_stream().wait_event(fo['reader_done'])
                fo['reader_done'] = None
        nets = [self._net_cached(f) for f in frames]
        # (next_frames: the frame-only side of the NEXT group goes to the side stream behind this group's decoder, i.e. underneath
        # its memorize / update / label tail -- Engine.prefetch_group)
        nxt = [self._net_cached(f) for f in next_frames] if next_frames and self.lookahead > 0 else None
        scores = self.model.segment_group(nets, self.fb, prefetch=nxt)        # :108 for the G frames
        if gs['prob'] is None or gs['prob'].shape[1:] != scores.shape[1:] or gs['prob'].shape[0] < G:
            gs['prob'] = torch.empty((len(gs['frames']),) + tuple(scores.shape[1:]), device=scores.device, dtype=scores.dtype)
        for g in range(G):
            ops.softmax_objects(scores[g:g + 1], out=gs['prob'][g:g + 1])    # :109
        self.t += G
        has_update = self.t % self.mem_every == 0
        if has_update:
            k, v = self.model.memorize(nets[-1], gs['prob'][G - 1:G])        # :111-112
            self.fb.update(k, v, self.t)
        H0, W0 = self.ori_size
        for g in range(G):
            fo = outs[g]
            ops.resize_argmax(gs['prob'][g:g + 1], H0, W0, out=fo['label'])  # :114-115
            src = fo['label']
            if self.postprocess:
                src = ops.postprocess_pred_device(fo['label'], fo['post'], self._ccl_scratch)
            if want_label:
                fo['pinned'].copy_(src, non_blocking=True)
        gs['stats'].copy_(self.fb.stats_device(), non_blocking=True)
        gs['done'].record()
        gs['has_update'], gs['G'], gs['t'] = has_update, G, self.t
        self._gpending.append(gs)
        self._glast = gs
        return gs

    def collect_group(self):
        """Wait for the oldest outstanding ``launch_group``; returns its G uint8 label maps [H0,W0] (pinned host tensors, valid
        until the group after next of the same size)."""
        gs = self._gpending.pop(0)
        gs['done'].synchronize()
        before = list(self.fb._len_host)
        self.fb.absorb_stats(gs['stats'], in_flight=sum(1 for x in self._gpending if x['has_update']))
        self.size_log += [before] * (gs['G'] - 1) + [list(self.fb._len_host)]
        return [fo['pinned'] for fo in gs['frames'][:gs['G']]]

    def step_group(self, frames, want_label=True, next_frames=None):
        self.launch_group(frames, want_label, next_frames)
        labs = self.collect_group()
        return labs if want_label else None

    def group_labels_device(self):
        """The label maps of the last group as they left the GPU (post-processed when ``postprocess``), still on the device."""
        return [fo['post'] if self.postprocess else fo['label'] for fo in self._glast['frames'][:self._glast['G']]]

    def label_device(self):
        """The label map of the last step as it left the GPU (post-processed when ``postprocess``), still on the device."""
        return self._post_dev if self.postprocess else self._label_dev

    def bank_sizes(self):
        return list(self.fb._len_host)


def run_clip(model, frames, first_mask_u8, budget=250000, update_rate=0.1, thres_close=0.95, size=480,
             mem_every=1, postprocess=False, overlap=True, group=False):
    """frames f32[T,3,H0,W0] on the GPU, first mask u8[H0,W0] (>0 = water).
    Returns labels u8[T,H0,W0] (host; frame 0 = the given mask) and per-frame bank sizes.
    ``group`` (with ``mem_every`` > 1): the frames between two memorised frames as one batched pass (``ClipRunner.launch_group``)."""
    T, _, H0, W0 = frames.shape
    m = (first_mask_u8 > 0).to(torch.uint8)
    onehot = torch.stack([1 - m, m], 0).unsqueeze(0).to(frames.device)        # Water_DS.py:93-101
    runner = ClipRunner(model, 2, budget, update_rate, thres_close, size, mem_every, postprocess=postprocess)
    runner.start(frames[0:1], onehot)
    labels = torch.empty(T, H0, W0, dtype=torch.uint8)
    labels_np = labels.numpy()
    labels[0] = m.cpu()
    sizes = []
    if group and mem_every > 1:
        t = 1
        while t < T:
            g = runner.group_len(T - t)
            nxt = [frames[u:u + 1] for u in range(t + g, min(T, t + g + mem_every))] if overlap else None
            labs = runner.step_group([frames[u:u + 1] for u in range(t, t + g)], next_frames=nxt or None)
            for i, lab in enumerate(labs):
                np.copyto(labels_np[t + i], lab.numpy())
            sizes += [list(x) for x in runner.size_log[-g:]]
            t += g
        return dict(labels=labels, bank_sizes=sizes, fb=runner.fb)
    for t in range(1, T):
        lab = runner.step(frames[t:t + 1], next_frames=[frames[u:u + 1] for u in range(t + 1, min(T, t + 4))] if overlap else None)
        # plain memcpy: a torch CPU copy_ wakes the whole OpenMP pool (128 threads on the MI355X hosts), whose
        # spinning starves this launch thread -- measured 41 instead of 7.9 ms per frame
        np.copyto(labels_np[t], lab.numpy())
        sizes.append(runner.bank_sizes())
    return dict(labels=labels, bank_sizes=sizes, fb=runner.fb)


def main(args, device):
    """test_video_seg.py:41-123."""
    model = AFB_URR(device, update_bank=True, load_imagenet_params=False)
    model = model.to(device)
    model.eval()

    downsample_size = getattr(args, 'size', 480)

    if os.path.isfile(args.model_path):
        checkpoint = torch.load(args.model_path, map_location='cpu')
        end_epoch = checkpoint['epoch']
        model.load_state_dict(checkpoint['model'], strict=False)
        train_loss = checkpoint['loss']
        seed = checkpoint['seed']
        print(gct(), f'Loaded checkpoint {args.model_path}. (end_epoch: {end_epoch}, train_loss: {train_loss}, seed: {seed})')
    else:
        print(gct(), f'No checkpoint found at {args.model_path}')
        raise IOError

    img_list = sorted(glob(os.path.join(args.test_path, '*.jpg')) + glob(os.path.join(args.test_path, '*.png')))
    first_frame = load_image_in_PIL(img_list[0])
    first_name = os.path.basename(img_list[0])[:-4]

    out_dir = './output/segs'
    mask_dir = os.path.join(out_dir, args.test_name, 'mask')
    mask_path = os.path.join(mask_dir, first_name + '.png')
    if not os.path.exists(mask_path):
        # test_video_seg.py:67-69: bootstrap the first mask with the image model -- LinkNet over EfficientNet-B4 on the HIP
        # path (linknet.LinknetB4 behind image_seg.test_waterseg; its parameters are read by name out of the reference's file)
        image_model_path = getattr(args, 'image_model_path', None) or './records/link_efficientb4_model.pth'
        if not os.path.isfile(image_model_path):
            raise IOError(f'first-frame mask {mask_path} not found and no image model at {image_model_path}')
        from .image_seg import test_waterseg
        test_waterseg(image_model_path, img_list[0], args.test_name, out_dir, device)

    first_mask = load_image_in_PIL(mask_path, 'P')
    # JPEG frames: the workers undo the entropy coding only, the rest of the decode runs on the GPU (jpeg_device);
    # PNG frames: the workers inflate them, the scanline filters are undone on the GPU (png_decode).  Either way a frame
    # crosses PCIe at ~1 byte per sample.
    seq_dataset = Video_DS(img_list, first_frame, first_mask, decode=getattr(args, 'decode', 'device'), raw_u8=True)
    # (the reference uses one worker; the per-item hand-over costs more than the decode)
    n_load = int(getattr(args, 'load_workers', 4))
    seq_loader = torch.utils.data.DataLoader(seq_dataset, batch_size=1, shuffle=False, num_workers=n_load,
                                             prefetch_factor=4 if n_load > 0 else None, collate_fn=Video_DS.collate)

    seg_dir = os.path.join(out_dir, args.test_name, 'mask')
    os.makedirs(seg_dir, exist_ok=True)
    if args.viz:
        overlay_dir = os.path.join(out_dir, args.test_name, 'overlay')
        os.makedirs(overlay_dir, exist_ok=True)

    obj_n = seq_dataset.obj_n
    runner = ClipRunner(model, obj_n, args.budget, update_rate=args.update_rate, thres_close=args.merge_thres,
                        size=downsample_size, mem_every=getattr(args, 'mem_every', 1), postprocess=True, autotune=True, capture_graphs=True)
    if runner.mem_every > 1 and os.environ.get('VFN_MAIN_GROUP', '1') != '0':
        runner.group_capture = runner.mem_every        # (the batched lists are built, tuned and captured in start(), before the loop)

    ori_first_frame = seq_dataset.first_frame.unsqueeze(0).to(device)
    ori_first_mask = seq_dataset.first_mask.unsqueeze(0).to(device)

    pred = seq_dataset.first_mask.numpy().argmax(0).astype(np.uint8)          # :91
    keep = [torch.from_numpy(pred).to(device)] if getattr(args, 'keep_labels', False) else None   # (batch_video_seg: mask gather)
    seg_path = os.path.join(seg_dir, f'{first_name}.png')
    save_seg_mask(pred, seg_path, color_palette)
    if args.viz:
        overlay_path = os.path.join(overlay_dir, f'{first_name}.png')
        save_overlay(ori_first_frame[0], pred, overlay_path, color_palette)

    # host-side pipelining around the GPU loop (results identical to the sequential reference loop):
    #   * one frame of look-ahead, so the next frame's encoder work overlaps this frame's memorize (ClipRunner.launch)
    #   * frame t+1 is enqueued before the host waits for frame t (launch / collect): decoding hand-over, uploads and
    #     the bank bookkeeping of one frame hide under the kernels of the next
    #   * the PNGs are compressed on the GPU on a side stream (png_device.PngSink); writer threads only frame + write
    writer = AsyncWriter(getattr(args, 'png_workers', 4))
    with torch.no_grad():
        # The loader is drained by a thread of its own: ``next()`` on an exhausted DataLoader iterator joins the worker processes
        # (~100 ms), which on the launch thread left the device idle for the last frames of every clip
        # (profiles/r06_main_throughput.txt: one 86-110 ms gap per 100-frame clip = 0.9 ms per frame of a short clip).
        import queue
        import threading
        feed = queue.Queue(maxsize=16)

        loader_it = iter(seq_loader)             # (the worker processes are forked HERE, on this thread, before the first HIP graph capture)

        def _drain():
            try:
                for item_ in loader_it:
                    feed.put(item_)
            except BaseException as exc_:         # (a decode error in a worker: re-raised on the launch thread)
                feed.put(exc_)
            feed.put(None)
        threading.Thread(target=_drain, name='vfn-loader', daemon=True).start()

        def _next_item():
            item_ = feed.get()
            if isinstance(item_, BaseException):
                raise item_
            return item_
        it = iter(_next_item, None)              # (workers start decoding while the first frame is memorised)
        runner.start(ori_first_frame, ori_first_mask)
        side_ = model.engine().side_stream()

        # Uploads go through pinned staging buffers on their own stream: a pageable-memory ``.to(device)`` is ordered
        # behind everything already enqueued on the compute stream and blocks the host until it has run -- that would
        # undo the launch / collect pipelining.  The rest of the decode (JPEG: inverse DCT ..., PNG: the scanline
        # filters, a serial recurrence that keeps ONE CU busy for ~0.8 ms per 480p frame) runs on a stream of its own, two
        # frames ahead of the network, so that it sits underneath the kernels of the frames before it.
        # (streams are picked so that none shares a hardware queue with the frame loop's, _lib.independent_stream; uploads and decode
        # kernels share ONE: five live streams on four hardware queues is how the PNG sink came to run in order with the loop)
        from ._lib import independent_stream
        copy_stream = decode_stream = independent_stream(device, beside=[side_])
        sink = PngSink(device, writer, beside=[side_, decode_stream])
        main_stream = torch.cuda.current_stream()
        staging = {}
        slot = [0]
        N_SLOTS = 6                              # frames t .. t+3 in flight + the one being filled + one spare

        def to_device_async(t):
            key = (tuple(t.shape), t.dtype, slot[0] % N_SLOTS)
            if key not in staging:
                staging[key] = (torch.empty(t.shape, dtype=t.dtype).pin_memory(), torch.cuda.Event())
            pinned, free = staging[key]
            free.synchronize()                   # the copy that last used this slot has finished (long ago)
            np.copyto(pinned.numpy(), t.numpy())     # plain memcpy (a torch CPU copy_ would wake the whole OpenMP pool)
            with torch.cuda.stream(copy_stream):
                d = pinned.to(device, non_blocking=True)
                free.record()
            torch.cuda.current_stream().wait_stream(copy_stream)
            d.record_stream(torch.cuda.current_stream())
            return d

        def upload(item):                        # the rest of Video_DS.__getitem__ (Water_DS.py:105-109) on the device
            slot[0] += 1
            payload = item[0]
            with torch.cuda.stream(decode_stream):
                if isinstance(payload, dict) and 'jpeg' in payload:
                    coef, qt, info = payload['jpeg']
                    t = jpeg_device.to_tensor(to_device_async(coef), to_device_async(qt), info, device).unsqueeze(0)
                elif isinstance(payload, dict) and 'png' in payload:
                    filtered, info, pal = payload['png']
                    t = png_decode.to_tensor(to_device_async(filtered), info, to_device_async(pal), device).unsqueeze(0)
                else:
                    u8 = payload['u8'] if isinstance(payload, dict) else payload
                    t = ops.to_tensor_device(to_device_async(u8)).unsqueeze(0)
                ready = torch.cuda.Event()
                ready.record()
            t.record_stream(main_stream)         # allocated on the decode stream, consumed on the compute stream
            return item, t, ready

        from collections import deque
        ahead = deque()                          # (item, frame on the device, its decode-finished event)

        # a key-frame interval (--mem-every n > 1): the frames between two memorize calls go through the network as ONE batched pass
        # (ClipRunner.launch_group; VFN_MAIN_GROUP=0: frame by frame).  The loop then keeps two groups of frames decoded ahead
        group_n = runner.mem_every if (runner.mem_every > 1 and os.environ.get('VFN_MAIN_GROUP', '1') != '0') else 0
        n_ahead = max(4, 2 * group_n)
        N_SLOTS = max(N_SLOTS, n_ahead + 2)

        def fill():
            while len(ahead) < n_ahead:
                item = next(it, None)
                if item is None:
                    return
                ahead.append(upload(item))

        prof = [0.0, 0.0, 0.0, 0.0, 0] if os.environ.get('VFN_MAIN_TIMING') else None   # host seconds: fill, launch, save, collect
        dev_marks = []
        fill()
        while ahead and group_n:
            g = runner.group_len(len(ahead))                      # up to and including the next key frame
            grp = [ahead.popleft() for _ in range(g)]
            fill()
            nxt = list(ahead)[:min(group_n, len(ahead))] if runner.lookahead > 0 else []
            for x_ in grp + nxt:
                main_stream.wait_event(x_[2])
            gs = runner.launch_group([x_[1] for x_ in grp], want_label=False, next_frames=[x_[1] for x_ in nxt] or None)
            for (item_, dev_, _), fo, lab in zip(grp, gs['frames'], runner.group_labels_device()):
                name = item_[1]
                if keep is not None:
                    keep.append(lab.clone())
                fo['reader_done'] = sink.save(lab, os.path.join(seg_dir, f'{name}.png'), color_palette,
                                              frame=dev_[0] if args.viz else None,
                                              overlay_path=os.path.join(overlay_dir, f'{name}.png') if args.viz else None)
            if len(runner._gpending) == 2:
                runner.collect_group()
        while runner._gpending:
            runner.collect_group()
        while ahead:
            t0 = time.perf_counter()
            cur, cur_dev, cur_ready = ahead.popleft()
            fill()                               # frame t+2 starts decoding now, a whole frame before it is needed
            t1 = time.perf_counter()
            if prof is not None:                 # device-side: when the stream got here / got past the waits / finished the frame
                ev_a = torch.cuda.Event(enable_timing=True)
                ev_a.record()
            main_stream.wait_event(cur_ready)
            for a_ in list(ahead)[:max(0, runner.lookahead)]:
                main_stream.wait_event(a_[2])            # (their decodes were enqueued one to three iterations ago; the frame uploaded
                                                         # in THIS iteration is not looked at yet: waiting for it too held every frame
                                                         # 0.3-0.4 ms behind an upload that had only just been enqueued)
            if prof is not None:
                ev_b = torch.cuda.Event(enable_timing=True)
                ev_b.record()
            buf = runner.launch(cur_dev, next_frames=[a_[1] for a_ in ahead], want_label=False)   # postprocessing_pred (:116) runs on the GPU
            if prof is not None:
                ev_c = torch.cuda.Event(enable_timing=True)
                ev_c.record()
                dev_marks.append((ev_a, ev_b, ev_c, t0, t1, time.perf_counter()))
            t2 = time.perf_counter()
            name = cur[1]
            if keep is not None:
                keep.append(runner.label_device().clone())
            buf['reader_done'] = sink.save(runner.label_device(), os.path.join(seg_dir, f'{name}.png'), color_palette,
                                           frame=cur_dev[0] if args.viz else None,
                                           overlay_path=os.path.join(overlay_dir, f'{name}.png') if args.viz else None)
            t3 = time.perf_counter()
            if len(runner._pending) == 2:
                runner.collect()                 # frame t-1: its bank statistics
            if prof is not None:
                t4 = time.perf_counter()
                for i_, d_ in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                    prof[i_] += d_
                prof[4] += 1
        if prof is not None and prof[4]:
            print('main loop host time per frame: next()+upload %.2f ms, launch %.2f ms, png sink %.2f ms, collect (waits for frame t-1) %.2f ms'
                  % tuple(1e3 * v / prof[4] for v in prof[:4]))
        while runner._pending:
            runner.collect()
        if prof is not None and len(dev_marks) > 12:
            torch.cuda.synchronize()
            mk = dev_marks[8:]
            n_ = len(mk) - 1
            blocked = sum(m_[0].elapsed_time(m_[1]) for m_ in mk) / len(mk)
            frame = sum(m_[1].elapsed_time(m_[2]) for m_ in mk) / len(mk)
            period = mk[0][0].elapsed_time(mk[-1][0]) / n_
            gaps = sorted(((mk[j_][2].elapsed_time(mk[j_ + 1][0]), j_ + 9) for j_ in range(len(mk) - 1)), reverse=True)
            print('main loop: the largest gaps between a frame\'s last marker and the next frame\'s first (ms, frame): '
                  + ' '.join('%.2f@%d' % g_ for g_ in gaps[:8]) + '; median %.3f' % gaps[len(gaps) // 2][0])
            print('main loop device time per frame: period %.2f ms = %.2f ms the main stream is blocked in front of the frame (decode / look-ahead '
                  'events) + %.2f ms from the first kernel of the frame to its last (queue idle between frames: %.2f ms)'
                  % (period, blocked, frame, period - blocked - frame))
        if not png_decode.check_status(device):          # second line of defence: png_decode.inflate rejects such a frame itself
            raise RuntimeError('corrupt PNG frame data (invalid scanline filter type) in ' + args.test_path)
    writer.close()

    runner.fb.print_peak_mem()
    runner.kept_labels = torch.stack(keep, 0) if keep is not None else None    # uint8 [T,H0,W0] on the device, frame 0 = given mask
    runner.kept_sizes = torch.tensor(runner.size_log, dtype=torch.int32)       # int32 [T,obj_n]: the bank after frame 0, 1, ... (dist.gather_bank_sizes)
    return runner


if __name__ == '__main__':
    args = get_args()
    print(gct(), 'Args =', args)
    if args.gpu >= 0 and torch.cuda.is_available():
        device = torch.device('cuda', args.gpu)
    else:
        raise ValueError('CUDA is required. --gpu must be >= 0.')
    assert os.path.isdir(args.test_path)
    main(args, device)
    print(gct(), 'Test video segmentation done.')
