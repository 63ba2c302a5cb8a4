"""End-to-end rate of the drop-in loop ``vfloodnet_amd.video_seg.main`` (PNG frames on disk -> mask / overlay PNGs on
disk), the part of the path the kernel benchmark leaves out.  usage: main_throughput.py [frames] [viz 0|1]; FRAME_FMT=png
writes the input frames as PNG instead of JPEG; DECODE=pil decodes them with PIL in the workers"""
import sys, os, time, argparse, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vfloodnet_amd
from PIL import Image
from vfloodnet_amd import video_seg
from tools import synth
from vfloodnet_amd.data import save_seg_mask, color_palette

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
viz = (sys.argv[2] != '0') if len(sys.argv) > 2 else True
tmp = tempfile.mkdtemp()
fdir = os.path.join(tmp, 'frames'); os.makedirs(fdir)
frames, m0 = synth.clip(1, T, 480, 854)
for i in range(T):
    im = Image.fromarray((frames[i].permute(1, 2, 0).numpy() * 255).astype(np.uint8))
    if os.environ.get('FRAME_FMT', 'jpg') == 'png':
        im.save(os.path.join(fdir, f'{i:05d}.png'))
    else:
        im.save(os.path.join(fdir, f'{i:05d}.jpg'), quality=92)
ckpt = os.path.join(tmp, 'ckpt.pth')
torch.save({'epoch': 0, 'model': synth.make_state_dict(20200212), 'loss': 0.0, 'seed': 20200212}, ckpt)
os.chdir(tmp)
os.makedirs('output/segs/clip/mask')
save_seg_mask(m0.numpy(), 'output/segs/clip/mask/00000.png', color_palette)
args = argparse.Namespace(gpu=0, budget=250000, viz=viz, model_path=ckpt, update_rate=0.1, merge_thres=0.95,
                          test_path=fdir, test_name='clip', decode=os.environ.get('DECODE', 'device'))
dev = torch.device('cuda', 0)
video_seg.main(argparse.Namespace(**{**vars(args), 'test_name': 'warm'}) if False else args, dev)   # warm-up (plans, tables, page cache)
# time the frame loop from iteration SKIP+1 to the end (files flushed), so that model construction, checkpoint loading
# and the DataLoader workers' start-up are excluded; with the defaults these are frames 5-99 of the C2 clip
marks = {'n': 0}
SKIP = int(sys.argv[3]) if len(sys.argv) > 3 else 4      # loop iterations before the clock starts (DataLoader start-up)
orig_step = video_seg.ClipRunner.launch
def step(self, *a, **k):
    marks['n'] += 1
    if marks['n'] == SKIP + 1:
        torch.cuda.synchronize(); marks['t0'] = time.perf_counter()
    return orig_step(self, *a, **k)
video_seg.ClipRunner.launch = step
video_seg.main(args, dev)
dt = time.perf_counter() - marks['t0']
T = T - SKIP
print('main() frame loop: %d frames, viz=%s: %.1f frames/s end to end, files on disk (%.1f ms/frame)' % (T - 1, viz, (T - 1) / dt, 1e3 * dt / (T - 1)))
