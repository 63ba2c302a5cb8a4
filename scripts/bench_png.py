"""Device PNG encoder: time per image (kernels only, and with the D2H of the stream) for a 480p mask and overlay."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, vfloodnet_amd
from vfloodnet_amd import png_device, _lib
from vfloodnet_amd._lib import ptr, stream
from tools import synth

dev = torch.device('cuda', 0)
frames, m0 = synth.clip(1, 1, 480, 854)
rgb = (frames[0].permute(1, 2, 0) * 255).to(torch.uint8).contiguous().to(dev)
mask = m0.to(dev)
for name, img in (('mask', mask), ('overlay', rgb)):
    bpp = 1 if img.dim() == 2 else 3
    enc = png_device.encoder_for(img.shape[0], img.shape[1], bpp, dev)
    sl = enc._slots[0]
    L = _lib.lib()
    for _ in range(3):
        L.vfn_png_deflate_u8(ptr(img), enc.H, enc.W, bpp, ptr(enc._work), ptr(sl['out']), ptr(sl['stats']), stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(50):
        L.vfn_png_deflate_u8(ptr(img), enc.H, enc.W, bpp, ptr(enc._work), ptr(sl['out']), ptr(sl['stats']), stream())
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us device per image, host enqueue {1e6 * (t1 - t0) / 50:.1f} us; '
          f'{int(sl["stats"][0])} bytes')
    t0 = time.perf_counter()
    for _ in range(50):
        data = enc.finish(enc.encode(img), [0, 0, 0, 255, 255, 255])
    print(f'   encode+finish round trip {1e3 * (time.perf_counter() - t0) / 50:.2f} ms, file {len(data)} B')
