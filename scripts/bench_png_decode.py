"""Input-side PNG: host inflate, PIL's full decode, and the device scanline reconstruction + ToTensor, per frame."""
import sys, io, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, vfloodnet_amd
from vfloodnet_amd import png_decode
from PIL import Image
from tools import synth
dev = torch.device('cuda', 0)
for (H, W) in [(480, 854), (1080, 1920)]:
    frames, _ = synth.clip(1, 1, H, W)
    a = (frames[0].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format='PNG'); data = buf.getvalue()
    t0 = time.perf_counter()
    for _ in range(10): filtered, info, pal = png_decode.inflate(data)
    t_inf = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10): np.array(Image.open(io.BytesIO(data)).convert('RGB'))
    t_pil = (time.perf_counter() - t0) / 10
    fd = torch.from_numpy(filtered).to(dev)
    for _ in range(3): png_decode.to_tensor(fd, info, pal, dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): png_decode.to_tensor(fd, info, pal, dev)
    e1.record(); torch.cuda.synchronize()
    print('%dx%d RGB PNG (%d KB): host inflate %.2f ms, PIL full decode %.2f ms, device unfilter + ToTensor %.3f ms (one CU)' % (
        H, W, len(data) // 1024, t_inf * 1e3, t_pil * 1e3, e0.elapsed_time(e1) / 20))
