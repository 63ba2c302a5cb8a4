"""PNG input side: host inflate + device scanline reconstruction / pixel conversion / ToTensor against PIL, which is what
the reference's Video_DS hands to the model for a PNG frame (Water_DS.py:105-109, myutils/data.py:87-90).

CPU: the product's chunk walker + inflate feed the numpy restatement of the PNG filters (oracle/png_ref.py), which must
reproduce PIL's RGB output exactly -- that pins both.  GPU: the device kernels against PIL, bit for bit, including files
written with every filter type forced on every row."""
import io
import struct
import zlib

import numpy as np
import pytest
import torch
from PIL import Image


def _img(H, W, seed):
    from tools import synth
    frames, _ = synth.clip(seed, 1, max(H, 8), max(W, 8))
    a = (frames[0, :, :H, :W].permute(1, 2, 0).numpy() * 255).astype(np.uint8)
    a[::3, ::5] = np.random.RandomState(seed).randint(0, 256, a[::3, ::5].shape)     # texture: exercises every predictor
    return a


def _pil_png(arr, mode, **kw):
    im = Image.fromarray(arr)
    if mode == 'L':
        im = im.convert('L')
    elif mode == 'LA':
        im = im.convert('LA')
    elif mode == 'RGBA':
        im = im.convert('RGBA')
        a = np.array(im); a[..., 3] = (a[..., 0] // 2 + 64); im = Image.fromarray(a, 'RGBA')
    elif mode == 'P':
        im = im.convert('P', palette=Image.ADAPTIVE, colors=200)
    buf = io.BytesIO()
    im.save(buf, format='PNG', **kw)
    return buf.getvalue()


def _chunk(typ, body):
    return struct.pack('>I', len(body)) + typ + body + struct.pack('>I', zlib.crc32(typ + body) & 0xffffffff)


def _forced_filter_png(arr, ftypes):
    """A PNG whose row r uses filter type ftypes[r % len(ftypes)] (encoder written here: PIL picks filters itself)."""
    H, W, C = arr.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[C]
    raw = arr.reshape(H, W * C).astype(np.int32)
    out = bytearray()
    prev = np.zeros(W * C, np.int32)
    for r in range(H):
        ft = ftypes[r % len(ftypes)]
        cur = raw[r]
        a = np.concatenate([np.zeros(C, np.int32), cur[:-C]])
        c = np.concatenate([np.zeros(C, np.int32), prev[:-C]])
        if ft == 0: p = 0
        elif ft == 1: p = a
        elif ft == 2: p = prev
        elif ft == 3: p = (a + prev) >> 1
        else:
            pp = a + prev - c
            pa, pb, pc = np.abs(pp - a), np.abs(pp - prev), np.abs(pp - c)
            p = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        out.append(ft)
        out += bytes(((cur - p) & 255).astype(np.uint8))
        prev = cur
    ihdr = struct.pack('>IIBBBBB', W, H, 8, ctype, 0, 0, 0)
    z = zlib.compress(bytes(out), 6)
    idat = b''.join(_chunk(b'IDAT', z[i:i + 4096]) for i in range(0, len(z), 4096))      # several IDAT chunks
    return b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', ihdr) + idat + _chunk(b'IEND', b'')


def _cases():
    c = [('rgb_c2_frame', _pil_png(_img(480, 854, 1), 'RGB')), ('rgb_odd', _pil_png(_img(97, 131, 2), 'RGB')),
         ('rgba', _pil_png(_img(64, 83, 3), 'RGBA')), ('grey', _pil_png(_img(50, 70, 4), 'L')),
         ('grey_alpha', _pil_png(_img(33, 41, 5), 'LA')), ('palette', _pil_png(_img(90, 110, 6), 'P')),
         ('rgb_optimized', _pil_png(_img(40, 60, 7), 'RGB', optimize=True)), ('one_px', _pil_png(_img(1, 1, 8)[:1, :1], 'RGB')),
         ('w3', _pil_png(_img(9, 3, 9)[:, :3], 'RGB'))]
    for C in (1, 2, 3, 4):
        a = _img(37, 45, 10 + C)
        a = a[:, :, :C] if C <= 3 else np.concatenate([a, a[:, :, :1] // 2], 2)
        c.append((f'forced_all_filters_{C}ch', _forced_filter_png(np.ascontiguousarray(a), [0, 1, 2, 3, 4, 4, 3, 1])))
        for ft in (1, 3, 4):
            c.append((f'forced_filter{ft}_{C}ch', _forced_filter_png(np.ascontiguousarray(a), [ft])))
    return c


CASES = _cases()


def _pil_rgb(data):
    return np.array(Image.open(io.BytesIO(data)).convert('RGB'))          # myutils.load_image_in_PIL (data.py:87-90)


@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_host_inflate_and_oracle_match_pil(case):
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import png_decode
    from oracle import png_ref
    filtered, info, pal = png_decode.inflate(case[1])
    W, H, ctype, bpp = (int(v) for v in info)
    ref = _pil_rgb(case[1])
    assert (W, H) == (ref.shape[1], ref.shape[0])
    got = png_ref.to_rgb(png_ref.unfilter(filtered, W, H, bpp), W, H, ctype, pal)
    assert np.array_equal(got, ref), case[0]


def test_unsupported_and_corrupt_png_fail_loudly():
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import png_decode
    a = _img(20, 30, 1)
    im16 = Image.fromarray((a[:, :, 0].astype(np.uint16) * 257))            # 16-bit grey
    buf = io.BytesIO(); im16.save(buf, format='PNG')
    with pytest.raises(RuntimeError, match='unsupported'):
        png_decode.inflate(buf.getvalue())
    im1 = Image.fromarray(a[:, :, 0] > 128)                                  # 1-bit
    buf = io.BytesIO(); im1.save(buf, format='PNG')
    with pytest.raises(RuntimeError, match='unsupported'):
        png_decode.inflate(buf.getvalue())
    with pytest.raises(RuntimeError, match='not a PNG'):
        png_decode.inflate(b'\xff\xd8\xff\xe0' + b'0' * 64)
    good = _pil_png(a, 'RGB')
    with pytest.raises((RuntimeError, zlib.error)):
        png_decode.inflate(good[:len(good) // 2])


def test_dataset_device_decode_hands_out_inflated_png(tmp_path):
    """Video_DS(decode='device'): a PNG frame comes as {'png': ...}; unsupported variants fall back to PIL's uint8 frame."""
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd.dataset import Video_DS
    a = _img(24, 36, 3)
    p0, p1, p2 = tmp_path / '00000.png', tmp_path / '00001.png', tmp_path / '00002.png'
    Image.fromarray(a).save(p0); Image.fromarray(a).save(p1)
    Image.fromarray((a[:, :, 0].astype(np.uint16) * 257)).save(p2)          # 16-bit: PIL path
    ds = Video_DS([str(p0), str(p1), str(p2)], Image.fromarray(a), Image.fromarray((a[:, :, 0] > 100).astype(np.uint8)), decode='device')
    item, name = ds[0]
    assert name == '00001' and set(item) == {'png'}
    filtered, info, pal = item['png']
    assert filtered.dtype == torch.uint8 and list(info[:2]) == [36, 24]
    item, _ = ds[1]
    assert set(item) == {'u8'} and item['u8'].shape == (24, 36, 3)


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=[c[0] for c in CASES])
def test_device_png_decode_matches_pil(gpu, case):
    from vfloodnet_amd import png_decode
    filtered, info, pal = png_decode.inflate(case[1])
    ref = _pil_rgb(case[1])
    out, u8 = png_decode.to_tensor(filtered, info, pal, gpu, want_u8=True)
    torch.cuda.synchronize()
    assert png_decode.check_status(gpu)
    assert np.array_equal(u8.cpu().numpy(), ref), case[0]
    want = torch.from_numpy(ref).permute(2, 0, 1).float().div(255)            # torchvision ToTensor
    assert torch.equal(out.cpu(), want)


@pytest.mark.gpu
def test_device_png_decode_large_and_corrupt(gpu):
    """1080p RGB (1080 rows, 480 block columns: the widest / tallest shape of the configs) and a corrupt filter byte."""
    from vfloodnet_amd import png_decode
    data = _pil_png(_img(1080, 1920, 21), 'RGB')
    filtered, info, pal = png_decode.inflate(data)
    out, u8 = png_decode.to_tensor(filtered, info, pal, gpu, want_u8=True)
    assert np.array_equal(u8.cpu().numpy(), _pil_rgb(data))
    assert png_decode.check_status(gpu)
    bad = filtered.copy(); bad[(1920 * 3 + 1) * 7] = 9
    png_decode.to_tensor(bad, info, pal, gpu)
    assert not png_decode.check_status(gpu)
    assert png_decode.check_status(gpu)                                      # the flag was cleared


def test_invalid_filter_byte_is_rejected_at_the_frame(tmp_path):
    """A scanline filter type outside 0..4 is caught by ``inflate`` -- in the loader worker, at that frame, before the frame
    is segmented or memorised -- and ``Video_DS`` then hands the file to PIL, whose error reaches the loop there (as in the
    reference, which opens every frame with PIL)."""
    import struct
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import png_decode
    from vfloodnet_amd.dataset import Video_DS
    a = _img(12, 20, 4)
    W, H = 20, 12
    raw = bytearray()
    for y in range(H):
        raw += bytes([0]) + a[y].tobytes()
    raw[5 * (1 + 3 * W)] = 7                                                 # row 5: filter type 7

    def chunk(t, body):
        return struct.pack('>I', len(body)) + t + body + struct.pack('>I', zlib.crc32(t + body) & 0xffffffff)
    data = b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', W, H, 8, 2, 0, 0, 0)) + \
        chunk(b'IDAT', zlib.compress(bytes(raw))) + chunk(b'IEND', b'')
    with pytest.raises(RuntimeError, match='filter type'):
        png_decode.inflate(data)
    good, bad = tmp_path / '00000.png', tmp_path / '00001.png'
    Image.fromarray(a).save(good)
    bad.write_bytes(data)
    ds = Video_DS([str(good), str(bad)], Image.fromarray(a), Image.fromarray((a[:, :, 0] > 100).astype(np.uint8)), decode='device')
    with pytest.raises(Exception):                                          # PIL's own error for the bad file
        ds[0]
