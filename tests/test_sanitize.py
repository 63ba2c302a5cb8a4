"""Sanitized CPU build of the library's host-only C++ (`make -C v-floodnet_amd/csrc asan`).

``vfn_jpeg_entropy_decode`` walks user-supplied files in DataLoader workers; ``vfn_postprocess_pred_u8`` is the host
connected-component filter.  Both are compiled with AddressSanitizer + UBSan (``-fno-sanitize-recover``) into a fuzz-style
driver that feeds the decoder the 12 JPEG cases of tests/test_jpeg.py, every truncation of their headers, every single-bit
flip of their DHT / DQT / SOF / SOS / DRI segments and pseudo-random damage to the entropy-coded data, each from an
exactly-sized heap copy.  Any return code is fine for a damaged file; a sanitizer report (non-zero exit) is not.
Also: damaged files reach Python as RuntimeError through the product library (no crash, no garbage tables)."""
import os
import subprocess

import pytest

from test_jpeg import CASES, _jpeg_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'v-floodnet_amd', 'csrc')


def test_host_code_under_asan_ubsan(tmp_path):
    subprocess.check_call(['make', '-C', CSRC, 'asan'], stdout=subprocess.DEVNULL)
    files = []
    for case in CASES:
        p = tmp_path / (case[0] + '.jpg')
        p.write_bytes(_jpeg_bytes(*case))
        files.append(str(p))
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([os.path.join(CSRC, 'build_asan', 'vfn_host_fuzz')] + files, capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'no sanitizer report' in r.stdout
    cases = int(r.stdout.split(':')[1].split('cases')[0])
    assert cases > 20000, r.stdout


def _damaged(data):
    """Truncations at and inside every marker segment + short-table files (the ADVICE round-2 cases)."""
    out = []
    p = 2
    while p + 4 <= len(data) and data[p] == 0xFF:
        m, ln = data[p + 1], (data[p + 2] << 8) | data[p + 3]
        for cut in (p, p + 1, p + 2, p + 3, p + 4, p + 2 + ln // 2, p + 1 + ln):
            out.append(data[:cut])
        if m == 0xDB:                                   # DQT whose length field claims less than one table
            out.append(data[:p + 2] + bytes([0, 10]) + data[p + 4:p + 12])
        if m == 0xC4:                                   # DHT cut inside the 16 counts, at end of file
            out.append(data[:p + 2] + bytes([0, 9]) + data[p + 4:p + 11])
        if m == 0xDA:
            break
        p += 2 + ln
    out.append(data[:p] + b'\xff')                      # fill byte as the very last byte
    return out


@pytest.mark.parametrize('case', [CASES[1], CASES[4], CASES[9]], ids=lambda c: c[0])
def test_truncated_segments_raise_in_the_product_library(case):
    import vfloodnet_amd  # noqa: F401
    from vfloodnet_amd import jpeg_device
    data = _jpeg_bytes(*case)
    jpeg_device.entropy_decode(data)
    for d in _damaged(data):
        with pytest.raises(RuntimeError):
            jpeg_device.entropy_decode(d)
