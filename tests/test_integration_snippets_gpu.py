"""INTEGRATION.md section 2 shows the reference-side ctypes bindings a maintainer would write.  This test executes those
snippets VERBATIM (own ctypes structs, no vfloodnet_amd._lib; only '/path/to/libvfn_hip.so' is substituted) and
checks the results against torch: a wrong struct layout in the document fails here."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'v-floodnet_amd', 'libvfn_hip.so')


def _snippets():
    txt = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    sec = txt[txt.index('## 2. Bind the kernels directly'):txt.index('## 3. Multi-GPU')]
    blocks = re.findall(r'```python\n(.*?)```', sec, flags=re.S)
    assert len(blocks) == 2, 'INTEGRATION.md section 2 is expected to hold the scatter and the conv snippet'
    return [b.replace('/path/to/libvfn_hip.so', LIB) for b in blocks]


@pytest.mark.gpu
def test_scatter_snippet(gpu):
    ns = {}
    exec(_snippets()[0], ns)
    g = torch.Generator().manual_seed(3)
    D, S, B = 128, 300, 57
    src = torch.randn(D, S, generator=g).to(gpu)
    idx_row = torch.randint(0, B, (S,), generator=g)
    index = idx_row.unsqueeze(0).expand(D, S).to(gpu)
    out = torch.zeros(D, B, device=gpu)
    ns['scatter_mean'](src, index, dim=1, out=out)
    torch.cuda.synchronize()
    ref = torch.zeros(D, B)
    ref.scatter_add_(1, idx_row.unsqueeze(0).expand(D, S), src.cpu())
    cnt = torch.bincount(idx_row, minlength=B).clamp(min=1).float()
    assert (out.cpu() - ref / cnt).abs().max() < 1e-5


@pytest.mark.gpu
def test_conv_snippet(gpu):
    import torch.nn.functional as F
    s0, s1 = _snippets()
    g = torch.Generator().manual_seed(4)
    N, H, W, Cin, Cout = 2, 24, 40, 64, 96
    Ho, Wo = H, W
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24.0
    gamma, beta = 1 + 0.1 * torch.randn(Cout, generator=g), 0.1 * torch.randn(Cout, generator=g)
    mean, var = 0.1 * torch.randn(Cout, generator=g), 0.5 + torch.rand(Cout, generator=g)
    # the two lines of prose under the snippet
    w_packed = torch.zeros(256, 9 * Cin)
    w_packed[:Cout] = w.permute(0, 2, 3, 1).reshape(Cout, -1)
    scale = gamma / torch.sqrt(var + 1e-5)
    shift = beta - mean * scale
    ns = dict(x_nhwc=x.permute(0, 2, 3, 1).contiguous().to(gpu), w_packed=w_packed.to(gpu), scale=scale.to(gpu),
              shift=shift.to(gpu), y_nhwc=torch.empty(N, Ho, Wo, Cout, device=gpu), N=N, H=H, W=W, Cin=Cin, Cout=Cout,
              Ho=Ho, Wo=Wo)
    exec(s0, ns)                                           # defines _vfn (the conv snippet continues the same session)
    exec(s1, ns)
    torch.cuda.synchronize()
    ref = F.relu(F.batch_norm(F.conv2d(x.double(), w.double(), padding=1), mean.double(), var.double(), gamma.double(),
                              beta.double(), False, 0.0, 1e-5))
    got = ns['y_nhwc'].cpu().permute(0, 3, 1, 2).double()
    assert (got - ref).abs().max() < 2e-4 * ref.abs().max()


def test_snippets_parse_and_name_the_current_abi():
    """CPU part: the snippets compile, the struct in the document has the header's fields in the header's order."""
    import ctypes
    s0, s1 = _snippets()
    compile(s0, 'INTEGRATION.md#scatter', 'exec')
    compile(s1, 'INTEGRATION.md#conv', 'exec')
    hdr = open(os.path.join(ROOT, 'include', 'vfn_hip.h')).read()
    body = hdr[hdr.index('typedef struct vfn_conv_desc {'):hdr.index('} vfn_conv_desc;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    fields = []
    for decl in body.split('{', 1)[1].split(';'):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.replace('*', ' ').split(None, 1 if not decl.startswith('const') else 2)[-1]
        fields += [n.strip() for n in names.split(',')]
    doc_fields = re.findall(r"'(\w+)'", s1[:s1.index('_vfn.vfn_sizeof_desc')])
    assert [f.rstrip('_') for f in doc_fields] == fields
    abi = re.search(r'#define VFN_ABI_VERSION (\d+)', hdr).group(1)
    assert 'vfn_abi_version() == ' + abi in s1
