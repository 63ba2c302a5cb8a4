"""ctypes binding of ``libvfn_hip.so`` (the C ABI declared in ``include/vfn_hip.h``).

The library is the product: if it is missing, or a launcher reports a non-zero
status, this module raises ``RuntimeError`` -- there is no eager / CPU fallback.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libvfn_hip.so')

_lib = None

c_fp = C.c_void_p          # device pointers travel as void*


class ConvDesc(C.Structure):
    _fields_ = [('inp', c_fp), ('w', c_fp), ('scale', c_fp), ('shift', c_fp), ('res', c_fp), ('out', c_fp),
                ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('in_ld', C.c_int),
                ('Ho', C.c_int), ('Wo', C.c_int), ('Cout', C.c_int), ('cout_pad', C.c_int),
                ('out_ld', C.c_int), ('res_ld', C.c_int),
                ('KH', C.c_int), ('KW', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
                ('relu_in', C.c_int), ('relu_out', C.c_int), ('M', C.c_int)]


def lib():
    """Load the shared library once; raise loudly if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build the HIP kernels first '
                '(python -c "import __graft_entry__ as g; g.build()" or make -C v-floodnet_amd/csrc)')
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    i, f, p = C.c_int, C.c_float, C.c_void_p
    L.vfn_abi_version.restype = i
    L.vfn_conv_cfg_count.restype = i
    L.vfn_conv_cfg_tile.argtypes = [i, C.POINTER(i), C.POINTER(i)]
    L.vfn_conv2d_nhwc_f32.argtypes = [C.POINTER(ConvDesc), i, p]
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i


# name -> argtypes for the plain-argument launchers (kept next to the header order)
_i, _f, _p = C.c_int, C.c_float, C.c_void_p
SIGNATURES = {}


def check(status, what):
    if status != 0:
        raise RuntimeError(f'HIP kernel launcher {what} failed with status {status}')


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t, what='tensor'):
    if not t.is_cuda:
        raise RuntimeError(f'{what} must live on the GPU: the V-FloodNet hot path has no CPU fallback')
