"""ctypes binding of ``libvfn_hip.so`` (the C ABI declared in ``include/vfn_hip.h``).

The library is the product: if it is missing, or a launcher reports a non-zero
status, this module raises ``RuntimeError`` -- there is no eager / CPU fallback.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VFN_LIB_PATH') or os.path.join(_HERE, 'libvfn_hip.so')     # (override: ablation builds, scripts/)

_lib = None

c_fp = C.c_void_p          # device pointers travel as void*


class ConvDesc(C.Structure):
    _fields_ = [('inp', c_fp), ('w', c_fp), ('scale', c_fp), ('shift', c_fp), ('res', c_fp), ('out', c_fp),
                ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('in_ld', C.c_int),
                ('Ho', C.c_int), ('Wo', C.c_int), ('Cout', C.c_int), ('cout_pad', C.c_int),
                ('out_ld', C.c_int), ('res_ld', C.c_int),
                ('KH', C.c_int), ('KW', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
                ('relu_in', C.c_int), ('relu_out', C.c_int), ('M', C.c_int), ('ksplit', C.c_int),
                ('split_from', C.c_int), ('res_mod', C.c_int), ('partial', c_fp), ('tile_counters', c_fp),
                ('w_packed', C.c_int), ('in_lp', C.c_int), ('out_lp_relu', C.c_int), ('out_lp', c_fp), ('mask', c_fp), ('mask_ld', C.c_int), ('mask_after', C.c_int), ('w_batch_rows', C.c_int), ('k_rot', C.c_int)]


class WgradDesc(C.Structure):
    _fields_ = [('x', c_fp), ('gy', c_fp), ('rowscale', c_fp), ('dw', c_fp), ('partial', c_fp),
                ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('ld_x', C.c_int),
                ('Ho', C.c_int), ('Wo', C.c_int), ('Cout', C.c_int), ('ld_g', C.c_int),
                ('k', C.c_int), ('stride', C.c_int), ('pad', C.c_int), ('relu', C.c_int), ('accumulate', C.c_int), ('ksplit', C.c_int),
                ('tile_counters', c_fp), ('batch', C.c_int), ('reserved', C.c_int), ('x_bstride', C.c_longlong), ('g_bstride', C.c_longlong)]


class RefreshFilter(C.Structure):
    _fields_ = [('src', c_fp), ('dst', c_fp), ('gamma', c_fp), ('var', c_fp), ('eps', C.c_float), ('kind', C.c_int),
                ('cout', C.c_int), ('cin', C.c_int), ('cin_off', C.c_int), ('cin_total', C.c_int), ('kh', C.c_int), ('kw', C.c_int),
                ('dst_ld', C.c_int), ('dst_row0', C.c_int), ('dst_col0', C.c_int), ('cout_ld', C.c_int), ('block0', C.c_int),
                ('reserved', C.c_int)]


class RefreshEpilogue(C.Structure):
    _fields_ = [('gamma', c_fp), ('beta', c_fp), ('mean', c_fp), ('var', c_fp), ('scale', c_fp), ('shift', c_fp),
                ('eps', C.c_float), ('C', C.c_int)]


class GatherEntry(C.Structure):
    _fields_ = [('src', c_fp), ('dst_offset', C.c_longlong), ('stride', C.c_longlong * 4), ('shape', C.c_int * 4), ('block0', C.c_int),
                ('reserved', C.c_int)]


class StemDesc(C.Structure):
    _fields_ = [('frame', c_fp), ('mask', c_fp), ('w', c_fp), ('scale', c_fp), ('shift', c_fp), ('out', c_fp),
                ('mean', C.c_float * 3), ('std', C.c_float * 3),
                ('N', C.c_int), ('cin', C.c_int), ('H0', C.c_int), ('W0', C.c_int),
                ('pad_top', C.c_int), ('pad_left', C.c_int), ('Hp', C.c_int), ('Wp', C.c_int),
                ('Ho', C.c_int), ('Wo', C.c_int)]


class BankScanDesc(C.Structure):
    _fields_ = [('q', c_fp), ('bank_k', c_fp), ('bank_len', c_fp), ('rowscale', c_fp), ('part', c_fp),
                ('stride_q', C.c_longlong), ('stride_k', C.c_longlong), ('stride_rs', C.c_longlong),
                ('scale', C.c_float),
                ('ldq', C.c_int), ('q_per_obj', C.c_int), ('HW', C.c_int), ('obj_n', C.c_int),
                ('nsplit', C.c_int), ('mode', C.c_int), ('precision', C.c_int), ('work_counter', c_fp), ('bank_k_lp', c_fp), ('scores', c_fp), ('stride_scores', C.c_longlong)]


class MemReadDesc(C.Structure):
    _fields_ = [('q', c_fp), ('qv', c_fp), ('bank_k', c_fp), ('bank_v', c_fp), ('bank_len', c_fp), ('ml', c_fp),
                ('o_part', c_fp), ('cnt', c_fp), ('info', c_fp), ('out', c_fp),
                ('stride_k', C.c_longlong), ('stride_v', C.c_longlong), ('stride_cnt', C.c_longlong),
                ('stride_info', C.c_longlong),
                ('scale', C.c_float), ('thres', C.c_float),
                ('ldq', C.c_int), ('ldqv', C.c_int), ('ld_out', C.c_int), ('HW', C.c_int), ('obj_n', C.c_int),
                ('nsplit', C.c_int), ('precision', C.c_int), ('bank_k_lp', c_fp), ('bank_v_lp', c_fp), ('scores', c_fp), ('stride_scores', C.c_longlong)]


class BankDesc(C.Structure):
    _fields_ = [('bank_k', c_fp), ('bank_v', c_fp), ('info', c_fp),
                ('scratch_k', c_fp), ('scratch_v', c_fp), ('scratch_info', c_fp),
                ('bank_len', c_fp), ('bank_len_rw', c_fp), ('bank_knorm', c_fp), ('bank_vnorm', c_fp),
                ('match_idx', c_fp), ('match_corr', c_fp), ('new_k', c_fp), ('new_knorm', c_fp), ('new_vnorm', c_fp),
                ('app_pos', c_fp), ('keep_dst', c_fp), ('plan', c_fp), ('stats', c_fp),
                ('stride_k', C.c_longlong), ('stride_v', C.c_longlong), ('stride_info', C.c_longlong),
                ('stride_n', C.c_longlong), ('stride_new', C.c_longlong),
                ('class_budget', C.c_double),
                ('thres_close', C.c_float), ('update_rate', C.c_float), ('new_hit_init', C.c_float),
                ('frame_idx', C.c_int), ('ld_new', C.c_int), ('voff', C.c_int), ('HW', C.c_int),
                ('obj_n', C.c_int), ('cap', C.c_int), ('rm_class', C.c_int), ('rm_request', C.c_int)]


ABI_VERSION = 12         # include/vfn_hip.h VFN_ABI_VERSION; csrc/abi.hip
DESC_IDS = {0: ConvDesc, 1: StemDesc, 2: BankScanDesc, 3: MemReadDesc, 4: BankDesc, 5: WgradDesc, 6: RefreshFilter, 7: RefreshEpilogue, 8: GatherEntry}     # vfn_sizeof_desc(which)


def lib():
    """Load the shared library once; raise loudly if it is absent or stale."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build the HIP kernels first '
                '(python -c "import __graft_entry__ as g; g.build()" or make -C v-floodnet_amd/csrc)')
        L = C.CDLL(LIB_PATH)
        L.vfn_abi_version.restype = C.c_int
        if L.vfn_abi_version() != ABI_VERSION:
            raise RuntimeError(f'{LIB_PATH} has ABI version {L.vfn_abi_version()}, this package expects {ABI_VERSION} '
                               '(descriptor layouts differ): rebuild it (make -C v-floodnet_amd/csrc)')
        L.vfn_sizeof_desc.restype = C.c_int
        for which, cls in DESC_IDS.items():
            if L.vfn_sizeof_desc(which) != C.sizeof(cls):
                raise RuntimeError(f'{LIB_PATH}: sizeof({cls.__name__}) is {L.vfn_sizeof_desc(which)} in the library, '
                                   f'{C.sizeof(cls)} in this binding: rebuild it (make -C v-floodnet_amd/csrc)')
        _declare(L)
        _lib = L
    return _lib


def _declare(L):
    i, f, p = C.c_int, C.c_float, C.c_void_p
    L.vfn_abi_version.restype = i
    L.vfn_conv_cfg_count.restype = i
    L.vfn_conv_cfg_tile.argtypes = [i, C.POINTER(i), C.POINTER(i)]
    L.vfn_conv_cfg_info.argtypes = [i] + [C.POINTER(i)] * 5
    L.vfn_conv_cfg_wk.argtypes = [i]
    L.vfn_conv_cfg_wk.restype = i
    L.vfn_conv_cfg_tpb.argtypes = [i]
    L.vfn_conv_cfg_tpb.restype = i
    L.vfn_conv_cfg_kind.argtypes = [i]
    L.vfn_conv_cfg_name.argtypes = [i, C.c_char_p, i]
    L.vfn_conv_cfg_name.restype = i
    L.vfn_conv_cfg_kind.restype = i
    L.vfn_sizeof_desc.argtypes = [i]
    L.vfn_conv2d_nhwc_f32.argtypes = [C.POINTER(ConvDesc), i, p]
    L.vfn_conv2d_nhwc_bf16.argtypes = [C.POINTER(ConvDesc), i, p]
    L.vfn_conv2d_nhwc_bf16x3.argtypes = [C.POINTER(ConvDesc), i, p]
    L.vfn_stem_conv7x7_f32.argtypes = [C.POINTER(StemDesc), p]
    L.vfn_bank_scan.argtypes = [C.POINTER(BankScanDesc), p]
    L.vfn_memread_apply.argtypes = [C.POINTER(MemReadDesc), p]
    L.vfn_memread_finish.argtypes = [C.POINTER(MemReadDesc), p]
    L.vfn_bank_merge.argtypes = [C.POINTER(BankDesc), p]
    L.vfn_bank_append.argtypes = [C.POINTER(BankDesc), p]
    L.vfn_bank_remove.argtypes = [C.POINTER(BankDesc), p]
    L.vfn_bank_refresh_norms.argtypes = [C.POINTER(BankDesc), p, p, p, p]
    L.vfn_conv_wgrad_f32.argtypes = [C.POINTER(WgradDesc), p]
    L.vfn_bank_refresh_lp.argtypes = [C.POINTER(BankDesc), p, p, i, p]
    L.vfn_stem_wgrad_scratch_floats.argtypes = [i]
    L.vfn_stem_wgrad_scratch_floats.restype = i
    for name, args in SIGNATURES.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = i


# name -> argtypes for the plain-argument launchers (kept next to the header order)
_i, _f, _p = C.c_int, C.c_float, C.c_void_p
_ll = C.c_longlong
SIGNATURES = {
    'vfn_maxpool3x3s2_nhwc_f32': [_p, _p, _i, _i, _i, _i, _p],
    'vfn_upsample2x_add_nhwc_f32': [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    'vfn_upsample2x_add_lp_nhwc_f32': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    'vfn_rough_uncertainty_f32': [_p, _p, _p, _p, _i, _i, _i, _p],
    'vfn_transpose_taps_f32': [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p],
    'vfn_dilate2_f32': [_p, _p, _i, _i, _i, _i, _i, _i, _p],
    'vfn_bn_param_grads_f32': [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _p, _p],
    'vfn_bn_param_grads_acc_f32': [_p, _p, _p, _p, _p, _i, _i, _p, _i, _p, _p, _i, _p, _p],
    'vfn_maxpool3x3s2_backward_f32': [_p, _p, _p, _i, _i, _i, _i, _p, _i, _p],
    'vfn_softmax_cols_f32': [_p, _i, _i, _i, _f, _p, _p],
    'vfn_softmax_cols_backward_f32': [_p, _p, _i, _i, _i, _f, _p, _p],
    'vfn_adamw_f32': [_p, _p, _p, _p, _ll, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _i, _p],
    'vfn_colsum_f32': [_p, _i, _i, _i, _p, _i, _p, _p],
    'vfn_colsum_acc_f32': [_p, _i, _i, _i, _p, _i, _p, _i, _p, _p],
    'vfn_upsample2x_add_backward_f32': [_p, _p, _p, _i, _i, _i, _i, _i, _p],
    'vfn_tail_grad_o_f32': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'vfn_segment_loss_f32': [_p, _p, _i, _i, _i, _f, _p, _p, _p, _p],
    'vfn_segment_uncertainty_backward_f32': [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    'vfn_tail_split_f32': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    'vfn_local_stats_backward_f32': [_p] * 13 + [_i, _i, _i, _i, _p],
    'vfn_local_hpass_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'vfn_local_vpass_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    'vfn_local_stats_f32': [_p, _p, _p, _p, _i, _i, _i, _i, _p],
    'vfn_pred2_gather_f32': [_p, _p, _p, _i, _i, _i, _i, _p],
    'vfn_final_logits_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'vfn_bank_scan_finish': [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p],
    'vfn_row_norms': [_p, _ll, _i, _i, _p, _i, _i, _p, _p, _ll, _p],
    'vfn_scatter_mean_f32': [_p, _ll, _ll, _p, _i, _p, _ll, _ll, _i, _p],
    'vfn_winograd_tiles': [_i, _i, _i],
    'vfn_winograd_input_f32': [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    'vfn_winograd_output_f32': [_p, _i, _i, _i, _i, _i, _p, _p, _p, _i, _i, _i, _p, _i, _p],
    'vfn_winograd_gy_f32': [_p, _i, _i, _i, _i, _i, _p, _i, _p],
    'vfn_winograd_gemm_f32': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'vfn_conv1x1_persistent_f32': [_p, _i, _i, _p],
    'vfn_winograd_input_bf16': [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p],
    'vfn_winograd_gemm_bf16': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'vfn_winograd_dw_f32': [_p, _i, _i, _p, _p, _i, _p],
    'vfn_winograd_output_masked_f32': [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _i, _p, _i, _p],
    'vfn_scatter_mean_checked_f32': [_p, _ll, _ll, _p, _ll, _i, _p, _ll, _ll, _i, _ll, _p, _p],
    'vfn_resize_bicubic_f32': [_p, _p, _i, _i, _i, _i, _i, _p],
    'vfn_resize_nearest_f32': [_p, _p, _i, _i, _i, _i, _i, _p],
    'vfn_softmax_objects_f32': [_p, _p, _i, _i, _p],
    'vfn_resize_argmax_u8': [_p, _p, _i, _i, _i, _i, _i, _p],
    'vfn_postprocess_pred_u8': [_p, _i, _i, _p],
    'vfn_postprocess_pred_device_u8': [_p, _p, _p, _i, _i, _p],
    'vfn_to_tensor_u8': [_p, _p, _i, _i, _p],
    'vfn_overlay_u8': [_p, _p, _p, _p, _p, _i, _i, C.c_double, C.c_double, _p],
    'vfn_segment_uncertainty_f32': [_p, _i, _i, _i, _p, _p, _p],
    'vfn_jpeg_entropy_decode': [_p, _ll, _p, _ll, _p, _p],
    'vfn_jpeg_idct_u8': [_p, _p, _p, _i, _i, _i, _p],
    'vfn_jpeg_to_tensor_f32': [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    'vfn_png_sizes': [_i, _i, _i, C.POINTER(_ll), C.POINTER(_ll)],
    'vfn_png_deflate_u8': [_p, _i, _i, _i, _p, _p, _p, _p],
    'vfn_png_unfilter_sizes': [_i, _i, _i, C.POINTER(_i), C.POINTER(_ll)],
    'vfn_png_unfilter_u8': [_p, _i, _i, _i, _p, _p, _p, _p],
    'vfn_png_to_tensor_f32': [_p, _i, _i, _i, _i, _p, _p, _p, _p],
    'vfn_refresh_elems_per_block': [],
    'vfn_refresh_filters_f32': [_p, _i, _i, _p],
    'vfn_refresh_epilogues_f32': [_p, _i, _p],
    'vfn_gather_strided_f32': [_p, _i, _i, _p, _p],
    'vfn_ln_stem_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    'vfn_ln_dwconv_f32': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p],
    'vfn_ln_se_gate_f32': [_p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    'vfn_ln_scale_cols_f32': [_p, _p, _p, _i, _i, _p],
    'vfn_ln_add_f32': [_p, _p, _p, _ll, _p],
    'vfn_ln_head_f32': [_p, _p, _f, _p, _ll, _i, _i, _i, _p],
    'vfn_stem_wgrad_f32': [_p, _p, _p, _p, _p, _ll, _i, _i, _i, _i, _i, _i, _i, _p],
}
# every symbol include/vfn_hip.h declares (checked by tests/test_abi.py)
ALL_SYMBOLS = sorted(list(SIGNATURES) + [
    'vfn_abi_version', 'vfn_sizeof_desc', 'vfn_conv_cfg_count', 'vfn_conv_cfg_tile', 'vfn_conv_cfg_info', 'vfn_conv_cfg_wk', 'vfn_conv_cfg_tpb', 'vfn_conv_cfg_kind', 'vfn_conv_cfg_name', 'vfn_conv2d_nhwc_f32', 'vfn_conv2d_nhwc_bf16', 'vfn_conv2d_nhwc_bf16x3',
    'vfn_stem_conv7x7_f32',
    'vfn_bank_scan', 'vfn_memread_apply', 'vfn_memread_finish', 'vfn_bank_merge', 'vfn_bank_append', 'vfn_bank_remove', 'vfn_bank_refresh_norms', 'vfn_bank_refresh_lp', 'vfn_conv_wgrad_f32',
    'vfn_stem_wgrad_scratch_floats'])


def check(status, what):
    if status != 0:
        raise RuntimeError(f'HIP kernel launcher {what} failed with status {status}')


def ptr(t):
    """Device pointer of a tensor (or None)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """hipStream_t of torch's current stream on the current device (one C call: this runs before every kernel launch)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(t, what='tensor'):
    if not t.is_cuda:
        raise RuntimeError(f'{what} must live on the GPU: the V-FloodNet hot path has no CPU fallback')


# ---------------------------------------------------------------------------------------------- streams that really run beside each other
# A HIP stream is not a hardware queue.  The runtime multiplexes every stream of a priority onto a few hardware queues
# (GPU_MAX_HW_QUEUES, default 4), and PyTorch creates its whole pool of 32 streams per priority on the first
# ``torch.cuda.Stream()``: streams that share a hardware queue execute IN ORDER, whatever their events say.  Round 6 found the
# PNG sink's stream on the queue of the frame loop's stream: its five latency-bound kernels per image ran between two frames
# instead of underneath the next one -- 0.85 ms of an idle matrix pipe per frame in ``video_seg.main`` (median gap between a
# frame's last marker and the next frame's first: 0.855 ms; 0.012 ms with 24 hardware queues, which is no cure: the command
# processor then time-slices and the loop runs at 76 frames/s).  So side streams are PICKED: a candidate is kept only if a tiny
# kernel on it finishes while every stream in ``beside`` is still busy with a few milliseconds of queued work.
PROBES = []          # (found, candidates tried) of every independent_stream call of this process (bench.py reports them)


def independent_stream(device, beside=(), tries=12, priority=0):
    """A ``torch.cuda.Stream`` whose hardware queue is not the one of the current stream nor of any stream in ``beside``
    (measured, see above).  Falls back to the last candidate if none qualifies within ``tries`` (the loop then still works,
    serialised as before)."""
    device = torch.device(device)
    cur = torch.cuda.current_stream(device)
    others = [cur] + [s_ for s_ in beside if s_ is not None]
    big, small = torch.zeros(16 * 1024 * 1024, device=device), torch.zeros(16, device=device)     # (64 MB, returned to the allocator at exit)

    def overlaps(cand, ref):
        torch.cuda.synchronize(device)
        ref_end, cand_end = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(ref):
            for _ in range(64):                     # ~2 ms of bandwidth-bound work queued on ``ref``
                big.add_(1.0)
            ref_end.record()
        with torch.cuda.stream(cand):
            small.add_(1.0)
            cand_end.record()
        cand_end.synchronize()
        ok = not ref_end.query()                    # the candidate's kernel finished while ``ref`` was still busy
        torch.cuda.synchronize(device)
        return ok

    cand = None
    for n_ in range(tries):
        cand = torch.cuda.Stream(device=device, priority=priority)
        if all(overlaps(cand, o_) for o_ in others):
            PROBES.append((True, n_ + 1))
            return cand
    PROBES.append((False, tries))
    import warnings
    warnings.warn('vfloodnet_amd: no stream with a hardware queue of its own found; side-stream work will run in order with the frame loop')
    return cand
