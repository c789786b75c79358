"""ctypes binding of libimgcomp_hip.so (the C ABI declared in include/imgcomp_hip.h).

The HIP library IS the product: there is no CPU or eager-PyTorch fallback anywhere in this
package.  If the shared object is missing or a symbol cannot be resolved, importing this module
raises -- run ``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C
imgcomp_cvpr_amd/csrc``) first.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_longlong, c_size_t, c_uint32, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
# IMGCOMP_HIP_LIB: tuning aid (A/B two builds of the library on one GPU box); default = the in-tree build
LIB_PATH = os.environ.get('IMGCOMP_HIP_LIB') or os.path.join(_HERE, 'libimgcomp_hip.so')


class HipLibraryError(RuntimeError):
    pass


# per-call plan flags (include/imgcomp_hip.h, IC_CONV3_* / IC_EDGE_* / IC_PC_*): the library keeps no process-wide state
CONV3_FORM_MASK = 0x0f
CONV3_AUTO, CONV3_DIRECT, CONV3_WINO, CONV3_WINO_WHOLEK, CONV3_WINO_WHOLEK_PW = 0, 1, 2, 3, 4
CONV3_WINO_KSPLIT, CONV3_WINO_T16, CONV3_WINO_SEG1, CONV3_WINO_SEG2, CONV3_WINO_SEG3 = 5, 6, 7, 8, 9
CONV3_WINO_PAIR = 10
CONV3_WINO4 = 11                    # Winograd F(4x4,3x3)
CONV3_NO_WINO4 = 0x800000
CONV3_WINO4_WG8 = 0x8000000       # F(4x4) kernel: 8-wave work-groups (all 128 channels of a segment)
CONV3_WINO4_WG4 = 0x10000000      # F(4x4) kernel: always the 4-wave form
CONV5_BOTH_PACKED = 0x1000000         # h2 / h12 blobs carry the F(4x4)-over-phases fragments too (ic_pack_conv5s2_both_f32)
CONV5_WINO4 = 0x2000000               # h2 / h12 on the F(4x4) kernel wherever the shape allows
CONV5_NO_WINO4 = 0x4000000
CONV3_LEAVE_IDLE_CUS = 0x10
CONV3_NO_XCD_RUNS = 0x20
CONV3_PACKED_TRANSFORM = 0x40
CONV3_STACK_KERNEL = 0x80


def CONV3_IN_FLIGHT(n):
    return (min(int(n), 15) & 0xf) << 19
PC_DECODE_PER_LAYER = 0x01
PC_DECODE_RECOMPUTE = 0x02


def conv3_leave_idle_layers(n):
    return (n & 0x7f) << 12


def conv3_direct_variant(v):
    return ((v + 1) & 0xf) << 8


def edge_tiles_per_wg(n):
    return n & 0xff


# name -> (restype, argtypes); mirrors include/imgcomp_hip.h one-to-one (tests/test_abi.py checks
# that every ic_* prototype of the header is listed here and exported by the .so).
PROTOTYPES = {
    'ic_abi_version': (c_int, []),
    'ic_strerror': (c_char_p, [c_int]),
    'ic_crc32c': (c_uint32, [c_void_p, c_size_t, c_uint32]),
    'ic_conv2d_bn_act_f32': (c_int, [c_void_p] * 7 + [c_int] * 9 + [c_void_p, c_void_p, c_void_p]),
    'ic_deconv2d_bn_act_f32': (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p, c_void_p, c_int, c_void_p]),
    'ic_conv3x3_c128_packed_floats': (c_size_t, []),
    'ic_pack_conv3x3_c128_f32': (c_int, [c_void_p, c_void_p, c_void_p]),
    'ic_conv3x3_c128_bn_act_f32': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    'ic_wino3x3_c128_packed_floats': (c_size_t, []),
    'ic_pack_wino3x3_c128_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'ic_pack_wino3x3_c128_batch_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'ic_wino3x3_c128_bn_act_f32': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    'ic_wino3x3_c128_workgroups': (c_longlong, [c_int, c_int, c_int, c_int]),
    'ic_wino3x3_c128_plan': (c_int, [c_int, c_int, c_int, c_int, POINTER(c_longlong)]),
    'ic_conv3x3_c128_both_packed_floats': (c_size_t, []),
    'ic_pack_conv3x3_c128_both_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'ic_conv3x3_c128_pick_algo': (c_int, [c_int, c_int, c_int, c_int]),
    'ic_conv3x3_c128_auto_f32': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    'ic_conv2d_mfma_packed_floats': (c_size_t, [c_int] * 6),
    'ic_pack_conv2d_mfma_f32': (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    'ic_conv2d_mfma_bn_act_f32': (c_int, [c_void_p] * 5 + [c_int] * 10 + [c_void_p]),
    'ic_quantize_f32': (c_int, [c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                c_longlong, c_void_p]),
    'ic_heatmap_quantize_f32': (c_int, [c_void_p, c_void_p, c_int, c_float] + [c_void_p] * 6 +
                                [c_int] * 4 + [c_void_p]),
    'ic_pc_workspace_bytes': (c_size_t, [c_int] * 5),
    'ic_pc_packed_floats': (c_size_t, [c_int, c_int]),
    'ic_pc_pack_filters_f32': (c_int, [POINTER(c_void_p), c_int, c_int, c_void_p, c_void_p]),
    'ic_pc_logits_f32': (c_int, [c_void_p, POINTER(c_void_p), c_int, c_int, c_float, c_void_p] +
                         [c_int] * 4 + [c_void_p, c_size_t, c_void_p]),
    'ic_pc_logits_padded_f32': (c_int, [c_void_p, POINTER(c_void_p), c_int, c_int, c_void_p] +
                                [c_int] * 4 + [c_void_p, c_size_t, c_void_p]),
    'ic_pc_bitcost_f32': (c_int, [c_void_p, c_void_p, POINTER(c_void_p), c_int, c_int, c_float,
                                  c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_size_t, c_void_p]),
    'ic_pc_logits_to_freqs_f32': (c_int, [c_void_p, c_longlong, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    'ic_pc_decode_workspace_bytes': (c_size_t, [c_int] * 4),
    'ic_pc_decode_f32': (c_int, [c_void_p, c_longlong, c_int, POINTER(c_void_p), c_void_p, c_int, c_int, c_float,
                                 c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    'ic_sum_f32': (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p]),
    'ic_mean_f32': (c_int, [c_void_p, c_longlong, c_float, c_void_p, c_void_p, c_void_p]),
    'ic_mean_rows_f32': (c_int, [c_void_p, c_int, c_longlong, c_float, c_void_p, c_void_p, c_void_p]),
    'ic_ae_workspace_bytes': (c_size_t, [c_int] * 4),
    'ic_ae_sync_pos_bytes': (c_size_t, [c_int] * 4),
    'ic_ae_res_stack_sync_pos_bytes': (c_size_t, [c_int] * 3),
    'ic_ae_encode_f32': (c_int, [c_void_p, POINTER(c_void_p)] + [c_int] * 5 + [c_void_p] * 6 +
                         [c_int] * 3 + [c_void_p, c_size_t, c_int, c_void_p]),
    'ic_ae_decode_f32': (c_int, [c_void_p, POINTER(c_void_p)] + [c_int] * 3 + [c_void_p] +
                         [c_int] * 3 + [c_void_p, c_size_t, c_int, c_void_p]),
    'ic_ae_res_stack_workspace_bytes': (c_size_t, [c_int] * 3),
    'ic_ae_res_stack_f32': (c_int, [c_void_p, POINTER(c_void_p), c_int, c_void_p] + [c_int] * 3 + [c_void_p, c_size_t, c_int, c_void_p]),
    'ic_peer_region_bytes': (c_size_t, []),
    'ic_peer_max_values': (c_int, []),
    'ic_peer_max_world': (c_int, []),
    'ic_peer_region_create': (c_int, [c_void_p, c_void_p]),
    'ic_peer_region_open': (c_int, [c_void_p, c_void_p]),
    'ic_peer_region_close': (c_int, [c_void_p]),
    'ic_peer_region_destroy': (c_int, [c_void_p]),
    'ic_peer_allreduce_f64': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_uint32, c_void_p, c_void_p]),
    'ic_build_has_tuning_forms': (c_int, []),
    'ic_conv3x3_c128_pick_form': (c_int, [c_int, c_int, c_int, c_int]),
    'ic_wino4_3x3_c128_packed_floats': (c_size_t, []),
    'ic_pack_wino4_3x3_c128_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'ic_pack_wino4_3x3_c128_batch_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'ic_wino4_3x3_c128_supported': (c_int, [c_int, c_int, c_int]),
    'ic_wino4_3x3_c128_workgroups': (c_longlong, [c_int, c_int, c_int]),
    'ic_wino4_3x3_c128_waves': (c_int, [c_int, c_int, c_int, c_int]),
    'ic_wino4_3x3_c128_bn_act_f32': (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_void_p]),
    'ic_wino4_3x3_c128_stats_parts': (c_longlong, [c_int, c_int, c_int]),
    'ic_wino4_3x3_c128_raw_stats_f32': (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    'ic_wino4_conv5s2_packed_floats': (c_size_t, []),
    'ic_pack_wino4_conv5s2_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'ic_wino4_conv5s2_supported': (c_int, [c_int, c_int, c_int]),
    'ic_wino4_conv5s2_workgroups': (c_longlong, [c_int, c_int, c_int, c_int]),
    'ic_wino4_conv5s2_c64_c128_bn_act_f32': (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    'ic_wino4_deconv5s2_c128_c64_bn_act_f32': (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    'ic_val_metrics_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'ic_val_metrics_u8_f64': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ic_val_metrics_per_image_u8_f64': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'ic_space_to_depth2_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'ic_conv5s2_both_packed_floats': (c_size_t, [c_int]),
    'ic_pack_conv5s2_both_f32': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'ic_msssim_plan_bytes': (c_size_t, [c_int, c_int]),
    'ic_msssim_plan_fill': (c_int, [c_int, c_int, c_void_p, c_size_t]),
    'ic_msssim_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'ic_msssim_loss_grad_f32': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_size_t, c_void_p]),
    'ic_peer_allreduce_f64_bounded': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_uint32, c_uint32, c_void_p, c_void_p]),
    'ic_bn_workspace_bytes': (c_size_t, [c_int]),
    'ic_bn_stats_f32': (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p, c_void_p]),
    'ic_bn_train_stats_f32': (c_int, [c_void_p] * 5 + [c_float] * 2 + [c_void_p] * 4 + [c_int] * 3 + [c_void_p, c_void_p]),
    'ic_bn_train_forward_f32': (c_int, [c_void_p] * 5 + [c_float] * 2 + [c_void_p] * 7 + [c_int] * 4 + [c_void_p, c_void_p]),
    'ic_bn_train_forward_cstats_f32': (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 4 + [c_float] * 2 + [c_void_p] * 7 + [c_int] * 4 + [c_void_p]),
    'ic_bn_moments_f32': (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p, c_void_p]),
    'ic_bn_train_fold_moments_f32': (c_int, [c_void_p, c_longlong, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float] +
                                     [c_void_p] * 4 + [c_int, c_void_p]),
    'ic_bn_backward_reduce_f32': (c_int, [c_void_p] * 9 + [c_int] * 4 + [c_void_p, c_void_p]),
    'ic_bn_backward_apply_f32': (c_int, [c_void_p] * 8 + [c_longlong, c_void_p] + [c_int] * 4 + [c_void_p]),
    'ic_bn_apply_f32': (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    'ic_bn_backward_f32': (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_void_p, c_void_p]),
    'ic_conv2d_wgrad_workspace_bytes': (c_size_t, [c_int] * 7),
    'ic_conv2d_wgrad_f32': (c_int, [c_void_p] * 3 + [c_int] * 8 + [c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    'ic_conv3x3_c128_wgrad_workspace_bytes': (c_size_t, [c_int] * 3),
    'ic_conv3x3_c128_wgrad_f32': (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    'ic_pack_conv3x3_c128_bwd_f32': (c_int, [c_void_p, c_void_p, c_void_p]),
    'ic_heatmap_quantize_bwd_workspace_bytes': (c_size_t, [c_int]),
    'ic_heatmap_quantize_bwd_f32': (c_int, [c_void_p, c_void_p, c_int, c_float] + [c_void_p] * 4 + [c_int] * 5 +
                                    [c_void_p, c_void_p]),
    'ic_pc_dlogits_f32': (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_void_p]),
    'ic_pc_bwd_data_workspace_bytes': (c_size_t, [c_int] * 6),
    'ic_pc_bwd_data_f32': (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    'ic_pc_wgrad_workspace_bytes': (c_size_t, [c_int] * 6),
    'ic_pc_wgrad_f32': (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p] + [c_int] * 7 +
                        [c_void_p, c_size_t, c_void_p]),
    'ic_channel_sum_workspace_bytes': (c_size_t, [c_int]),
    'ic_channel_sum_f32': (c_int, [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p, c_void_p]),
    'ic_adam_tf_f32': (c_int, [c_void_p] * 4 + [c_longlong] + [c_float] * 4 + [c_void_p]),
    'ic_stream_create_cu_range': (c_int, [c_int, c_int, POINTER(c_void_p)]),
    'ic_stream_destroy': (c_int, [c_void_p]),
    'ic_event_create': (c_int, [POINTER(c_void_p)]),
    'ic_event_destroy': (c_int, [c_void_p]),
    'ic_event_record': (c_int, [c_void_p, c_void_p]),
    'ic_event_elapsed_ms': (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
}


def _load():
    if not os.path.isfile(LIB_PATH):
        raise HipLibraryError(
            'HIP extension not built: {} is missing.  Build it with '
            '`make -C {}` (hipcc --offload-arch=gfx950).  There is no CPU fallback.'.format(
                LIB_PATH, os.path.join(_HERE, 'csrc')))
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryError('cannot load {}: {}'.format(LIB_PATH, e))
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipLibraryError('{} does not export {}'.format(LIB_PATH, name))
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


def check(rc, what=''):
    if rc != 0:
        msg = lib.ic_strerror(rc)
        raise HipLibraryError('{} failed with code {}: {}'.format(
            what or 'libimgcomp_hip call', rc, msg.decode() if msg else '?'))


def ptr(t):
    """device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), 'C ABI takes dense tensors'
    return c_void_p(t.data_ptr())


def ptr_table(tensors):
    """host array of device pointers (keeps no reference: caller must keep tensors alive)."""
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def current_stream(device=None):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t, name='tensor'):
    if not t.is_cuda:
        raise HipLibraryError(
            '{} lives on {}; the imgcomp hot path runs only on a HIP device (no CPU fallback)'.format(
                name, t.device))
