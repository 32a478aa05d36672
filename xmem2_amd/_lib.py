"""ctypes binding of the C-ABI library (include/xmem_hip.h).

The product path has NO CPU fallback: if the shared library is missing or fails to load, the first
kernel call raises.  PyTorch is used only for device memory and the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libxmem_hip.so')
_lib = None

c_float_p = C.c_void_p
c_void_p = C.c_void_p


ABI_VERSION = 3          # include/xmem_hip.h XMEM_ABI_VERSION: the layout of ConvDesc below belongs to it


class ConvDesc(C.Structure):
    _fields_ = [('inp', C.c_void_p), ('B', C.c_int), ('H', C.c_int), ('W', C.c_int), ('Cin', C.c_int), ('ldin', C.c_int),
                ('w', C.c_void_p), ('Cout', C.c_int), ('KH', C.c_int), ('KW', C.c_int), ('stride', C.c_int), ('pad', C.c_int),
                ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('res', C.c_void_p), ('ldres', C.c_int),
                ('out', C.c_void_p), ('ldout', C.c_int),
                ('relu_in', C.c_int), ('relu_out', C.c_int), ('plan_tile', C.c_int), ('plan_splitk', C.c_int), ('w_winograd', C.c_void_p),
                ('res_broadcast', C.c_int), ('w_winograd_f16', C.c_void_p), ('w_winograd4', C.c_void_p),
                ('arith', C.c_int), ('w_split', C.c_void_p), ('w_winograd_split', C.c_void_p), ('w_winograd4_split', C.c_void_p),
                ('in_half', C.c_int), ('out_half', C.c_int), ('w_half', C.c_void_p)]


class AugDesc(C.Structure):
    _fields_ = [('type', C.c_int), ('factor', C.c_float), ('image_matrix', C.c_double * 6), ('mask_grid', C.c_float * 6)]


class KeySegment(C.Structure):
    _fields_ = [('key', C.c_void_p), ('shrinkage', C.c_void_p), ('n', C.c_int), ('rows16', C.c_void_p)]


class AffinityHint(C.Structure):
    _fields_ = [('idx', C.c_void_p), ('top_k', C.c_int), ('n_seg', C.c_int), ('seg_n', C.c_int * 4), ('grid_w', C.c_int)]


class ValueSegment(C.Structure):
    _fields_ = [('value', C.c_void_p), ('n', C.c_int)]


_SIGS = {
    'xmem_version': (C.c_int, []),
    'xmem_last_error_string': (C.c_char_p, [C.c_int]),
    'xmem_trace_marker': (C.c_int, [C.c_int, C.c_void_p]),
    'xmem_conv2d_workspace_bytes': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'xmem_conv2d_nhwc': (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_size_t, C.c_void_p]),
    'xmem_maxpool3x3s2': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_maxpool3x3s2_t': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_upsample2x_add_t': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_area_downsample_t': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_copy_channels_t': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_cbam_residual_t': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    'xmem_gru_gate_t': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_upsample2x_add': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_area_downsample': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_copy_channels': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_hidden_update_gather': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p]),
    'xmem_cbam_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'xmem_cbam_residual': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    'xmem_gru_gate': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_add3': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'xmem_pack_image': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_pack_image_u8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p]),
    'xmem_pack_value_input': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_key_post': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    'xmem_logits_to_prob': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_aggregate_masks': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_merge_masks': (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_resize_bilinear': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_argmax_u8': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_nhwc_to_nchw': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_nchw_to_nhwc': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_affinity_topk_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'xmem_affinity_debug_offsets': (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    'xmem_augment_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    'xmem_augment_frames': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(AugDesc), C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    'xmem_affinity_profile_events': (C.c_int, [C.c_void_p, C.c_void_p]),
    'xmem_affinity_rows16': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'xmem_affinity_topk': (C.c_int, [C.POINTER(KeySegment), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'xmem_affinity_topk_hinted': (C.c_int, [C.POINTER(KeySegment), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.POINTER(AffinityHint), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    'xmem_usage_update': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'xmem_readout_sparse': (C.c_int, [C.POINTER(ValueSegment), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    'xmem_readout_sparse_t': (C.c_int, [C.POINTER(ValueSegment), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p]),
    'xmem_similarity_dense': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    'xmem_usage_ratio': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'xmem_topk_1d': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'xmem_gather_rows': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'xmem_softmax_rows_suffix': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_softmax_rows_topk': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'xmem_weighted_rows': (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    'xmem_select_greater': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    'xmem_selector_prepare': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    'xmem_cycle_dissimilarity_workspace_bytes': (C.c_size_t, [C.c_int, C.c_int]),
    'xmem_cycle_dissimilarity': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())


class XMemHipError(RuntimeError):
    pass


def load():
    """Load libxmem_hip.so (once).  Raises loudly when it is absent - there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise XMemHipError(
            f'{LIB_PATH} not found: the MI355X kernels are not built. Run `python -m xmem2_amd.build` '
            '(or __graft_entry__.build()). xmem2_amd has no CPU / PyTorch fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.xmem_version() != ABI_VERSION:
        raise XMemHipError('libxmem_hip.so ABI version mismatch')
    _lib = lib
    return lib


def check(code):
    if code != 0:
        msg = load().xmem_last_error_string(code).decode()
        raise RuntimeError(f'xmem_hip: {msg} (status {code})')


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())
