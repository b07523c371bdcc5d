"""ctypes binding of librcu_hip.so (the C ABI declared in include/rcu.h).

The library is the product: if it is missing or a call fails this module raises -- there is no
Python / CPU fallback anywhere in the package.
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t,
                    c_uint64, c_void_p)

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# RCU_HIP_LIBRARY: an experiment build of the same library (csrc/Makefile: BUILD= / OUT=) for the timing tools under tools/
LIB_PATH = os.environ.get('RCU_HIP_LIBRARY') or os.path.join(PKG_DIR, 'librcu_hip.so')

RCU_MC_MI = 1
RCU_MC_VAR = 2
RCU_MC_INPUT_PROBS = 4
RCU_MC_EXACT = 8
RCU_MC_EXACT_MAX_PASSES = 2048
RCU_MAX_BINS = 32
RCU_MAX_THRESHOLDS = 16


class RcuError(RuntimeError):
    """A librcu_hip call returned a negative status."""

    def __init__(self, status, message):
        super().__init__('librcu_hip error {}: {}'.format(status, message))
        self.status = status


class UnetDesc(Structure):
    _fields_ = [(n, c_int32) for n in ('nb_classes', 'in_channels', 'depth', 'start_filters', 'has_dropout',
                                       'dropout_center', 'sigma_out', 'bn', 'height', 'width', 'max_batch', 'residual',
                                       'provide_features')]


class UnetOptions(Structure):
    """rcu_unet_options (include/rcu.h): the planner's kernel-family / layout choices; defaults = the shipped path."""
    _fields_ = [(n, c_int32) for n in ('conv_winograd', 'conv_winograd4', 'conv_first', 'act_layout', 'fuse_head', 'head_winograd4', 'pad_levels')] + [('reserved', c_int32 * 1)]


class LayerInfo(Structure):
    _fields_ = [('name', c_char * 96), ('kernel', c_char * 64), ('cin', c_int32), ('cout', c_int32),
                ('height', c_int32), ('width', c_int32), ('grid_height', c_int32), ('grid_width', c_int32), ('upsample', c_int32), ('pooled', c_int32),
                ('dual_source', c_int32), ('head_fusable', c_int32), ('flops_per_slice', c_double), ('mfma_flops_per_slice', c_double)]


class EceResult(Structure):
    _fields_ = [('count', c_uint64 * RCU_MAX_BINS), ('sum_conf', c_double * RCU_MAX_BINS),
                ('sum_pos', c_uint64 * RCU_MAX_BINS)]


# name -> (restype, argtypes); every symbol include/rcu.h declares
SIGNATURES = {
    'rcu_last_error': (c_char_p, []),
    'rcu_version': (c_char_p, []),
    'rcu_unet_create': (c_int, [POINTER(UnetDesc), POINTER(c_void_p)]),
    'rcu_unet_destroy': (c_int, [c_void_p]),
    'rcu_unet_default_options': (None, [POINTER(UnetOptions)]),
    'rcu_unet_plan': (c_int, [POINTER(UnetDesc), POINTER(UnetOptions), POINTER(c_void_p)]),
    'rcu_unet_create_with': (c_int, [POINTER(UnetDesc), POINTER(UnetOptions), c_void_p, POINTER(c_void_p)]),
    'rcu_unet_set_fuse_head': (c_int, [c_void_p, c_int]),
    'rcu_unet_workspace_bytes': (c_int64, [c_void_p]),
    'rcu_unet_num_dropout_sites': (c_int, [c_void_p]),
    'rcu_unet_dropout_site_channels': (c_int, [c_void_p, c_int]),
    'rcu_unet_dropout_site_name': (c_char_p, [c_void_p, c_int]),
    'rcu_unet_mask_floats_per_sample': (c_int, [c_void_p]),
    'rcu_unet_load_weight': (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    'rcu_unet_finalize_weights': (c_int, [c_void_p]),
    'rcu_unet_forward': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rcu_unet_forward_accumulate': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'rcu_unet_forward_accumulate_passes': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'rcu_unet_forward_accumulate_sigma': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'rcu_unet_forward_accumulate_sigma_passes': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'rcu_unet_features': (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int), POINTER(c_int)]),
    'rcu_postnet_create': (c_int, [c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    'rcu_postnet_destroy': (None, [c_void_p]),
    'rcu_postnet_load_weight': (c_int, [c_void_p, c_char_p, c_void_p, c_size_t]),
    'rcu_postnet_finalize_weights': (c_int, [c_void_p]),
    'rcu_postnet_forward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'rcu_unet_num_layers': (c_int, [c_void_p]),
    'rcu_unet_layer_info': (c_int, [c_void_p, c_int, POINTER(LayerInfo)]),
    'rcu_unet_run_layer': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'rcu_unet_profile_begin': (c_int, [c_void_p, c_int]),
    'rcu_unet_profile_collect': (c_int, [c_void_p, POINTER(c_double), POINTER(c_int)]),
    'rcu_mc_stats_bytes': (c_size_t, [c_size_t, c_size_t, c_int, c_int]),
    'rcu_mc_begin': (c_int, [c_void_p, c_size_t, c_size_t, c_int, c_int, c_void_p]),
    'rcu_mc_accumulate': (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_int, c_void_p]),
    'rcu_mc_finalize': (c_int, [c_void_p, c_size_t, c_size_t, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    'rcu_softmax': (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_void_p]),
    'rcu_aleatoric': (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p]),
    'rcu_prediction_and_foreground': (c_int, [c_void_p, c_size_t, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    'rcu_dropout_masks': (c_int, [POINTER(c_uint64), c_int, c_int, c_uint64, POINTER(c_int32), POINTER(c_float), c_int, c_void_p, c_void_p]),
    'rcu_ece_thresholds': (c_int, [c_int, POINTER(c_float)]),
    'rcu_ece_workspace_bytes': (c_size_t, [c_size_t, c_int]),
    'rcu_ece_hist': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, POINTER(c_float), c_int, c_void_p,
                             c_void_p, c_void_p]),
    'rcu_ece_bin_ids': (c_int, [c_void_p, c_size_t, POINTER(c_float), c_int, c_void_p, c_void_p]),
    'rcu_calib_set_blocks_per_workgroup': (c_int, [c_int, c_int]),
    'rcu_unc_workspace_bytes': (c_size_t, [c_size_t, c_int]),
    'rcu_unc_counts': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, POINTER(c_double),
                               c_int, c_void_p, c_void_p, c_void_p]),
    'rcu_normalised_entropy': (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    'rcu_unc_from_p_num_thresholds': (c_int, []),
    'rcu_unc_from_p_threshold': (c_double, [c_int]),
    'rcu_unc_from_p_supported': (c_int, [POINTER(c_double), c_int]),
    'rcu_unc_from_p_exceeded': (c_int, [c_float, POINTER(c_double), c_int]),
    'rcu_unc_from_p_workspace_bytes': (c_size_t, [c_size_t, c_int]),
    'rcu_unc_counts_from_p': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, POINTER(c_double), c_int, c_void_p, c_void_p,
                                      c_void_p]),
}

_lib = None


def load():
    """Load librcu_hip.so (once).  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; import it first so that librcu_hip binds to the SAME HIP runtime
    # (two runtimes in one process do not share a device context: pointers and streams would not be portable).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('{} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           '(there is no CPU fallback)'.format(LIB_PATH))
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status):
    """Raise RcuError for a negative status; pass non-negative values through."""
    if status < 0:
        raise RcuError(status, load().rcu_last_error().decode())
    return status


def ptr(t):
    """Device (or host) address of a torch tensor, None -> NULL."""
    return None if t is None else c_void_p(t.data_ptr())


class _DeviceMemory:
    """Borrowed device memory for torch.as_tensor (CUDA array interface v2); ``owner`` keeps the allocation alive."""

    def __init__(self, address, shape, owner):
        self.__cuda_array_interface__ = {'shape': tuple(int(v) for v in shape), 'typestr': '<f4',
                                         'data': (int(address), False), 'version': 2, 'strides': None}
        self.owner = owner


def device_view(address, shape, device, owner=None):
    """float32 torch tensor over library-owned device memory (no copy)."""
    import torch
    return torch.as_tensor(_DeviceMemory(address, shape, owner), device=device)


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ece_thresholds(n_bins=10):
    arr = (c_float * max(n_bins - 1, 1))()
    check(load().rcu_ece_thresholds(n_bins, arr))
    return arr
