"""ctypes binding of libmscl_hip.so (the C ABI declared in include/mscl_hip.h).

The library is built in-tree by mscl_amd/csrc/build.sh (see __graft_entry__.build).  There is no CPU
fallback: if the library is missing, or a kernel is asked to run on a non-GPU tensor, this module
raises -- the product path never silently degrades to PyTorch ops.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MSCL_LIB', os.path.join(_HERE, 'csrc', 'libmscl_hip.so'))     # MSCL_LIB: timing-probe builds


class MsclError(RuntimeError):
    pass


class ConvDesc(Structure):
    _fields_ = [(n, c_int) for n in ('N', 'T', 'H', 'W', 'C', 'To', 'Ho', 'Wo', 'K', 'kT', 'kH', 'kW',
                                     'sT', 'sH', 'sW', 'pT', 'pH', 'pW')]


class BnParams(Structure):
    _fields_ = [(n, c_void_p) for n in ('sum', 'sumsq', 'gamma', 'beta', 'running_mean', 'running_var',
                                        'num_batches_tracked', 'save_mean', 'save_invstd')]


P = c_void_p
_SIGS = {
    'mscl_abi_version': [],
    'mscl_set_deterministic': [c_int],
    'mscl_debug_pp_launches': [],
    'mscl_debug_thin_launches': [],
    'mscl_debug_k1_launches': [],
    'mscl_debug_dgrad_s2_launches': [],
    'mscl_debug_halo_launches': [],
    'mscl_debug_stem_launches': [],
    'mscl_debug_wgrad_halo_launches': [],
    'mscl_debug_wgrad_stem_launches': [],
    'mscl_debug_thin_wgrad_launches': [],
    'mscl_get_deterministic': [],
    'mscl_tuning_reload': [],
    'mscl_bn_stats': [P, P, P, c_int64, c_int, c_int, P, c_int64, P],
    'mscl_det_parts_floats': [c_int64, c_int, c_int, c_int],
    'mscl_conv3d_wgrad_ws': [POINTER(ConvDesc), c_int],
    'mscl_wgrad_halo_ws': [POINTER(ConvDesc)],
    'mscl_wgrad_thin_ws': [POINTER(ConvDesc)],
    'mscl_conv3d_fwd': [POINTER(ConvDesc), P, P, P, P, P, c_int, P, P, P, c_int64, P],
    'mscl_conv3d_fwd_groups': [POINTER(ConvDesc), P, P, P, P, P, c_int, P, P, c_int, P, c_int64, P],
    'mscl_conv_halo64': [POINTER(ConvDesc), c_int, P, P, P, P, P, P, P],
    'mscl_conv3d_dgrad': [POINTER(ConvDesc), P, P, P, P, P, c_int64, P],
    'mscl_conv3d_wgrad': [POINTER(ConvDesc), P, P, P, P, P, c_int64, P],
    'mscl_conv3d_wgrad_groupable': [POINTER(ConvDesc)],
    'mscl_conv3d_wgrad_group': [c_int, POINTER(ConvDesc), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), P],
    'mscl_debug_wgrad_group_launches': [],
    'mscl_weight_transpose': [P, P, c_int, c_int, c_int, P],
    'mscl_weight_transpose_batched': [P, c_int, c_int, P],
    'mscl_bn_act_fwd': [P, POINTER(BnParams), P, POINTER(BnParams), P, c_int64, c_int, c_float, c_float, c_int, P],
    'mscl_bn_act_bwd': [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, P, c_int64, c_int, c_int, P],
    'mscl_bn_act_fwd_groups': [P, POINTER(BnParams), P, POINTER(BnParams), P, c_int64, c_int, c_float, c_float, c_int, c_int, P],
    'mscl_bn_act_bwd_groups': [P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, P, c_int64, c_int, c_int, c_int, P, c_int64, P],
    'mscl_pack_input': [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), P, P],
    'mscl_pack_input_ind': [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), P, P],
    'mscl_pair_w': [P, P, c_int64, c_int, P],
    'mscl_flow_visualize': [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    'mscl_flow_fra_visualize': [P, P, c_float, c_float, c_int, P, P, P, P, c_int, c_int, c_int, c_int, P, P],
    'mscl_color_aug': [P, P, P, c_int, c_int, c_int, c_int, P],
    'mscl_gauss_blur': [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    'mscl_crop_resize_u8': [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, P],
    'mscl_crop_resize_f32': [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, P],
    'mscl_add_relu': [P, P, P, P, c_int64, c_int, P],
    'mscl_relu_bwd': [P, P, P, c_int64, P],
    'mscl_upsample_add': [P, P] + [c_int] * 10 + [P],
    'mscl_upsample_bwd': [P, P] + [c_int] * 9 + [P],
    'mscl_pool_fwd': [P, P, c_int, c_int, c_int, P],
    'mscl_pool_bwd': [P, P, c_int, c_int, c_int, c_int, P],
    'mscl_maxpool_hw_fwd': [P, P, P, c_int, c_int, c_int, c_int, P],
    'mscl_maxpool_hw_bwd': [P, P, P, c_int, c_int, c_int, c_int, P],
    'mscl_linear_fwd': [P, P, P, P, c_int, c_int, c_int, c_int, P],
    'mscl_linear_bwd': [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    'mscl_l2norm_fwd': [P, P, P, c_int, c_int, P],
    'mscl_l2norm_bwd': [P, P, P, P, c_int, c_int, P],
    'mscl_nce_fwd': [P, P, P, P, P, c_int, c_int, c_int, c_float, P],
    'mscl_nce_fwd_virt': [P, P, P, P, P, c_int, c_int, c_int, c_float, P, c_int, P, P],
    'mscl_nce_bwd_virt': [P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_float, P, c_int, P, P, P, P],
    'mscl_nce_finish': [P, P, P, P, P, c_int, c_int, c_float, P],
    'mscl_step_logs': [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, P],
    'mscl_nce_bwd': [P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_float, P],
    'mscl_rowdot': [P, P, P, c_int, c_int, P],
    'mscl_nce_pos_bwd': [P, P, P, P, P, c_int, c_int, c_float, P],
    'mscl_loss_pack': [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P],
    'mscl_loss_unpack': [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    'mscl_queue_enqueue': [P, P, P, P, c_int, c_int, c_int, P],
    'mscl_lmcl': [P, P, P, P, P, P, c_int, c_int, c_int, c_float, P],
    'mscl_ema_update': [P, P, P, c_int64, c_float, P],
    'mscl_ema_update_dev': [P, P, P, c_int64, P, P],
    'mscl_sgd_step_dev': [P, P, P, P, c_int64, P, c_float, P, c_float, c_float, P],
    'mscl_sumsq': [P, P, c_int64, P, c_int, P],
    'mscl_sgd_step': [P, P, P, P, c_int64, P, c_float, c_float, c_float, c_float, c_int, P],
    'mscl_cast_bf16': [P, P, c_int64, P],
}
_INT64_RESULT = ('mscl_debug_wgrad_stem_launches', 'mscl_debug_dgrad_s2_launches', 'mscl_debug_wgrad_group_launches', 'mscl_wgrad_halo_ws', 'mscl_debug_halo_launches', 'mscl_debug_stem_launches', 'mscl_debug_wgrad_halo_launches', 'mscl_det_parts_floats', 'mscl_conv3d_wgrad_ws', 'mscl_debug_pp_launches', 'mscl_debug_thin_launches', 'mscl_debug_k1_launches', 'mscl_debug_thin_wgrad_launches', 'mscl_wgrad_thin_ws')
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """Load (once) and return the CDLL; raises MsclError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MsclError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        f'or mscl_amd/csrc/build.sh (needs hipcc, --offload-arch=gfx950). There is no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int64 if name in _INT64_RESULT else c_int
    _lib = lib
    return lib


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr():
    """handle of the current HIP stream (raw accessor: torch.cuda.current_stream() costs ~8 us per call on the host,
    a third of the eager step's Python time at ~900 launches per step)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())          # plain int: ctypes converts it for a c_void_p parameter
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """device pointer of a tensor (None -> NULL); refuses non-GPU tensors."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MsclError('mscl_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback')
    return t.data_ptr()                           # plain int (see stream_ptr): +3 % on the host-bound eager step


def check(code, what):
    if code != 0:
        kind = {-1: 'bad argument', -2: 'unsupported shape', -3: 'unsupported stride'}.get(code, f'hipError {code}')
        raise MsclError(f'{what} failed: {kind}')


_fns = {}


def call_raw(name, *args):
    """the entry point's return code as is (for the few whose positive codes are not errors)"""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    return fn(*args)


def call(name, *args):
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    code = fn(*args)
    if code != 0:
        check(code, name)


def set_deterministic(on=True):
    """the reference's `--deterministic` (tools/train.py:55-57,149): fixed-order sums in place of float atomics (include/mscl_hip.h,
    mscl_set_deterministic); two runs on the same inputs are then bit-identical.  Process-wide, set before the first step."""
    global DET, DET_GEN
    call('mscl_set_deterministic', int(bool(on)))
    DET = bool(on)
    DET_GEN += 1              # cached per-(module, shape) plans hold mode-dependent scratch sizes: they key on this


def tune(**switches):
    """set (value) or clear (None) MSCL_* tuning switches of the library for this process and make it re-read them: the
    library caches its environment switches (csrc/common.h, MsclTune), so os.environ alone is not seen after the first launch"""
    for k, v in switches.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    call('mscl_tuning_reload')
    global DET_GEN
    DET_GEN += 1              # scratch sizes depend on the switches too (MSCL_WGRAD_HALO_*, MSCL_THIN): cached plans must not outlive them


WGRAD_GROUP_MAX = 16     # = MSCL_WGRAD_GROUP_MAX (include/mscl_hip.h)
DET_GEN = 0              # generation of everything a cached per-(module, shape) plan depends on: the deterministic flag AND the tuning
# switches.  Bumped by set_deterministic() and tune(); code that flips either through lib.call() directly must bump it itself.
DET = False              # mirror of the library's flag for the per-launch Python paths (scratch sizing)


def deterministic():
    return bool(call_raw('mscl_get_deterministic'))


def det_parts_floats(rows, C, groups, vecs):
    """floats of scratch the deterministic BatchNorm sums want for their per-block partials (0 outside deterministic mode)"""
    return call_raw('mscl_det_parts_floats', rows, C, groups, vecs) if DET else 0
