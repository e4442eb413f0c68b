"""ctypes binding of libfieldconv_hip.so (the C ABI in include/fieldconv_hip.h).

There is deliberately no fallback: if the library is missing or a call fails, the product path
raises.  Nothing under oracle/ is ever imported from here.
"""
import ctypes
import os

from .build import DEV_LIB_PATH, LIB_PATH

_c_int32 = ctypes.c_int32
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t


class FcDims(ctypes.Structure):
    """fc_dims.  `mode` (fc_mfma_mode: the arithmetic of the contractions) travels with every call; left out, it is the binding's
    current mode (current_mode(): FC_MFMA of the environment, or the innermost `arithmetic(...)` block)."""
    _fields_ = [('N', _c_int32), ('E', _c_int32), ('I', _c_int32), ('O', _c_int32), ('R', _c_int32), ('B', _c_int32), ('mode', _c_int32)]

    def __init__(self, N=0, E=0, I=0, O=0, R=0, B=0, mode=None):
        super().__init__(N, E, I, O, R, B, current_mode() if mode is None else mode)


class FcCsr(ctypes.Structure):
    _fields_ = [('rowptr', _vp), ('nbr', _vp), ('runs', _vp)]


class FcFilterParams(ctypes.Structure):
    """fc_filter_params: module parameters and (backward) where their gradients go"""
    _fields_ = [('zonal', ctypes.c_void_p), ('spherical', ctypes.c_void_p), ('phase', ctypes.c_void_p), ('ftype', ctypes.c_int32),
                ('g_zonal', ctypes.c_void_p), ('g_spherical', ctypes.c_void_p), ('g_phase', ctypes.c_void_p),
                # rider of the finishing launch: the modReLU's bias-gradient partials summed into g_bias (None / 0: nothing)
                ('bias_partials', ctypes.c_void_p), ('bias_nparts', ctypes.c_int32), ('g_bias', ctypes.c_void_p)]


class FcEpilogue(ctypes.Structure):
    _fields_ = [('addend', _vp), ('modrelu_bias', _vp), ('activated', _vp)]


class FcMesh(ctypes.Structure):
    """fc_mesh: one mesh's support graph as the block-level entry points take it"""
    _fields_ = [('N', _c_int32), ('E', _c_int32), ('R', _c_int32), ('B', _c_int32), ('kind', _c_int32),
                ('by_target', ctypes.POINTER(FcCsr)), ('by_source', ctypes.POINTER(FcCsr)), ('fwd', _vp), ('bwd', _vp), ('mode', _c_int32)]

    def __init__(self, N=0, E=0, R=0, B=0, kind=0, by_target=None, by_source=None, fwd=None, bwd=None, mode=None):
        super().__init__(N, E, R, B, kind, by_target, by_source, fwd, bwd, current_mode() if mode is None else mode)


class FcResnetBlockParams(ctypes.Structure):
    _fields_ = [('C_in', _c_int32), ('C_mid', _c_int32), ('C_out', _c_int32), ('conv1', FcFilterParams), ('conv2', FcFilterParams),
                ('bias1', _vp), ('bias2', _vp), ('res_re', _vp), ('res_im', _vp),
                ('g_bias1', _vp), ('g_bias2', _vp), ('g_res_re', _vp), ('g_res_im', _vp)]


class FcEchoBlockParams(ctypes.Structure):
    _fields_ = [('C_in', _c_int32), ('n_des', _c_int32), ('n_bins', _c_int32), ('conv', FcFilterParams), ('bias', _vp), ('g_bias', _vp)]


class FcEchoHeadParams(ctypes.Structure):
    """fc_echo_head_params: ECHOBlock's dense tail (lin1, lin2, lin3, res) and where its gradients go"""
    _fields_ = [('D', _c_int32), ('H1', _c_int32), ('H2', _c_int32), ('C_in', _c_int32), ('C_out', _c_int32)] + \
               [(n, _vp) for n in ('w1', 'b1', 'w2', 'b2', 'w3', 'b3', 'wr', 'br', 'g_w1', 'g_b1', 'g_w2', 'g_b2', 'g_w3', 'g_b3', 'g_wr', 'g_br')]


class FcLiftBlockParams(ctypes.Structure):
    _fields_ = [('C_in', _c_int32), ('C_out', _c_int32), ('ftype', _c_int32), ('zonal_ang', _vp), ('zonal_mag', _vp), ('phase', _vp),
                ('bias', _vp), ('g_zonal_ang', _vp), ('g_zonal_mag', _vp), ('g_phase', _vp), ('g_bias', _vp)]


_DP = ctypes.POINTER(FcDims)
_MP = ctypes.POINTER(FcMesh)
_RBP = ctypes.POINTER(FcResnetBlockParams)
_EBP = ctypes.POINTER(FcEchoBlockParams)
_EHP = ctypes.POINTER(FcEchoHeadParams)
_LBP = ctypes.POINTER(FcLiftBlockParams)
_CP = ctypes.POINTER(FcCsr)
_EP = ctypes.POINTER(FcEpilogue)
_FP = ctypes.POINTER(FcFilterParams)

# name -> (restype, argtypes); must list every symbol declared in include/fieldconv_hip.h
SIGNATURES = {
    'fc_abi_version': (ctypes.c_int, []),
    'fc_dev_switches': (ctypes.c_int, []),
    'fc_debug_stamp_buffer': (None, [_vp]),
    'fc_status_string': (ctypes.c_char_p, [ctypes.c_int]),
    'fc_supported': (ctypes.c_int, [_DP]),
    'fc_describe_kernels': (ctypes.c_int, [_DP, _c_int32, ctypes.c_char_p, ctypes.c_size_t]),
    'fc_shape_compiled': (ctypes.c_int, [_c_int32, _c_int32]),
    'fc_generic_gather': (ctypes.c_int, [_vp, _vp, _CP, _vp, _c_int32, _c_int32, _c_int32, _c_int32, _c_int32, _vp]),
    'fc_wide_workspace_bytes': (ctypes.c_size_t, [_DP, _c_int32, _c_int32, _c_int32]),
    'fc_forward_wide': (ctypes.c_int, [_vp, _vp, _CP, _c_int32, _vp, _vp, _vp, _vp, ctypes.c_size_t, _DP, _c_int32, _c_int32, _vp]),
    'fc_backward_wide': (ctypes.c_int, [_vp, _vp, _vp, _CP, _c_int32, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _DP, _c_int32, _vp]),
    'fc_cgemm_workspace_bytes': (ctypes.c_size_t, [_c_int32, _c_int32, _c_int32, _c_int32]),
    'fc_cgemm': (ctypes.c_int, [_vp, _vp, _vp, _c_int32, _c_int32, _c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                _c_int32, ctypes.c_double, _c_int32, _vp, ctypes.c_size_t, _vp]),
    'fc_tangent_nonlin_forward_f64': (ctypes.c_int, [_vp, _vp, _vp, _c_int32, _c_int32, _vp]),
    'fc_tangent_nonlin_backward_workspace_bytes_f64': (ctypes.c_size_t, [_c_int32, _c_int32]),
    'fc_tangent_nonlin_backward_f64': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _c_int32, _c_int32, _vp]),
    'fc_generic_scatter': (ctypes.c_int, [_vp, _vp, _vp, _CP, _vp, _c_int32, _c_int32, _c_int32, _c_int32, _c_int32, _vp]),
    'fc_packed_filter_floats_fwd': (_sz, [_DP, _c_int32]),
    'fc_packed_filter_floats_bwd': (_sz, [_DP, _c_int32]),
    'fc_pack_filter': (ctypes.c_int, [_vp, _vp, _vp, _DP, _c_int32, _vp]),
    'fc_pack_filter_params': (ctypes.c_int, [_vp, _vp, _vp, _c_int32, _vp, _vp, _DP, _c_int32, _vp]),
    'fc_filter_param_grads': (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_int32, _vp, _vp, _vp, _DP, _vp]),
    'fc_forward': (ctypes.c_int, [_vp, _vp, _CP, _vp, _vp, _DP, _EP, _vp]),
    'fc_factored_record_floats': (ctypes.c_int, [_c_int32]),
    'fc_forward_workspace_bytes': (_sz, [_DP]),
    'fc_forward_factored': (ctypes.c_int, [_vp, _vp, _CP, _vp, _vp, _vp, _sz, _DP, _EP, _vp]),
    'fc_geometric_record_floats': (ctypes.c_int, []),
    'fc_forward_geometric': (ctypes.c_int, [_vp, _vp, _CP, _vp, _vp, _vp, _sz, _DP, _EP, _vp]),
    'fc_records_flags': (_c_int32, [_DP, _c_int32]),
    'fc_backward_workspace_bytes': (_sz, [_DP, _c_int32]),
    'fc_backward_data': (ctypes.c_int, [_vp, _vp, _vp, _CP, _vp, _vp, _vp, _sz, _DP, _vp]),
    'fc_backward_data_factored': (ctypes.c_int, [_vp, _vp, _vp, _CP, _vp, _vp, _vp, _sz, _DP, _c_int32, _vp]),
    'fc_backward_filter': (ctypes.c_int, [_vp, _vp, _sz, _DP, _c_int32, _vp]),
    'fc_backward_streams': (_c_int32, [_DP, _c_int32]),
    'fc_backward_gather': (ctypes.c_int, [_vp, _vp, _CP, _vp, _vp, _sz, _DP, _vp]),
    'fc_backward_stream': (ctypes.c_int, [_vp, _vp, _vp, _vp, _sz, _DP, _vp]),
    'fc_backward_finish': (ctypes.c_int, [_vp, _vp, _sz, _DP, _c_int32, _vp]),
    'fc_backward_finish_params': (ctypes.c_int, [_vp, _vp, _sz, _DP, _c_int32, _FP, _vp]),
    'fc_backward_all': (ctypes.c_int, [_vp, _vp, _vp, _CP, _c_int32, _vp, _vp, _vp, _FP, _vp, _sz, _DP, _vp]),
    'fc_forward_params': (ctypes.c_int, [_vp, _vp, _CP, _c_int32, _FP, _vp, _vp, _vp, _vp, _sz, _DP, _c_int32, _EP, _vp]),
    'fc_echo_hist_dim': (ctypes.c_int, [_c_int32]),
    'fc_echo_channel_block': (ctypes.c_int, [_c_int32]),
    'fc_echo_forward': (ctypes.c_int, [_vp, _vp, _vp, _CP, _vp, _vp, _c_int32, _c_int32, _c_int32, _c_int32, _vp]),
    'fc_echo_backward': (ctypes.c_int, [_vp, _vp, _vp, _CP, _vp, _vp, _vp, _vp, _c_int32, _c_int32, _c_int32, _c_int32, _vp]),
    'fc_adam_step': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _sz] + [ctypes.c_float] * 5 + [_vp]),
    'fc_precomp_workspace_bytes': (_sz, [_c_int32, _c_int32]),
    'fc_precomp_mark': (ctypes.c_int, [_vp, _vp, ctypes.c_float, _c_int32, _c_int32, _vp, _sz, _vp]),
    'fc_precomp_graph': (ctypes.c_int, [_vp] * 5 + [ctypes.c_float] + [_c_int32] * 7 + [_vp] * 16 + [_vp, _sz, _vp, _sz, _vp]),
    'fc_precomp_kept_count_ptr': (_vp, [_vp, _c_int32]),
    'fc_precomp_build': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, ctypes.c_float, _c_int32, _c_int32, _c_int32, _c_int32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'fc_graph_workspace_bytes': (_sz, [_c_int32, _c_int32, _c_int32, _c_int32, _c_int32]),
    'fc_graph_build': (ctypes.c_int, [_vp, _vp, _c_int32, _c_int32, _c_int32, _c_int32] + [_vp] * 12 + [_vp, _sz, _vp]),
    'fc_trans_field_forward': (ctypes.c_int, [_vp, _vp, _CP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int32, _c_int32, _c_int32, _c_int32, _c_int32, _c_int32, _vp]),
    'fc_trans_field_backward_workspace_bytes': (_sz, [_c_int32, _c_int32, _c_int32, _c_int32]),
    'fc_trans_field_backward': (ctypes.c_int, [_vp, _CP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _c_int32, _c_int32, _c_int32, _c_int32, _c_int32, _c_int32, _c_int32, _vp]),
    'fc_tangent_lin_forward': (ctypes.c_int, [_vp, _vp, _vp, _vp, _c_int32, _c_int32, _c_int32, _vp]),
    'fc_tangent_lin_backward_workspace_bytes': (_sz, [_c_int32, _c_int32, _c_int32]),
    'fc_tangent_lin_backward': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _c_int32, _c_int32, _c_int32, _vp]),
    'fc_soft_abs_forward': (ctypes.c_int, [_vp, _vp, _sz, _vp]),
    'fc_soft_abs_backward': (ctypes.c_int, [_vp, _vp, _vp, _sz, _vp]),
    'fc_tangent_nonlin_forward': (ctypes.c_int, [_vp, _vp, _vp, _c_int32, _c_int32, _vp]),
    'fc_tangent_nonlin_backward_workspace_bytes': (_sz, [_c_int32, _c_int32]),
    'fc_tangent_nonlin_backward': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _c_int32, _c_int32, _vp]),
    'fc_tangent_nonlin_backward_groups': (_c_int32, [_c_int32]),
    'fc_tangent_nonlin_backward_partial': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _sz, _c_int32, _c_int32, _vp]),
    'fc_trans_field_forward_generic': (ctypes.c_int, [_vp, _vp, _CP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp] + [_c_int32] * 7 + [_vp]),
    'fc_trans_field_backward_generic_workspace_bytes': (_sz, [_c_int32] * 5),
    'fc_trans_field_backward_generic': (ctypes.c_int, [_vp, _CP, _vp] + [_vp] * 11 + [_vp, _sz] + [_c_int32] * 8 + [_vp]),
    'fc_echo_hist_dim_generic': (ctypes.c_int, [_c_int32]),
    'fc_echo_generic_workspace_bytes': (_sz, [_c_int32]),
    'fc_echo_forward_generic': (ctypes.c_int, [_vp, _vp, _vp, _CP, _vp, _vp, _vp, _sz] + [_c_int32] * 5 + [_vp]),
    'fc_echo_backward_generic': (ctypes.c_int, [_vp, _vp, _vp, _CP, _vp, _vp, _vp, _vp, _vp, _sz] + [_c_int32] * 5 + [_vp]),
    'fc_resnet_block_saved_bytes': (_sz, [_MP, _RBP]),
    'fc_resnet_block_workspace_bytes': (_sz, [_MP, _RBP, _c_int32]),
    'fc_resnet_block_forward': (ctypes.c_int, [_vp, _MP, _RBP, _vp, _vp, _sz, _vp, _sz, _vp]),
    'fc_resnet_block_backward': (ctypes.c_int, [_vp, _vp, _MP, _RBP, _vp, _sz, _vp, _vp, _sz, _vp]),
    'fc_echo_block_saved_bytes': (_sz, [_MP, _EBP]),
    'fc_echo_block_workspace_bytes': (_sz, [_MP, _EBP, _c_int32]),
    'fc_echo_block_forward': (ctypes.c_int, [_vp, _MP, _vp, _vp, _EBP, _vp, _vp, _sz, _vp, _sz, _vp]),
    'fc_echo_block_backward': (ctypes.c_int, [_vp, _vp, _MP, _vp, _vp, _EBP, _vp, _sz, _vp, _vp, _sz, _vp]),
    'fc_echo_head_forward_workspace_bytes': (_sz, [_c_int32, _EHP]),
    'fc_echo_head_forward': (ctypes.c_int, [_vp, _vp, _EHP, _vp, _vp, _vp, _vp, _sz, _c_int32, _vp]),
    'fc_echo_head_backward_workspace_bytes': (_sz, [_c_int32, _EHP]),
    'fc_echo_head_backward': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _EHP, _vp, _vp, _vp, _vp, _sz, _c_int32, _vp]),
    'fc_lift_block_saved_bytes': (_sz, [_MP, _LBP]),
    'fc_lift_block_workspace_bytes': (_sz, [_MP, _LBP, _c_int32]),
    'fc_lift_block_forward': (ctypes.c_int, [_vp, _vp, _c_int32, _MP, _vp, _LBP, _vp, _vp, _sz, _vp]),
    'fc_lift_block_backward': (ctypes.c_int, [_vp, _vp, _c_int32, _MP, _vp, _LBP, _vp, _sz, _vp, _vp, _sz, _vp]),
}

_LIB = None


def keep_mode(fn_cls):
    """Class decorator for the autograd Functions that launch convolutions: the backward pass runs in the arithmetic mode of ITS
    forward pass -- the autograd engine calls it from its own thread, outside any `arithmetic(...)` block of the caller, and the
    packed filter images a forward pass saves are only valid in the mode they were packed in."""
    fwd, bwd = fn_cls.forward, fn_cls.backward

    def forward(ctx, *args):
        ctx.fc_mode = current_mode()
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        with arithmetic(ctx.fc_mode):
            return bwd(ctx, *grads)
    fn_cls.forward, fn_cls.backward = staticmethod(forward), staticmethod(backward)
    return fn_cls


class FieldConvNativeError(RuntimeError):
    pass


MFMA_MODES = {'': 0, 'split': 0, 'f32': 1, 'f16': 2}       # FC_MFMA / arithmetic(...) -> fc_mfma_mode

import threading as _threading

_MODE = _threading.local()


def current_mode():
    """fc_mfma_mode the binding puts into the dims of the calls it makes now: the innermost `arithmetic(...)` block of this thread, else
    FC_MFMA of the environment (default: split halves).  The library itself keeps no mode and reads no variable."""
    stack = getattr(_MODE, 'stack', None)
    if stack:
        return stack[-1]
    mode = os.environ.get('FC_MFMA', '')
    if mode not in MFMA_MODES:
        raise FieldConvNativeError(f"FC_MFMA={mode!r}: expected one of 'split' (default), 'f32', 'f16'")
    return MFMA_MODES[mode]


class arithmetic:
    """`with fieldconv_amd.arithmetic('f32'):` -- the convolutions launched inside run in that arithmetic mode ('split', 'f32',
    'f16'); a forward pass and its backward pass must run in the same one (the packed filter images follow the mode)."""

    def __init__(self, mode):
        if isinstance(mode, int) and mode in MFMA_MODES.values():       # (an fc_mfma_mode code: keep_mode below)
            self.mode = mode
            return
        if mode not in MFMA_MODES:
            raise FieldConvNativeError(f"arithmetic({mode!r}): expected one of 'split', 'f32', 'f16'")
        self.mode = MFMA_MODES[mode]

    def __enter__(self):
        if not hasattr(_MODE, 'stack'):
            _MODE.stack = []
        _MODE.stack.append(self.mode)
        return self

    def __exit__(self, *exc):
        _MODE.stack.pop()


def _pick_library():
    """FIELDCONV_HIP_LIB if given; else the product library -- which reads no environment variable.  The development build
    (-DFC_DEV_SWITCHES) is loaded only on request: FIELDCONV_DEV=1.  One of the library's development switches set (to a non-empty
    value) WITHOUT that request raises: a stray FC_* variable must neither be silently ignored (an A/B run that measures the default
    twice) nor silently swap the binary a consumer of the package runs."""
    explicit = os.environ.get('FIELDCONV_HIP_LIB')
    if explicit:
        return explicit
    from ._env import LIBRARY_SWITCHES
    wanted = sorted(k for k in LIBRARY_SWITCHES if os.environ.get(k, '') != '')
    dev = os.environ.get('FIELDCONV_DEV', '') == '1'
    if wanted and not dev:
        raise FieldConvNativeError(
            f'{", ".join(wanted)} set, but the product library has no environment switches: set FIELDCONV_DEV=1 to load the development '
            f'build ({DEV_LIB_PATH}), or unset the variable(s)')
    if dev:
        if not os.path.exists(DEV_LIB_PATH):
            raise FieldConvNativeError(
                f'FIELDCONV_DEV=1, but the development library {DEV_LIB_PATH} is not built: '
                'build it with `python -m fieldconv_amd.build --dev` (fieldconv_amd.build.build_dev())')
        return DEV_LIB_PATH
    return LIB_PATH


def load(path=None):
    """dlopen the library and bind every entry point; raises if it is not built.  The arithmetic mode travels in the dims of every
    call (FcDims.mode, current_mode()): the library itself keeps no mode and reads no environment variable."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    path = path or _pick_library()
    if not os.path.exists(path):
        raise FieldConvNativeError(
            f'{path} not found: build it with `python -m fieldconv_amd.build` '
            '(fieldconv_amd has no CPU or eager fallback for the field-convolution path)')
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    current_mode()          # (a bad FC_MFMA fails here, not at the first convolution)
    _LIB = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().fc_status_string(int(status)).decode()
        raise FieldConvNativeError(f'{what} failed: {msg} (status {status})')
