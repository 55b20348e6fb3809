"""ctypes binding of libtfnas_hip.so (C ABI: include/tfnas_hip.h).

The shared library is built in-tree by ``tf-nas_amd/csrc/Makefile`` (``__graft_entry__.build()``).  There is
NO fallback: if the library is missing or an entry point returns non-zero, a RuntimeError is raised.
"""
import ctypes as C
import os

# torch FIRST: the PyTorch-ROCm wheel bundles its own libamdhip64.so.  If libtfnas_hip.so were loaded before torch, the dynamic
# loader would bind it to the system ROCm's HIP runtime instead, the process would hold two HIP runtimes, and every launch on a
# torch-allocated pointer would fail with hipErrorNoDevice (100) -- seen when build() and smoke() ran in one process.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TFNAS_LIB') or os.path.join(_HERE, 'libtfnas_hip.so')     # (TFNAS_LIB: an experiment build of the same ABI)

MAX_GROUPS, MAX_SINK, MAX_CELLS = 8, 4, 32
ACT = {'relu': 0, 'swish': 1}
MODE_CELL, MODE_STEM, MODE_HEAD = 0, 1, 2

_W_FIELDS = ('w_expand', 'w_dw', 'w_proj', 'w_se_r', 'b_se_r', 'w_se_e', 'b_se_e')
_G_FIELDS = ('g_expand', 'g_dw', 'g_proj', 'g_se_r', 'gb_se_r', 'g_se_e', 'gb_se_e')


GEMM_EXPLICIT, GEMM_EVERYWHERE, CELL_LAZY_JOIN = 0x1000, 0x100, 1
# TfnasCellDesc.route (include/tfnas_hip.h: TFNAS_ROUTE_*) -- every kernel-variant switch of a launch; 0 = the library's policy
ROUTE_FX_OFF, ROUTE_FOLD_OFF, ROUTE_DWWG_OFF, ROUTE_DWWG2_OFF, ROUTE_XG_OFF, ROUTE_XG_ALL = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
ROUTE_DW = {'auto': 0, 'direct': 1 << 6, 'lds': 2 << 6, 'tiled': 3 << 6}
ROUTE_SE = {'wave': 0, 'fused': 1 << 8, 'gemm': 2 << 8}
ROUTE_WGRAD_INLINE, ROUTE_GRAM2 = 0x400, 0x800
ROUTE_TAKEN_VALID, ROUTE_TAKEN_FX = 1, 2
GEMM_MODES = {'f32': 0, 'bf16': 1, 'x2': 3, 'x3': 6}


class TfnasGroup(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('mc', 'k', 'se', 'mcp', 'off', 'se_off', 'pad0', 'pad1')]
                + [(n, C.c_void_p) for n in _W_FIELDS] + [(n, C.c_void_p) for n in _G_FIELDS])


class TfnasCellDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('N', 'H', 'W', 'ic', 'oc', 'stride', 'act', 'has_res', 'G', 'need_wgrad',
                                          'Ho', 'Wo', 'M', 'SE')]
                + [('eps', C.c_float), ('mode', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32),
                   ('stor', C.c_int32), ('gemm_mode', C.c_int32), ('flags', C.c_int32), ('g', TfnasGroup * MAX_GROUPS),
                   ('sync_fn', C.c_void_p), ('sync_user', C.c_void_p), ('sync_world', C.c_int32), ('route', C.c_int32),
                   ('fwd_route', C.c_int32), ('pad_route', C.c_int32), ('wgrad_stream', C.c_void_p * 3)])


class TfnasCellWs(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        'E', 'D', 'Pr', 'fsmall', 'off_pooled', 'off_gate', 'off_hpre', 'stats', 'off_stats1', 'off_stats2',
        'off_stats3', 'out', 'dZ', 'dEh', 'bsmall', 'off_dgate', 'off_dpooled', 'off_dgl', 'off_dhpre', 'off_cb1',
        'red', 'off_red3', 'off_red2', 'off_red1', 'off_resdot', 'part', 'dx', 'dxp')]


MAX_STAGES = 8


class TfnasStage(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('ncell', 'start_res', 'first_cell', 'nres')]
                + [('betas', C.c_void_p), ('dbetas', C.c_void_p)])


class TfnasPathDesc(C.Structure):
    _fields_ = ([(n, C.c_int32) for n in ('ncell', 'nstage', 'soft', 'need_dx0', 'efree_mask_lo', 'reserved0')]
                + [('stage', TfnasStage * MAX_STAGES), ('cell', TfnasCellDesc * MAX_CELLS)])


class TfnasPathWs(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ('saved', 'scratch', 'total', 'out_count')]
                + [(n, C.c_int32) for n in ('out_h', 'out_w', 'out_c', 'pad')])


class TfnasBnAffine(C.Structure):
    _fields_ = ([(n, C.c_void_p * 3) for n in ('weight', 'bias', 'g_weight', 'g_bias', 'running_mean', 'running_var')]
                + [('momentum', C.c_float), ('eval', C.c_int32)])


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_PROTOS = {
    'tfnas_abi_version': (C.c_int, []),
    'tfnas_shutdown': (C.c_int, []),
    'tfnas_sizeof': (C.c_uint64, [C.c_int]),
    'tfnas_cell_plan': (C.c_int, [C.POINTER(TfnasCellDesc)]),
    'tfnas_cell_ws': (C.c_int, [C.POINTER(TfnasCellDesc), C.POINTER(TfnasCellWs)]),
    'tfnas_mixedop_fwd': (C.c_int, [C.POINTER(TfnasCellDesc)] + [_P] * 10),
    'tfnas_mixedop_bwd': (C.c_int, [C.POINTER(TfnasCellDesc)] + [_P] * 17),
    'tfnas_mbconv_fwd': (C.c_int, [C.POINTER(TfnasCellDesc), C.POINTER(TfnasBnAffine)] + [_P] * 10),
    'tfnas_mbconv_bwd': (C.c_int, [C.POINTER(TfnasCellDesc), C.POINTER(TfnasBnAffine)] + [_P] * 17),
    'tfnas_head_affine_fwd': (C.c_int, [C.POINTER(TfnasCellDesc), C.POINTER(TfnasBnAffine)] + [_P] * 6),
    'tfnas_head_affine_bwd': (C.c_int, [C.POINTER(TfnasCellDesc), C.POINTER(TfnasBnAffine)] + [_P] * 11),
    'tfnas_head_fwd': (C.c_int, [C.POINTER(TfnasCellDesc)] + [_P] * 6),
    'tfnas_head_bwd': (C.c_int, [C.POINTER(TfnasCellDesc)] + [_P] * 11),
    'tfnas_head_wgrad': (C.c_int, [C.POINTER(TfnasCellDesc)] + [_P] * 6),
    'tfnas_cls_ce': (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, C.c_float, _P, _P, _P, _P, _P]),
    'tfnas_cls_wgrad': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _PP, _PP, _PP, C.c_float, _P, _P, _P, _P]),
    'tfnas_add_into': (C.c_int, [_P, _P, C.c_uint64, _P]),
    'tfnas_path_create': (C.c_int, [C.POINTER(C.c_void_p)]),
    'tfnas_path_destroy': (C.c_int, [C.c_void_p]),
    'tfnas_path_set_side_stream': (C.c_int, [C.c_void_p, C.c_void_p]),
    'tfnas_set_stats_sync': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    'tfnas_path_plan': (C.c_int, [C.c_void_p, C.POINTER(TfnasPathDesc), C.POINTER(TfnasPathWs)]),
    'tfnas_paths_fwd': (C.c_int, [C.c_int] + [_PP] * 8),
    'tfnas_paths_bwd': (C.c_int, [C.c_int] + [_PP] * 11 + [C.c_int, C.c_int]),
    'tfnas_pack_ranges': (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _P]),
    'tfnas_sgd_clip_step': (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64), C.c_float, C.c_float,
                                      C.c_float, C.c_float, C.c_float, _P, C.c_uint64, _P, _P]),
    'tfnas_arch_adam_project': (C.c_int, [C.c_int, _PP, _PP, C.POINTER(C.c_int32), _P, _P, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, _P, _P]),
    'tfnas_arch_fwd': (C.c_int, [C.c_int, C.POINTER(_P), _P, _P, C.c_float, _P, _P, _P]),
    'tfnas_arch_bwd': (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_float, C.POINTER(_P), _P]),
    'tfnas_efree_supported': (C.c_int, [C.POINTER(TfnasCellDesc)]),
    'tfnas_fx_supported': (C.c_int, [C.POINTER(TfnasCellDesc)]),
    'tfnas_cell_route': (C.c_int, [C.POINTER(TfnasCellDesc)]),
    'tfnas_arch_project': (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(C.c_int32), _P]),
    'tfnas_arch_sample': (C.c_int, [C.c_int, C.POINTER(_P), _P, _P, C.c_float, C.c_int, _P, _P]),
    'tfnas_sink_fwd': (C.c_int, [C.c_int, _P, C.POINTER(_P), _P, C.c_uint64, _P, _P, _P, _P]),
    'tfnas_prof_enable': (C.c_int, [C.c_uint]),
    'tfnas_set_lazy_join': (C.c_int, [C.c_int]),
    'tfnas_set_gemm_mode': (C.c_int, [C.c_int]),
    'tfnas_gemm_mode': (C.c_int, []),
    'tfnas_side_stream': (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    'tfnas_side_join': (C.c_int, [_P]),
    'tfnas_prof_count': (C.c_int, []),
    'tfnas_prof_name': (C.c_char_p, [C.c_int]),
    'tfnas_prof_collect': (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    'tfnas_prof_last_split': (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    'tfnas_sink_bwd': (C.c_int, [C.c_int, _P, C.POINTER(_P), _P, _P, _P, C.c_uint64, C.POINTER(_P), _P, _P, _P, _P]),
}

_lib_handle = None


def lib():
    """Load (once) and return the shared library; raises RuntimeError when it is absent."""
    global _lib_handle
    l = _lib_handle
    if l is None:
        path = LIB_PATH
        if not os.path.exists(path):
            raise RuntimeError(
                'tfnas_amd: HIP extension %s is missing -- run `python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C tf-nas_amd/csrc`). There is no CPU/PyTorch fallback for the MixedOP hot path.' % path)
        l = C.CDLL(path)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)        # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if l.tfnas_abi_version() != 4:
            raise RuntimeError('tfnas_amd: ABI version mismatch')
        for which, st in enumerate((TfnasGroup, TfnasCellDesc, TfnasCellWs, TfnasStage, TfnasPathDesc, TfnasPathWs, TfnasBnAffine)):
            if l.tfnas_sizeof(which) != C.sizeof(st):
                raise RuntimeError('tfnas_amd: struct layout mismatch for %s' % st.__name__)
        # the library reads no environment variable (ABI 4): TFNAS_GEMM only seeds its process-wide default here
        g = os.environ.get('TFNAS_GEMM')
        if g is not None:
            if g not in GEMM_MODES:
                raise RuntimeError('tfnas_amd: TFNAS_GEMM must be one of %s' % sorted(GEMM_MODES))
            if l.tfnas_set_gemm_mode(GEMM_MODES[g]) != 0:
                raise RuntimeError('tfnas_amd: tfnas_set_gemm_mode failed')
        _lib_handle = l
    return l


def check(rc, what):
    if rc != 0:
        raise RuntimeError('tfnas_hip: %s failed with code %d (%s)' % (
            what, rc, 'invalid argument' if rc < 0 else 'hipError_t'))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def raw_array(values):
    """(void* x n) from raw integers / None."""
    arr = (C.c_void_p * len(values))()
    for i, v in enumerate(values):
        arr[i] = v
    return arr


def exported_names():
    return list(_PROTOS)
