"""ctypes binding of libbeta_cores.so (the C ABI declared in include/beta_cores.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is
visible, the calls below raise.  Status codes map to the reference's exception
types: 1 -> NumericalPrecisionError, 2 -> ValueError, <0 -> RuntimeError.
"""
import ctypes as C
import os

from .util.errors import NumericalPrecisionError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('BETA_CORES_LIB') or os.path.join(_HERE, 'libbeta_cores.so')   # override: A/B builds side by side

BC_OK, BC_NUMERICAL_PRECISION, BC_INVALID_ARGUMENT, BC_RETRY_EXACT = 0, 1, 2, 3
ALG_GIGA, ALG_FW, ALG_OMP = 0, 1, 2
TILE_ROWS = 128

c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)

# name -> argtypes ; every function returns int except the two noted below
_SIGNATURES = {
    'bc_ctx_create': [C.c_int, vp, vpp],
    'bc_ctx_destroy': [vp],
    'bc_ctx_sync': [vp],
    'bc_ctx_kernel_time': [vp, C.c_int, c_dp, c_i64p],
    'bc_ctx_kernel_time_reset': [vp],
    'bc_ctx_enable_timing': [vp, C.c_int],
    'bc_ctx_timing_classes': [vp, C.c_uint32],
    'bc_ctx_phase_times': [vp, vp, C.c_int32, c_i64p, C.c_int],
    'bc_data_from_host': [vp, vp, C.c_int64, C.c_int32, vpp],
    'bc_data_from_device': [vp, vp, C.c_int64, C.c_int32, vpp],
    'bc_data_create': [vp, C.c_int64, C.c_int32, vpp],
    'bc_data_upload': [vp, vp, C.c_int64],
    'bc_data_gather_rows': [vp, vp, C.c_int64, vp],
    'bc_data_zero_feature_keys': [vp, C.c_int32, C.c_int64, vp, C.POINTER(C.c_int64)],
    'bc_ctx_set_constant_row_values': [vp, C.c_int, vp, C.c_int32, vp, vp, C.c_int64],
    'bc_data_destroy': [vp],
    'bc_phi_from_host': [vp, vp, C.c_int64, C.c_int32, C.c_int64, vpp],
    'bc_phi_create': [vp, C.c_int64, C.c_int32, vpp],
    'bc_project': [vp, vp, C.c_int, vp, C.c_int32, vp, C.c_int32, C.c_int64, vpp],
    'bc_project_from_host': [vp, vp, C.c_int64, C.c_int32, C.c_int, vp, C.c_int32, vp, C.c_int32, C.c_int64, vpp, vpp],
    'bc_project_grad_x': [vp, vp, C.c_int, vp, C.c_int32, vp, C.c_int32, vp],
    'bc_project_colsum': [vp, vp, C.c_int, vp, C.c_int32, vp, C.c_int32, vp, vp],
    'bc_vi_gradient': [vp, vp, vp, C.c_int64, C.c_int, vp, C.c_int32, vp, C.c_int32, vp, C.c_double, vp, vp, vp],
    'bc_vi_gradient_begin': [vp, vp, vp, C.c_int64, C.c_int, vp, C.c_int32, vp, C.c_int32, vp, C.c_double, vp],
    'bc_vi_gradient_end': [vp, vp, vp],
    'bc_phi_shape': [vp, c_i64p, c_i32p, c_i64p],
    'bc_phi_colsum': [vp, vp],
    'bc_phi_norms': [vp, vp],
    'bc_phi_norm_stats': [vp, c_i64p, c_dp],
    'bc_phi_to_host': [vp, vp],
    'bc_phi_group_sum': [vp, vp, vp, C.c_int64, vpp],
    'bc_phi_gather_rows': [vp, vp, C.c_int64, vp],
    'bc_phi_matvec': [vp, vp, vp],
    'bc_phi_destroy': [vp],
    'bc_phi_argmax': [vp, C.c_int, vp, C.c_double, c_i64p, c_dp],
    'bc_snnls_create': [vp, vp, vp, C.c_int, C.c_double, C.c_int, vpp],
    'bc_snnls_destroy': [vp],
    'bc_snnls_set_tolerance': [vp, C.c_double],
    'bc_comm_load': [C.c_char_p],
    'bc_comm_unique_id': [vp, C.c_int32],
    'bc_comm_create': [vp, vp, C.c_int32, C.c_int32, vpp],
    'bc_comm_destroy': [vp],
    'bc_comm_info': [vp, c_i32p, c_i32p],
    'bc_comm_all_gather': [vp, vp, vp, C.c_int64],
    'bc_comm_selftest': [vp],
    'bc_comm_precheck': [vp],
    'bc_comm_abort': [vp],
    'bc_comm_sum_doubles': [vp, vp, C.c_int64, vp],
    'bc_comm_rank_order_sum_selftest': [vp, vp, C.c_int32, C.c_int64, vp],
    'bc_phi_colsum_all': [vp, vp, vp],
    'bc_snnls_bind_comm': [vp, vp],
    'bc_snnls_prefilter_active': [vp, c_ip],
    'bc_snnls_prefilter_form': [vp, c_ip],
    'bc_snnls_prefilter_fallbacks': [vp, C.POINTER(C.c_int64)],
    'bc_snnls_prefilter_stats': [vp, c_i64p, c_i64p, c_i64p],
    'bc_snnls_prefilter_levels': [vp, c_i64p, c_i64p, c_i64p],
    'bc_snnls_bind_exchange': [vp, C.c_int, vp, vp],
    'bc_snnls_record_doubles': [vp, c_i32p],
    'bc_snnls_build_begin': [vp, C.c_int],
    'bc_snnls_step_local': [vp],
    'bc_snnls_step_local_exact': [vp],
    'bc_snnls_step_finish': [vp],
    'bc_snnls_build_end': [vp, c_ip, c_ip, c_ip],
    'bc_snnls_build': [vp, C.c_int, c_ip],
    'bc_snnls_select': [vp, c_i64p],
    'bc_snnls_select_local': [vp],
    'bc_snnls_select_local_exact': [vp],
    'bc_snnls_select_pick': [vp, c_i64p],
    'bc_snnls_reweight': [vp, C.c_int64],
    'bc_snnls_error': [vp, c_dp],
    'bc_snnls_size': [vp, c_i64p],
    'bc_snnls_weights': [vp, C.c_int64, vp, vp, c_i64p],
    'bc_snnls_set_weights': [vp, C.c_int64, vp, vp, vp],
    'bc_snnls_columns': [vp, C.c_int64, vp, c_i64p],
    'bc_snnls_reset': [vp],
    'bc_snnls_get_flags': [vp, c_ip],
    'bc_snnls_set_flags': [vp, C.c_int],
    'bc_snnls_trace': [vp, C.c_int64, vp, vp, vp, c_i64p],
    'bc_weighted_gram': [vp, vp, vp, vp, vp],
    'bc_weighted_gram_host': [vp, vp, C.c_int64, C.c_int32, vp, vp, vp],
}
EXPORTS = sorted(list(_SIGNATURES) + ['bc_version', 'bc_last_error'])

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's).  A process must run ONE HIP
    runtime: if ours resolved to the system copy first, a later `import torch` would find a runtime it was
    not built against ("no ROCm-capable device is detected").  So when torch is installed, map its copy
    first -- without importing torch -- and let the loader hand that one to libbeta_cores.so as well."""
    try:
        import importlib.util
        spec = importlib.util.find_spec('torch')
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load the shared library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'beta_cores_amd: %s is missing -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C beta_cores_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.bc_version.restype = C.c_int
    lib.bc_version.argtypes = []
    lib.bc_last_error.restype = C.c_char_p
    lib.bc_last_error.argtypes = []
    _lib = lib
    return lib


def last_error():
    return load().bc_last_error().decode('utf-8', 'replace')


def check(status):
    """Translate a C status into the reference's exception conventions."""
    if status == BC_OK:
        return
    msg = last_error()
    if status == BC_NUMERICAL_PRECISION:
        raise NumericalPrecisionError(msg)
    if status == BC_INVALID_ARGUMENT:
        raise ValueError(msg)
    raise RuntimeError('beta_cores HIP failure (%d): %s' % (status, msg))


def call(name, *args):
    check(getattr(load(), name)(*args))
