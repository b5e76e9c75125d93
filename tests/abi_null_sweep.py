"""Call EVERY entry point of include/beta_cores.h with NULL / zero arguments and with a few malformed ones (driven by
tests/test_sanitized_cpu.py against a host-sanitized build, and by tests/test_abi_cpu.py against the shipped library): the
argument-validation paths must hand back a status -- BC_INVALID_ARGUMENT, or a HIP failure on a box without a GPU -- never
touch memory through the NULL handles, and (under ASan / UBSan) leave no report.  Prints `swept <n> entry points`."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beta_cores_amd import _native as N           # noqa: E402  (reads BETA_CORES_LIB)

lib = N.load()
# entry points for which "nothing to do" is a legitimate answer to all-NULL arguments
MAY_SUCCEED = {'bc_ctx_destroy', 'bc_data_destroy', 'bc_phi_destroy', 'bc_snnls_destroy', 'bc_comm_destroy', 'bc_comm_load',
               'bc_comm_abort'}


def zero(t):
    if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
        return None
    if t in (C.c_double, C.c_float):
        return t(0.0)
    return t(0)


swept = 0
for name, argtypes in sorted(N._SIGNATURES.items()):
    fn = getattr(lib, name)
    rc = fn(*[zero(t) for t in argtypes])
    if rc == 0 and name not in MAY_SUCCEED:
        raise SystemExit('%s accepted all-NULL arguments (status 0)' % name)
    if rc != 0 and not lib.bc_last_error():
        raise SystemExit('%s returned %d without a message' % (name, rc))
    swept += 1

# a context cannot exist without a GPU; where one does exist (the -m gpu suite runs this file too) malformed sizes on a live
# context must be refused as well
ctx = C.c_void_p()
rc = lib.bc_ctx_create(0, None, C.byref(ctx))
if rc == 0:
    import numpy as np
    z = np.zeros((4, 3))
    h = C.c_void_p()
    bad = [
        ('bc_data_from_host', (ctx, z.ctypes.data_as(C.c_void_p), -1, 3, C.byref(h))),
        ('bc_data_from_host', (ctx, z.ctypes.data_as(C.c_void_p), 4, 0, C.byref(h))),
        ('bc_data_from_host', (ctx, None, 4, 3, C.byref(h))),
        ('bc_phi_create', (ctx, -5, 10, C.byref(h))),
        ('bc_phi_create', (ctx, 10, 0, C.byref(h))),
        ('bc_project_from_host', (ctx, z.ctypes.data_as(C.c_void_p), 4, 3, 99, z.ctypes.data_as(C.c_void_p), 2, None, 0, 0, C.byref(h), C.byref(h))),
        ('bc_project_from_host', (ctx, z.ctypes.data_as(C.c_void_p), 0, 3, 0, z.ctypes.data_as(C.c_void_p), 2, None, 0, 0, C.byref(h), C.byref(h))),
        ('bc_weighted_gram_host', (ctx, z.ctypes.data_as(C.c_void_p), 4, 1, None, z.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p))),
    ]
    for name, args in bad:
        if getattr(lib, name)(*args) != N.BC_INVALID_ARGUMENT:
            raise SystemExit('%s%r was not refused' % (name, args[2:5]))
    lib.bc_ctx_destroy(ctx)
    print('live-context checks ok')
print('swept %d entry points' % swept)
