"""bc_np_exp.h restates the routine NumPy evaluates float64 np.exp with on AVX512_SKX hosts (Intel SVML's
__svml_exp8_ha, bundled with NumPy) so that K1 can give the constant rows of a beta-likelihood projection the
reference's bits (golden F13).  Here the header is compiled for the host and compared with np.exp of
the running NumPy, bit for bit, on two million arguments -- where NumPy dispatches to that routine."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


from conftest import numpy_uses_svml_exp as _numpy_uses_svml_exp


@pytest.mark.skipif(not _numpy_uses_svml_exp(), reason='this NumPy / CPU does not evaluate np.exp with SVML (no AVX512_SKX dispatch)')
def test_header_reproduces_numpy_exp_bits(tmp_path):
    exe = str(tmp_path / 'np_exp_harness')
    cmd = ['gcc', '-O2', '-mfma', '-ffp-contract=off', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'beta_cores_amd', 'csrc'),
           os.path.join(ROOT, 'tests', 'np_exp_harness.c'), '-o', exe, '-lm']
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rng = np.random.RandomState(1)
    n = 500_000
    x = np.concatenate([rng.uniform(-707, 707, n), rng.uniform(-50, 0, n), -rng.uniform(0, 1, n) ** 4 * 40,
                        rng.normal(0, 1e-3, n), np.array([0., -0., 1., -1., 1e-300, -1e-300, -707.7, 707.7, -745., 800., np.nan])])
    with np.errstate(over='ignore'):
        y = np.exp(x)
    path = str(tmp_path / 'd.bin')
    with open(path, 'wb') as f:
        f.write(x.tobytes())
        f.write(y.tobytes())
    res = subprocess.run([exe, path, str(x.shape[0])], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'mismatches=0' in res.stdout
    not_cov = int(res.stdout.split('not_covered=')[1])
    assert not_cov <= 8            # only the far tails and NaN are left to the caller's ordinary exp


def test_f13_constants_carry_the_goldens_bits():
    """The constants of golden F13's zero-feature rows (beta-likelihood of linear regression, model_neurlinr.py:102-110),
    recomputed with the restated exp through the same expression order, equal what the reference produced -- on any
    host (the header is plain arithmetic; only the comparison above needs an AVX-512 NumPy)."""
    import ctypes
    import tempfile
    from conftest import load_golden
    src = r'''
    #include "bc_np_exp.h"
    double np_exp(double x) { int c; double e = bc_np_exp(x, &c); return c ? e : exp(x); }
    '''
    with tempfile.TemporaryDirectory() as tmp:
        cfile = os.path.join(tmp, 'e.c')
        open(cfile, 'w').write(src)
        so = os.path.join(tmp, 'e.so')
        subprocess.check_call(['gcc', '-O2', '-mfma', '-ffp-contract=off', '-shared', '-fPIC', '-I',
                               os.path.join(ROOT, 'beta_cores_amd', 'csrc'), cfile, '-o', so, '-lm'])
        lib = ctypes.CDLL(so)
        lib.np_exp.restype = ctypes.c_double
        lib.np_exp.argtypes = [ctypes.c_double]
        g = load_golden('f13_greedy_vi_zero_rows')
        beta, sigsq = 0.1, 1.0
        checked = 0
        for S in (16, 100):
            Z = g['S%d_Z' % S]
            zero = np.flatnonzero((Z[:, :-1] == 0).all(axis=1))
            assert zero.size > 0
            for r in zero:
                y = Z[r, -1]
                q = y ** 2 - 2 * 0. * y + 0. ** 2
                want = 1. / (2 * np.pi * sigsq) ** (beta / 2.) * (-(beta + 1.) / beta * np.exp(-beta / (2. * sigsq) * np.array([q]))[0]
                                                                    + 1. / np.sqrt(1. + beta))
                got = 1. / (2 * np.pi * sigsq) ** (beta / 2.) * (-(beta + 1.) / beta * lib.np_exp(-beta / (2. * sigsq) * q)
                                                                   + 1. / np.sqrt(1. + beta))
                if _numpy_uses_svml_exp():
                    assert got == want, (S, r)
                checked += 1
        assert checked > 0
