"""bc_np_pow2.h restates, for base 2, the routine NumPy evaluates float64 np.power with on AVX512_SKX hosts (Intel SVML's
__svml_pow8_ha, bundled with NumPy), so that the C layer can give the constant projection row of a data row z = 0 under
the logistic beta-likelihood (model_lr.py:85) the reference's bits without NumPy (golden F20).  The
header is compiled for the host and compared with np.power(2., y) of the running NumPy, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden, numpy_uses_svml_exp as _numpy_uses_svml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / 'np_pow2_harness')
    cmd = ['gcc', '-O2', '-mfma', '-ffp-contract=off', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'beta_cores_amd', 'csrc'),
           os.path.join(ROOT, 'tests', 'np_pow2_harness.c'), '-o', exe, '-lm']
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


@pytest.mark.skipif(not _numpy_uses_svml(), reason='this NumPy / CPU does not evaluate np.power with SVML (no AVX512_SKX dispatch)')
def test_header_reproduces_numpy_power_of_two_bits(tmp_path):
    exe = build(tmp_path)
    rng = np.random.RandomState(2)
    n = 500_000
    y = np.concatenate([-rng.uniform(0, 4, n), -rng.uniform(0, 1.2, n), rng.uniform(-1000, 1000, n), rng.normal(0, 1e-3, n),
                        -np.arange(0, 64) / 16., np.array([0., -0., -0.1, -1.1, -0.2, -1.2, -0.5, -1.5, -1., -2., 1021.5, -1021.5, 1e-300, 2000., np.nan])])
    with np.errstate(over='ignore', under='ignore'):
        w = np.power(np.full(y.shape, 2.0), y)
    path = str(tmp_path / 'd.bin')
    with open(path, 'wb') as f:
        f.write(y.tobytes())
        f.write(w.tobytes())
    res = subprocess.run([exe, path, str(y.shape[0])], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and 'mismatches=0' in res.stdout, res.stdout + res.stderr
    assert int(res.stdout.split('not_covered=')[1]) <= 4          # |y| > 1021.5 and NaN are left to the ordinary pow


def test_zero_row_constant_from_the_restated_powers_matches_the_goldens():
    """c(beta) = -((b+1)/b 2^-b - (2^(-b-1) + 2^(-b-1))) through bc_np_pow2 gives, centred with NumPy's mean, the residues golden
    F20 holds for its z = 0 rows (on any host: the header is plain arithmetic; the golden carries the AVX-512 NumPy's bits)."""
    import ctypes
    import tempfile
    src = '#include "bc_np_pow2.h"\ndouble p2(double y) { int c; double v = bc_np_pow2(y, &c); return c ? v : pow(2.0, y); }\n'
    g = load_golden('f20_logistic_beta_constant_rows')
    with tempfile.TemporaryDirectory() as tmp:
        cfile, so = os.path.join(tmp, 'p.c'), os.path.join(tmp, 'p.so')
        open(cfile, 'w').write(src)
        subprocess.check_call(['gcc', '-O2', '-mfma', '-ffp-contract=off', '-shared', '-fPIC', '-I', os.path.join(ROOT, 'beta_cores_amd', 'csrc'),
                               cfile, '-o', so, '-lm'])
        lib = ctypes.CDLL(so)
        lib.p2.restype, lib.p2.argtypes = ctypes.c_double, [ctypes.c_double]
        for S in (16, 100, 200):
            for beta in (0.1, 0.2, 0.5):
                c = -(((beta + 1.) / beta) * lib.p2(-beta) - (lib.p2(-beta - 1.) + lib.p2(-beta - 1.)))
                row = np.full((1, S), c)
                row -= row.mean(axis=1)[:, np.newaxis]
                np.testing.assert_array_equal(row[0], g['S%d_b%g_phi_const' % (S, beta)][0])
