"""bc_np_sum.h (the mean of a constant row with NumPy's own rounding: DESIGN section 7, golden F12) compiled for the host and
compared with np.full(n, c).sum() bit for bit: every n up to 3000, random n up to 2M, random constants of all magnitudes."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constant_sums_equal_numpy_bits(tmp_path):
    exe = str(tmp_path / 'np_sum_harness')
    cmd = ['gcc', '-O2', '-ffp-contract=off', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'beta_cores_amd', 'csrc'),
           os.path.join(ROOT, 'tests', 'np_sum_harness.c'), '-o', exe]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    rng = np.random.RandomState(4)
    ns = list(range(1, 3001)) + [4096, 4097, 10_000, 65_537, 100_000, 1_000_003, 2_000_000] + list(rng.randint(3001, 300_000, 300))
    rows = []
    for n in ns:
        for c in (rng.randn() * 10.0 ** rng.uniform(-6, 6), -0.6931471805599453, 1.0 / 3.0):
            rows.append((float(n), c, float(np.full(n, c).sum())))
    t = np.array(rows)
    path = str(tmp_path / 't.bin')
    t.tofile(path)
    res = subprocess.run([exe, path, str(t.shape[0])], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'mismatches=0' in res.stdout
