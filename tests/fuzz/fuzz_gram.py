"""Fuzz (not collected by pytest): random shapes through K4 -- the coreset-sized one-launch path (k_gram_small), the tiled
MFMA path with triangular diagonal tiles, ndarray and DeviceData inputs, with / without weights -- against NumPy's
(w[:,None]*X).T.dot(X) and (w[:,None]*Y[:,None]*X).sum(axis=0) (model_linreg.py:29,31).
Usage: python tests/fuzz/fuzz_gram.py SEED SECONDS (needs a GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import beta_cores_amd as bc

bc.default_context()
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 60.)
cases = bad = 0
while time.time() < t_end:
    d = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 300]))
    n = int(10 ** rng.uniform(0, 4.2))
    mode = rng.randint(4)
    if rng.rand() < 0.12:            # the LDS-DMA kernel (k_gram_dma): one unweighted tile, rows of >= 128 doubles, >= 4096 rows
        d, n, mode = int(rng.choice([127, 128])), int(rng.randint(4096, 40000)), 0
    if n * (d + 1) > 6_000_000:
        continue
    Z = rng.randn(n, d + 1) * 10.0 ** rng.uniform(-2, 2)
    w = None if mode == 0 else rng.rand(n) * 3.
    if mode == 2:
        w[rng.rand(n) < 0.5] = 0.
    src = bc.DeviceData(Z) if rng.rand() < 0.5 else Z
    G, v = bc.weighted_gram(src, w)
    ww = np.ones(n) if w is None else w
    Gr = (ww[:, None] * Z[:, :d]).T.dot(Z[:, :d])
    vr = (ww[:, None] * Z[:, d][:, None] * Z[:, :d]).sum(axis=0)
    sg = max(np.abs(Gr).max(), 1e-300)
    sv = max(np.abs(vr).max(), np.sqrt(sg) * np.abs(Z[:, d]).max() * 1e-3, 1e-300)
    ok = np.abs(G - Gr).max() <= 1e-11 * sg and np.abs(v - vr).max() <= 1e-10 * sv and np.array_equal(G, G.T)
    if not ok:
        bad += 1
        print('MISMATCH n=%d d=%d mode=%d device_input=%s  dG=%.2e dv=%.2e' % (n, d, mode, not isinstance(src, np.ndarray),
                                                                              np.abs(G - Gr).max() / sg, np.abs(v - vr).max() / sv))
    cases += 1
print('fuzz gram: %d problems, %d mismatches' % (cases, bad))
sys.exit(1 if bad else 0)
