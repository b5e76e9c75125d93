"""Fuzz (not collected by pytest): random small problems, fused device loop vs step-wise protocol (must be bit-identical)
and vs the NumPy oracle (selections identical, weights within 1e-5).  Differences from the oracle are reported unless the
target was already matched to 1e-9 (both sides then choose among rounding noise).  What remains are exact-tie geometries:
S = 2 (all centred rows are multiples of (1, -1)) and S = 3 (rows in a plane), where the argmax is decided by the last bit of
a BLAS dot product.  Usage: python tests/fuzz/fuzz_paths.py SEED SECONDS (needs a GPU)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import beta_cores_amd as bc
from oracle import RefGIGA, RefFrankWolfe
bc.default_context()
rng = np.random.RandomState(int(sys.argv[1]))
t_end = time.time() + float(sys.argv[2])
cases = bad = 0
while time.time() < t_end:
    n = int(10 ** rng.uniform(0.5, 3.7)); s = rng.randint(2, 80)
    r = max(1, s // 5)
    phi = rng.randn(n, r).dot(rng.randn(r, s)) + 0.3 * rng.randn(n, s)
    phi -= phi.mean(axis=1)[:, None]
    if (np.linalg.norm(phi, axis=1) == 0).any():
        continue
    giga = rng.rand() < 0.6
    cls, ref_cls = (bc.snnls.GIGA, RefGIGA) if giga else (bc.snnls.FrankWolfe, RefFrankWolfe)
    steps = min(n, rng.randint(1, 30))
    pref = rng.choice([0, 4, 8, 16])        # (4: the two-level form, csrc/bc_prefilter_i4.h)
    os.environ['BC_PREFILTER'] = str(pref)
    os.environ['BC_FINISH_NOPF'] = str(rng.randint(2))
    a = cls(phi.T, phi.sum(axis=0)); a.build(steps)
    b = cls(phi.T, phi.sum(axis=0)); b.build_stepwise(steps)
    ref = ref_cls(phi.T, phi.sum(axis=0)); ref.build(steps)
    ia, va = a._eng.sparse_weights(); ib, vb = b._eng.sparse_weights()
    ok = np.array_equal(ia, ib) and np.array_equal(va, vb) and a.error() == b.error()
    wr = ref.w
    ridx = np.where(wr != 0)[0] if False else None
    dense = np.zeros(n); dense[ia] = va
    sel_dev = a._eng.trace()[0]; sel_ref = np.array([t[0] for t in ref.trace])
    ok2 = np.array_equal(sel_dev, sel_ref) and np.allclose(dense, ref.w, rtol=1e-5, atol=1e-12)
    if not (ok and ok2):
        k = next((i for i in range(min(len(sel_dev), len(sel_ref))) if sel_dev[i] != sel_ref[i]), min(len(sel_dev), len(sel_ref)))
        errs = a._eng.trace()[2]
        rel = (errs[k - 1] / np.linalg.norm(phi.sum(axis=0))) if k > 0 else 1.0
        if rel < 1e-9:
            cases += 1
            continue            # both are choosing among rounding noise: the target is already matched to ~1e-9
        print('first difference at step', k, 'relative error before it %.3e' % rel, 'scores tie?')
        bad += 1
        print('MISMATCH n=%d s=%d giga=%s steps=%d pref=%d fused==stepwise %s, ==oracle %s' % (n, s, giga, steps, pref, ok, ok2))
    cases += 1
print('fuzz paths: %d problems, %d mismatches' % (cases, bad))
sys.exit(1 if bad else 0)
