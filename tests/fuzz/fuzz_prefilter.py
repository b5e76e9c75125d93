"""Fuzz (not collected by pytest): random problems through the int8 (three forms) / fp16 / fp32 pre-filter mirrors against the fp64
sweep -- traces, weights and errors must be identical.  Usage: python tests/fuzz/fuzz_prefilter.py SEED SECONDS (needs a GPU)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import numpy as np
import beta_cores_amd as bc
bc.default_context()

def make(rng, n, s, kind):
    if kind == 0:
        phi = rng.randn(n, s)
    elif kind == 1:
        r = max(1, s // 6)
        phi = rng.randn(n, r).dot(rng.randn(r, s)) + 0.2 * rng.randn(n, s)
    elif kind == 2:
        phi = rng.randn(n, s) * (rng.rand(n, s) < 0.15)
        phi[np.abs(phi).sum(axis=1) == 0, 0] = 1.0
    elif kind == 3:
        phi = rng.randn(n, s) * 10.0 ** rng.uniform(-6, 6, size=(n, 1))
    elif kind == 4:
        base = rng.randn(max(1, n // 50), s)
        phi = base[rng.randint(base.shape[0], size=n)] * (1 + 1e-6 * rng.randn(n, s))
    else:
        phi = rng.randn(n, s) * 10.0 ** rng.uniform(-5, 0, size=(n, s))
    return phi - phi.mean(axis=1)[:, None] if s > 1 else phi

FORMS = {8: ('0', '0'), 'lists': ('0', '1'), 'bb': ('1', '0')}       # int8 mirror: two-pass / + block lists / branch-and-bound


def run(cls, phi, steps, pref):
    if pref in FORMS:
        os.environ['BC_I8_BB'], os.environ['BC_I8_BLKLIST'] = FORMS[pref]
        pref = 8
    os.environ['BC_PREFILTER'] = str(pref)
    sv = cls(phi.T, phi.sum(axis=0), allow_zero_rows=True) if cls is bc.snnls.GIGA else cls(phi.T, phi.sum(axis=0))
    sv.build(steps)
    tr = sv._eng.trace()
    idx, val = sv._eng.sparse_weights()
    return tr, idx, val, sv.error(), sv._eng.prefilter_stats()

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 60
cases = bad = 0
falls = 0
while time.time() < t_end:
    n = int(10 ** rng.uniform(0, 5.3)); s = rng.randint(1, 131) if rng.rand() < 0.85 else rng.randint(131, 270); kind = rng.randint(6)
    phi = make(rng, n, s, kind)
    if not np.isfinite(phi).all() or (np.linalg.norm(phi, axis=1) == 0).any():
        continue
    cls = bc.snnls.GIGA if rng.rand() < 0.6 else bc.snnls.FrankWolfe
    steps = min(n, rng.randint(1, 40))
    try:
        ref = run(cls, phi, steps, 0)
    except ValueError:
        continue
    for pref in (8, 'lists', 'bb', 4, 16, 32):
        got = run(cls, phi, steps, pref)
        ok = all(np.array_equal(a, b) for a, b in zip(ref[0], got[0])) and np.array_equal(ref[1], got[1]) and np.array_equal(ref[2], got[2]) and ref[3] == got[3]
        falls += got[4][2]
        if not ok:
            bad += 1
            print('MISMATCH n=%d s=%d kind=%d alg=%s steps=%d pref=%s' % (n, s, kind, cls.__name__, steps, pref), ref[0][0][:10], got[0][0][:10])
    cases += 1
print('fuzz: %d problems x 6 forms (int8 two-pass, + block lists, branch-and-bound, two-level 4-bit + int8; fp16; fp32), %d mismatches, %d fp64 fallbacks' % (cases, bad, falls))
sys.exit(1 if bad else 0)
