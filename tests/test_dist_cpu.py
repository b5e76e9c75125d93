"""Multi-process (world_size 2, gloo, CPU) tests of the sharded path: shard arithmetic, the
rank-order reductions, and the per-step candidate exchange protocol driven through the real
SparseNNLS host classes with a NumPy model of the device engine (tests/fake_engine.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(mode, tmp_path, world=2, timeout=240):
    out = str(tmp_path / mode)
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS='2', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), mode, out],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=timeout)
            logs.append(o.decode('utf-8', 'replace'))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, 'rank %d failed:\n%s' % (r, logs[r][-3000:])
    return [np.load(out + '.rank%d.npz' % r) for r in range(world)]


def test_shard_bounds():
    from beta_cores_amd.dist import shard_bounds
    for n in (1, 127, 128, 129, 1000, 10_000_000):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and len(b) == w + 1
            assert all(b[i] <= b[i + 1] for i in range(w))
            assert all(x % 128 == 0 or x == n for x in b)             # tiles never straddle ranks
            sizes = np.diff(b)
            assert sizes.max() - sizes.min() <= 256
    assert shard_bounds(10_000_000, 8)[1] == 1_250_048


def test_comm_reductions(tmp_path):
    r0, r1 = launch('comm', tmp_path)
    v0 = np.arange(5.) * 1
    v1 = np.arange(5.) * 2 + 0.1
    for r in (r0, r1):
        assert np.array_equal(r['sum'], v0 + v1)
        assert np.array_equal(r['gather'], np.stack((v0, v1)))
        assert int(r['total']) == 201
    assert int(r0['offset']) == 0 and int(r1['offset']) == 100


@pytest.mark.parametrize('mode', ['fake_giga', 'fake_fw', 'fake_giga_stepwise'])
def test_sharded_solver_protocol_matches_single_rank_oracle(tmp_path, mode):
    from dist_worker import problem
    from oracle import RefGIGA, RefFrankWolfe
    phi = problem()
    ref = (RefFrankWolfe if mode == 'fake_fw' else RefGIGA)(phi.T, phi.sum(axis=0))
    ref.build(30)
    ridx = np.where(ref.w > 0)[0]
    r0, r1 = launch(mode, tmp_path)
    for r in (r0, r1):
        np.testing.assert_array_equal(r['idx'], ridx)                 # global indices, both ranks
        np.testing.assert_allclose(r['val'], ref.w[ridx], rtol=1e-9)
        np.testing.assert_allclose(r['w_dense'], ref.w, rtol=1e-9, atol=1e-15)
        np.testing.assert_allclose(float(r['err']), ref.error(), rtol=1e-9)
    assert np.array_equal(r0['val'], r1['val'])                       # replicated state is bit-identical


@pytest.mark.parametrize('world', [4, 8])
def test_sharded_solver_protocol_at_more_ranks(tmp_path, world):
    """The same exchange protocol at the world sizes the scaling bench runs (4, 8): every rank ends with the single-rank oracle's
    coreset, the replicated state bit-identical on all of them (and therefore independent of the world size)."""
    from dist_worker import problem
    from oracle import RefGIGA
    phi = problem()
    ref = RefGIGA(phi.T, phi.sum(axis=0))
    ref.build(30)
    ridx = np.where(ref.w > 0)[0]
    res = launch('fake_giga', tmp_path, world=world, timeout=400)
    for r in res:
        np.testing.assert_array_equal(r['idx'], ridx)
        np.testing.assert_allclose(r['val'], ref.w[ridx], rtol=1e-9)
        np.testing.assert_allclose(float(r['err']), ref.error(), rtol=1e-9)
        assert np.array_equal(r['val'], res[0]['val'])
