"""Sanitized CPU leg (`-fsanitize=address,undefined` for the host-side unit tests): the host-compiled
restatements (NumPy's pairwise sum, SVML exp, the K1 epilogue's table-driven bodies) run under ASan + UBSan, and so does a
host model of the ragged-end index arithmetic (tests/layout_harness.c over csrc/bc_layout.h, which the kernels and the
allocation code use too): every address the int8 mirror builders and the chunked projection form for
n in 1..1025, odd tile counts and the row shards of shard_bounds(10M, 1/2/4/8) must stay inside allocations of exactly the
device sizes -- the class of round 3's GPU fault, caught here without a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ['-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-fno-omit-frame-pointer', '-g']
ENV = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')


def build(tmp_path, src, extra=()):
    exe = str(tmp_path / (os.path.splitext(src)[0] + '_san'))
    cmd = ['gcc', '-O1', '-mfma', '-ffp-contract=off', '-Wall', '-Werror'] + SAN + list(extra) + \
          ['-I', os.path.join(ROOT, 'beta_cores_amd', 'csrc'), os.path.join(ROOT, 'tests', src), '-o', exe, '-lm']
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def build_cxx(tmp_path, src, extra=()):
    exe = str(tmp_path / (os.path.splitext(src)[0] + '_san'))
    cmd = ['g++', '-std=c++17', '-O1', '-mfma', '-ffp-contract=off', '-Wall', '-Werror', '-Wno-unused-function', '-Wno-unknown-pragmas'] + SAN + list(extra) + \
          ['-I', os.path.join(ROOT, 'beta_cores_amd', 'csrc'), os.path.join(ROOT, 'tests', src), '-o', exe, '-lm']
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return exe


def run_clean(exe, args, timeout=600):
    res = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=timeout, env=ENV)
    assert 'AddressSanitizer' not in res.stderr and 'runtime error' not in res.stderr, res.stderr[-3000:]
    return res


def test_k1_math_bodies_under_asan_ubsan(tmp_path):
    """bc_k1_math.h incl. NaNs with payloads: their mantissa bits must not index past the 257-entry log table."""
    res = run_clean(build(tmp_path, 'k1_math_harness.c'), ['200000'])
    assert res.returncode == 0 and 'special ok' in res.stdout, res.stdout + res.stderr


def test_sweep_vector_quantisers_under_asan_ubsan(tmp_path):
    """bc_i4_quant.h / bc_i8_quant.h compiled for the host: digit ranges, Q = 16 d0 + d1 (128 d0 + d1), the half-step error of
    every element and the header bounds the sweeps' intervals rest on -- 4 000 random vectors, S = 1..260, both modes."""
    res = run_clean(build_cxx(tmp_path, 'i4_quant_harness.cpp'), ['2000'])
    assert res.returncode == 0 and res.stdout.startswith('ok 4000'), res.stdout + res.stderr


def test_np_sum_restatement_under_asan_ubsan(tmp_path):
    import numpy as np
    rng = np.random.RandomState(4)
    ns = list(range(1, 600)) + [4096, 4097, 8191, 8192, 8193, 65_537, 100_000, 300_007]
    rows = [(float(n), c, float(np.full(n, c).sum())) for n in ns for c in (rng.randn(), -0.6931471805599453)]
    t = np.array(rows)
    path = str(tmp_path / 't.bin')
    t.tofile(path)
    res = run_clean(build(tmp_path, 'np_sum_harness.c'), [path, str(t.shape[0])])
    assert res.returncode == 0 and 'mismatches=0' in res.stdout, res.stdout + res.stderr


def test_np_exp_restatement_under_asan_ubsan(tmp_path):
    """(bit equality with np.exp is test_np_exp_cpu's business and needs an AVX-512 NumPy; here only: no out-of-range table
    index, shift or conversion for any argument class, specials included)"""
    import numpy as np
    rng = np.random.RandomState(1)
    x = np.concatenate([rng.uniform(-745, 710, 50_000), rng.normal(0, 1e-3, 10_000),
                        np.array([0., -0., 1e-300, -1e-300, -707.7, 707.7, -745., 800., -1e308, 1e308, np.inf, -np.inf, np.nan])])
    with np.errstate(over='ignore'):
        y = np.exp(x)
    path = str(tmp_path / 'd.bin')
    with open(path, 'wb') as f:
        f.write(x.tobytes())
        f.write(y.tobytes())
    res = run_clean(build(tmp_path, 'np_exp_harness.c'), [path, str(x.shape[0])])
    assert 'mismatches=' in res.stdout, res.stdout + res.stderr


def test_np_pow2_restatement_under_asan_ubsan(tmp_path):
    import numpy as np
    rng = np.random.RandomState(7)
    y = np.concatenate([-rng.uniform(0, 4, 40_000), rng.uniform(-1021, 1021, 20_000), -np.arange(0, 64) / 16.,
                        np.array([0., -0., 1021.5, -1021.5, 1021.6, 1e300, -1e300, np.inf, -np.inf, np.nan])])
    with np.errstate(over='ignore', under='ignore'):
        w = np.power(np.full(y.shape, 2.0), y)
    path = str(tmp_path / 'p.bin')
    with open(path, 'wb') as f:
        f.write(y.tobytes())
        f.write(w.tobytes())
    res = run_clean(build(tmp_path, 'np_pow2_harness.c'), [path, str(y.shape[0])])
    assert 'mismatches=' in res.stdout, res.stdout + res.stderr


def shard_sizes():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bc_dist', os.path.join(ROOT, 'beta_cores_amd', 'dist.py'))
    src = open(os.path.join(ROOT, 'beta_cores_amd', 'dist.py')).read()
    # shard_bounds is pure Python; pick it out without importing the package (no GPU library here)
    ns = {}
    start = src.index('def shard_bounds')
    end = src.index('\ndef ', start + 10) if '\ndef ' in src[start + 10:] else len(src)
    end2 = src.index('\nclass ', start) if '\nclass ' in src[start:] else len(src)
    exec('import numpy as np\nTILE_ROWS = 128\n' + src[start:min(end, end2)], ns)
    sizes = set()
    for N in (10_000_000, 1_000_000, 2_000_000, 1234567):
        for G in (1, 2, 3, 4, 8):
            b = ns['shard_bounds'](N, G)
            assert b[0] == 0 and b[-1] == N and all(b[i] <= b[i + 1] for i in range(G))
            assert all(b[i] % 128 == 0 for i in range(G)), 'shards start on tile boundaries'
            sizes.update(int(b[i + 1] - b[i]) for i in range(G))
    return sorted(sizes)


def test_index_arithmetic_host_model(tmp_path):
    exe = build(tmp_path, 'layout_harness.c')
    small = list(range(0, 300)) + [128 * k + j for k in (3, 4, 5, 6, 7, 8, 9, 15, 234, 235) for j in (-1, 0, 1)] + [1025, 65537]
    res = subprocess.run([exe] + [str(n) for n in small], capture_output=True, text=True, timeout=900, env=ENV)
    assert res.returncode == 0 and res.stdout.startswith('ok '), (res.stdout + res.stderr)[-3000:]
    big = [n for n in shard_sizes() if n > 70000] + [262144, 262145, 300003, 327697, 600001]
    res = subprocess.run([exe] + [str(n) for n in big], capture_output=True, text=True, timeout=900, env=ENV)
    assert res.returncode == 0 and res.stdout.startswith('ok '), (res.stdout + res.stderr)[-3000:]


def test_model_catches_the_round3_fault(tmp_path):
    """The same replay with the clamp of bc_lay_i8_src_row removed (what the first one-pass builder did) must trip ASan on an
    odd tile count: the model is only worth something if it fails when the arithmetic is wrong."""
    exe = build(tmp_path, 'layout_harness.c', extra=['-DBC_LAY_TEST_NO_CLAMP'])
    res = subprocess.run([exe, str(128 * 234 + 1)], capture_output=True, text=True, timeout=300, env=ENV)
    assert res.returncode != 0 and 'AddressSanitizer' in res.stderr


def test_c_abi_validation_paths_under_asan_ubsan(tmp_path):
    """The whole library's HOST code built with -fsanitize=address,undefined (hipcc --offload-host-only: no device code is
    generated, zero-filled stand-ins take the place of the eight code-object symbols; ~6 s) and every entry point of
    include/beta_cores.h called with NULL / zero arguments through the same ctypes table the product uses
    (tests/abi_null_sweep.py): a status comes back from each, nothing is dereferenced, no sanitizer report."""
    import glob
    import sys
    hipcc = '/opt/rocm/bin/hipcc'
    asan = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    if not os.path.exists(hipcc) or not asan:
        pytest.skip('hipcc / clang ASan runtime not found')
    csrc = os.path.join(ROOT, 'beta_cores_amd', 'csrc')
    srcs = sorted(glob.glob(os.path.join(csrc, 'bc_*.hip')))
    procs = []
    for src in srcs:
        obj = str(tmp_path / (os.path.basename(src)[:-4] + '.o'))
        procs.append((obj, subprocess.Popen([hipcc, '-O1', '-g', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '--offload-host-only',
                                             '-ffp-contract=off', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined',
                                             '-I', csrc, '-c', src, '-o', obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for obj, p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
        objs.append(obj)
    lib0 = str(tmp_path / 'libbc_san0.so')
    out = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-fsanitize=address,undefined', '-shared-libsan', '-o', lib0] + objs +
                         ['-ldl', '-lpthread'], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    syms = sorted(set(l.split()[-1] for l in subprocess.run(['nm', '-u', lib0], capture_output=True, text=True).stdout.splitlines()
                      if '__hip_fatbin_' in l))
    assert syms, 'a host-only build refers to its (absent) device code objects'
    stub_c = str(tmp_path / 'fatbin_stubs.c')
    with open(stub_c, 'w') as f:
        f.write(''.join('__attribute__((aligned(4096))) const char %s[4096] = {0};\n' % s for s in syms))
    stub_o = str(tmp_path / 'fatbin_stubs.o')
    assert subprocess.run(['gcc', '-fPIC', '-c', stub_c, '-o', stub_o]).returncode == 0
    lib = str(tmp_path / 'libbc_san.so')
    out = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-fsanitize=address,undefined', '-shared-libsan', '-o', lib] + objs +
                         [stub_o, '-ldl', '-lpthread'], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    env = dict(os.environ, LD_PRELOAD=asan[0], ASAN_OPTIONS='detect_leaks=0', UBSAN_OPTIONS='print_stacktrace=1', BETA_CORES_LIB=lib)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'abi_null_sweep.py')], capture_output=True, text=True, timeout=300, env=env)
    assert 'AddressSanitizer' not in res.stderr and 'runtime error' not in res.stderr, res.stderr[-3000:]
    assert res.returncode == 0 and 'swept 8' in res.stdout, res.stdout + res.stderr[-2000:]
