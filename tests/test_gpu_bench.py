"""bench.py keeps its contract: one JSON line with the driver's keys, the roofline and cpu_baseline objects,
and a CPU-parity verdict -- on a small workload so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '12', '--warmup', '3',
           '--rows', '300000', '--dim', '24', '--samples', '40', '--cpu-sample', '30000', '--cpu-iters', '6'] + list(extra)
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, 'bench must print exactly one line: %r' % lines
    return json.loads(lines[0])


def test_bench_line_contract():
    d = run_bench()
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 12 and d['warmup'] == 3 and d['higher_is_better'] is True
    assert d['vs_baseline'] is None and d['scaling'] == 'strong' and 'workload' in d['config']
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert r['bound'] == 'hbm' and 0 < r['frac'] < 1 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['parity_on_sample'].startswith('ok')
    assert abs(d['value'] - 1e3 / d['ms_per_step']) < 1e-6 * d['value']
    assert d['config']['sweep'].startswith('int8') and d['prefilter']['fp64_fallbacks'] == 0
    # round 4: the line is compact (the driver's record keeps `roofline` and `cpu_baseline` whole, everything else by name only), so
    # the evidence sits INSIDE roofline: one short entry per timed kernel, the beta-Cores gradient loops (configs 2, 3 -- logistic with
    # the Laplace sampler -- and the headline shape), the host-array path; the oracle's loop timings inside cpu_baseline
    ks = r['kernels']
    names = ' | '.join(e['k'] for e in ks)
    for must in ('K1 k_project', 'headline', 'K3 k_sweep<GIGA> fp64', 'K4 k_gram+reduce', 'store-free beta-linreg', 'store-free beta-logistic',
                 'logistic N=1M D=128 log-likelihood', 'logistic N=1M D=128 beta-likelihood', 'D=512'):
        assert must in names, (must, names)
    for e in ks:
        assert e['ms'] > 0 and e['GB'] > 0 and 0 < e['hbm'] < 1.2 and e['n'] >= 1, e
        if 'mfma' in e:
            assert 0 < e['mfma'] < 1, e
    lp = r['loops']
    assert len(lp) == 7 and sum('logistic/newton' in e['loop'] for e in lp) == 2 and sum('logistic/bfgs' in e['loop'] for e in lp) == 1
    assert all(e['ms_grad'] > 0 and e['k1_ms'] > 0 and 0 <= e['non_k1'] < 1 and e['build_ms'] > 0 for e in lp)
    bf = [e for e in lp if 'logistic/bfgs' in e['loop']][0]
    nw = [e for e in lp if 'logistic/newton' in e['loop'] and 'M=100' in e['loop']][0]
    assert nw['samp_ms'] < bf['samp_ms'] and nw['ms_grad'] < bf['ms_grad']      # (K1 times differ with the clock the GPU holds: idle gaps between BFGS gradients)
    fh = r['from_host']
    assert fh['same_trace_as_resident_run'] is True and fh['first_iter_ms'] >= fh['construct_ms'] > 0 and fh['M100_ms'] >= fh['first_iter_ms']
    assert fh['upload_GBps'] > 0 and fh['copy_direct_GBps'] > 0 and fh['copy_staged8_GBps'] > 0
    assert len(c['loops']) == 2 and all(e['ms_grad'] > 0 for e in c['loops'])
    # round 5: the SURVEY 8(d) formulation as an object of its own, the M = 100 step time, and the baseline's host description
    f64 = r['fp64_formulation']
    assert f64['ms'] > 0 and 0 < f64['frac'] < 1 and f64['it_s'] > 0 and f64['same_sel'] is True
    assert d['ms_per_step_M100'] > 0 and abs(d['ms_per_step_M100'] - (fh['M100_ms'] - fh['first_iter_ms']) / 99.) < 1e-3
    assert isinstance(c['cpu_model'], str) and c['cpu_model'] and c['one_thread']['cores'] == 1
    assert c['one_thread']['value'] > 0 and c['one_thread']['same_selections'] is True
    st = d['step_stages']
    assert st['transport'].startswith('none') and st['sweep'] > 0 and st['finish'] > 0
    assert abs(sum(d['solver_init'].values()) - d['solver_init_ms']) <= 0.05 * d['solver_init_ms'] + 0.5
    assert r['traffic_source'] is None or r['traffic_source'].startswith('profiles/r')
    assert len(json.dumps(d, separators=(',', ':'))) < 9000              # short enough for the driver's tail


def test_bench_detail_goes_to_a_file_or_stderr(tmp_path):
    path = str(tmp_path / 'detail.json')
    d = run_bench('--no-cpu', '--no-extra', '--no-host', '--detail', path)
    full = json.load(open(path))
    assert full['line']['value'] == d['value'] and 'solver_init' in full['detail']


def test_bench_sweep_modes_agree():
    a = run_bench('--no-cpu', env={'BC_PREFILTER': '0'})
    b = run_bench('--no-cpu', env={'BC_PREFILTER': '8'})
    assert a['coreset'] == b['coreset']            # same size, same error, same number of failed steps
    assert a['config']['sweep'] == 'fp64' and b['config']['sweep'].startswith('int8')
    # the two-level form (a 4-bit first level; csrc/bc_prefilter_i4.h: the default from 2M rows, forced here), named in the line
    # with its bytes: S = 40 -> five dwords of eight nibbles + a 16-bit code per row
    c = run_bench('--no-cpu', '--no-extra', '--no-host', env={'BC_PREFILTER': '4'})
    assert c['coreset'] == a['coreset'] and c['config']['sweep'].startswith('4-bit') and c['prefilter']['form'] == 3
    lv = c['prefilter']['levels']
    assert lv['l1_sweeps'] >= c['steps'] and 0 < lv['rows_passed_on'] == lv['rows_refined_int8']
    assert c['roofline']['kernel'].startswith('k_sweep_i4') and abs(c['roofline']['bytes_per_launch'] - 22.0 * c['config']['N']) < 1


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver may call it) starts two fresh rank
    processes through torch.distributed.run and relays rank 0's line.  Rehearsed here with both ranks on the one
    GPU this box has and gloo carrying the records (RCCL refuses two ranks per device)."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '10', '--warmup', '2',
           '--rows', '400000', '--dim', '24', '--samples', '40', '--no-cpu', '--no-extra']
    e = dict(os.environ, BC_BENCH_DEVICE='0', BC_BENCH_BACKEND='gloo')
    e.pop('WORLD_SIZE', None)
    e.pop('RANK', None)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 10 and d['config']['rows_per_gpu'] == 200064 and d['value'] > 0
    one = run_bench('--no-cpu', '--no-extra', '--rows', '400000', '--steps', '10', '--warmup', '2')
    assert one['coreset']['size'] == d['coreset']['size']
    assert abs(one['coreset']['error'] - d['coreset']['error']) <= 1e-9 * one['coreset']['error']
