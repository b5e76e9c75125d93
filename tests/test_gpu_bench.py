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
    # round 3: the beta-Cores gradient loop, the per-stage split of a step, the parts of solver_init, honest K4 numbers
    legs = [e for e in d['beta_coreset'] if 'ms_per_gradient' in e]
    assert len(legs) == 4 and all(e['ms_per_gradient'] > 0 and 0 < e['roofline_hbm']['frac'] < 1 and 0 < e['roofline_fp64_mfma']['frac'] < 1
                                  and e['native_gradient_calls'] == e['gradients'] for e in legs)
    assert any('cpu_baseline' in e for e in d['beta_coreset'])
    st = d['step_stages']
    assert st['transport'].startswith('none') and any(k.startswith('stage_ms') for k in st)
    assert abs(sum(d['solver_init'].values()) - d['solver_init_ms']) <= 0.05 * d['solver_init_ms'] + 0.5
    k4 = [e['posterior_gram_K4'] for e in d['other_configs'] if 'posterior_gram_K4' in e][0]
    assert all(0 < v['frac_of_fp64_mfma_peak'] < 1 and v['symmetry_factor'] > 1.5 for v in k4.values())


def test_bench_sweep_modes_agree():
    a = run_bench('--no-cpu', env={'BC_PREFILTER': '0'})
    b = run_bench('--no-cpu', env={'BC_PREFILTER': '8'})
    assert a['coreset'] == b['coreset']            # same size, same error, same number of failed steps
    assert a['config']['sweep'] == 'fp64' and b['config']['sweep'].startswith('int8')


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
