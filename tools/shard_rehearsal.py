#!/usr/bin/env python3
"""1-rank REHEARSAL of the multi-GPU greedy step on one GPU -- NOT a scaling measurement.

bench.py runs under torch.distributed.run with ONE rank and BC_FORCE_EXCHANGE=1, so that the step takes the multi-rank route
(sweep -> k_rescore -> ncclAllGather of one (S+4)-double record over the library's own RCCL communicator -> replicated
finish) on the row count one of G GPUs would hold at N = 10M: what it shows is the FIXED cost of a shard step (everything but
the sweep's stream), which bounds strong scaling before any real link latency is paid.  A/B (round 5, second half): the default
pre-filter form at that shard size (two-level from 2M rows, csrc/bc_prefilter_i4.h) against the one-level int8 sweep
(BC_PREFILTER=8).

  python tools/shard_rehearsal.py [out.json]      (on the GPU box; ~1 minute)
"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(rows, one_level, steps=300, warmup=120):
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BC_FORCE_EXCHANGE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    if one_level:
        env['BC_PREFILTER'] = '8'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--rows', str(rows), '--steps', str(steps),
           '--warmup', str(warmup), '--no-cpu', '--no-extra', '--no-host', '--detail', '/dev/null']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    if out.returncode != 0:
        raise SystemExit('bench failed: ' + out.stderr[-2000:])
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    st = d['step_stages']
    return {'rows': rows, 'sweep': d['config']['sweep'], 'prefilter': d.get('prefilter'), 'us_per_step': round(1e3 * d['ms_per_step'], 2),
            'sweep_launch_us (HIP events, timed pass)': round(1e3 * d['roofline']['avg_launch_ms'], 2),
            'stage_us (separate instrumented pass: an event pair per stage costs stream time)':
                {k: (None if st[k] is None else round(1e3 * st[k], 2)) for k in ('sweep', 'rescore', 'gather', 'finish')},
            'instrumented_us_per_step': round(1e3 * st['instr_ms_per_step'], 2), 'rccl_ranks': st['rccl_ranks'], 'transport': st['transport'],
            'coreset': d['coreset']}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'profiles', 'r05_shard_rehearsal.json')
    only = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (2, 4, 8)
    sys.path.insert(0, ROOT)
    res = {'what': 'REHEARSAL on ONE GPU, one rank, RCCL all-gather with world = 1: the per-step fixed cost of a row shard; not a scaling curve',
           'workload': 'bench.py synthetic Zellner linreg, D = 128, S = 100, GIGA; rows = the shard one of G GPUs holds at N = 10M',
           'runs': []}
    for G, rows in ((2, 5_000_064), (4, 2_500_096), (8, 1_250_048)):
        if G not in only:
            continue
        for one_level in (0, 1, 0):
            r = run(rows, one_level)
            r['shard_of_G'] = G
            res['runs'].append(r)
            sys.stderr.write(json.dumps(r) + '\n')
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)
    print(out_path)


if __name__ == '__main__':
    main()
