#!/usr/bin/env python3
"""Per-launch HBM traffic of the two dominant kernels from rocprofv3 counter-collection CSVs.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir>/fetch -o r02 --output-format csv -- python3 bench.py --no-cpu --no-extra --steps 20 --warmup 5
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d <dir>/write -o r02 --output-format csv -- python3 bench.py --no-cpu --no-extra --steps 20 --warmup 5
    python3 tools/pmc_traffic_summary.py <dir>/fetch <dir>/write > profiles/rNN_pmc_traffic.json

Separate passes per counter, values in KB (x1024), means over the launches at the headline size (grid filter), the
gfx950 correction of MI355X_MICROARCH.md's HBM section (FETCH_SIZE counts the 128-B requests of wide streaming reads
as 64 B) applied where the guide calibrates it."""
import csv
import glob
import json
import os
import sys

N, D, S = 10_000_000, 128, 100


def rows(d):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not f:
        raise SystemExit('no counter_collection.csv under ' + d)
    with open(f[0]) as fh:
        return list(csv.DictReader(fh))


def per_launch(rs, counter, name_part, min_grid):
    vals = {}
    for r in rs:
        if r['Counter_Name'] != counter or name_part not in r['Kernel_Name']:
            continue
        if int(r['Grid_Size']) < min_grid:
            continue
        vals.setdefault(r['Dispatch_Id'], 0.)
        vals[r['Dispatch_Id']] += float(r['Counter_Value'])
    if not vals:
        return None, 0, None
    v = list(vals.values())
    kn = next(r['Kernel_Name'] for r in rs if name_part in r['Kernel_Name'] and int(r['Grid_Size']) >= min_grid)
    return 1024. * sum(v) / len(v), len(v), kn


def main():
    fetch, write = rows(sys.argv[1]), rows(sys.argv[2])
    out = {'config': {'N': N, 'D': D, 'S': S, 'n_gpus': 1},
           'command': 'rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace -- python3 bench.py --no-cpu --no-extra --steps 20 --warmup 5',
           'units': 'FETCH_SIZE / WRITE_SIZE are reported in KB (x1024 bytes); per-launch means over the launches at N = 10M'}
    f, n, kn = per_launch(fetch, 'FETCH_SIZE', 'k_sweep_i8', 100000)
    w, _, _ = per_launch(write, 'WRITE_SIZE', 'k_sweep_i8', 100000)
    alg = N * (S + 4.)
    out['k_sweep_i8'] = {'kernel': kn, 'launches': n, 'FETCH_SIZE_bytes': f, 'WRITE_SIZE_bytes': w,
                         'traffic_bytes_per_launch': 2 * f + w, 'algorithmic_bytes_per_launch': alg,
                         'ratio': (2 * f + w) / alg,
                         'correction': '2 x FETCH_SIZE + WRITE_SIZE (gfx950 counts the 128-B requests of 16-B/lane streaming reads as 64 B, MI355X_MICROARCH.md HBM section)'}
    f, n, kn = per_launch(fetch, 'FETCH_SIZE', 'k_project', 70000 * 256)
    w, _, _ = per_launch(write, 'WRITE_SIZE', 'k_project', 70000 * 256)
    ar, aw = 8. * N * (D + 1), 8. * N * S
    out['k_project'] = {'kernel': kn, 'launches': n, 'FETCH_SIZE_bytes': f, 'WRITE_SIZE_bytes': w,
                        'algorithmic_read_bytes': ar, 'algorithmic_write_bytes': aw, 'write_ratio': w / aw,
                        'fetch_x1_ratio': f / ar, 'fetch_x2_ratio': 2 * f / ar,
                        'note': 'stores are 16 B/lane (WRITE_SIZE calibrated: exact); the Z loads are 8 B/lane buffer loads and the Theta re-staging per tile is served by L2 -- FETCH_SIZE for that width is uncalibrated in the guide, both readings are given'}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
