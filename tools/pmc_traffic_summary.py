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


def per_launch(rs, counter, name_part, min_grid, max_grid=1 << 60):
    vals = {}
    for r in rs:
        if r['Counter_Name'] != counter or name_part not in r['Kernel_Name']:
            continue
        if int(r['Grid_Size']) < min_grid or int(r['Grid_Size']) > max_grid:
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
           'command': 'rocprofv3 --pmc FETCH_SIZE (and, separately, WRITE_SIZE) --kernel-trace -- python3 bench.py --no-cpu --no-extra --no-host --steps 20 --warmup 5',
           'units': 'FETCH_SIZE / WRITE_SIZE are reported in KB (x1024 bytes); per-launch means over the launches at N = 10M'}
    f, n, kn = per_launch(fetch, 'FETCH_SIZE', 'k_sweep_i8', 100000)
    w, _, _ = per_launch(write, 'WRITE_SIZE', 'k_sweep_i8', 100000)
    alg = N * (S + 4.)
    if f is not None and w is not None:
      out['k_sweep_i8'] = {'kernel': kn, 'launches': n, 'FETCH_SIZE_bytes': f, 'WRITE_SIZE_bytes': w,
                           'traffic_bytes_per_launch': 2 * f + w, 'algorithmic_bytes_per_launch': alg,
                           'ratio': (2 * f + w) / alg,
                           'correction': '2 x FETCH_SIZE + WRITE_SIZE (gfx950 counts the 128-B requests of 16-B/lane streaming reads as 64 B, MI355X_MICROARCH.md HBM section)'}
    # two-level form (round 5): the 4-bit sweep, with level 2's gathers inside
    f4, n4, kn4 = per_launch(fetch, 'FETCH_SIZE', 'k_sweep_i4', 100000)
    w4, _, _ = per_launch(write, 'WRITE_SIZE', 'k_sweep_i4', 100000)
    if f4 is not None and w4 is not None:
        alg4 = N * (4. * 13 + 2.)
        out['k_sweep_i4'] = {'kernel': kn4, 'launches': n4, 'FETCH_SIZE_bytes': f4, 'WRITE_SIZE_bytes': w4,
                             'traffic_bytes_per_launch': 2 * f4 + w4, 'algorithmic_bytes_per_launch': alg4,
                             'ratio': (2 * f4 + w4) / alg4,
                             'correction': '2 x FETCH_SIZE + WRITE_SIZE, as for k_sweep_i8; includes the int8 records of the rows passed on (128 B each)'}
    # K1 at the headline shape: the Theta-resident kernel (one 512-thread block per CU: grid 131072) since round 3, the
    # staged one (one block per tile) before; both read Z with 16 B / 8 B per lane and write 16 B per lane
    for key, part, min_grid, max_grid in (('k_project_r', 'k_project_r<0, 6, 4, true>', 100000, 200000),
                                          ('k_project', 'k_project<0, 6, 32, 2, false, 4', 70000 * 256, 1 << 40)):
        f, n, kn = per_launch(fetch, 'FETCH_SIZE', part, min_grid)
        w, _, _ = per_launch(write, 'WRITE_SIZE', part, min_grid)
        if f is None:
            continue
        ar, aw = 8. * N * (D + 1), 8. * N * S
        out[key] = {'kernel': kn, 'launches': n, 'FETCH_SIZE_bytes': f, 'WRITE_SIZE_bytes': w,
                    'algorithmic_read_bytes': ar, 'algorithmic_write_bytes': aw, 'write_ratio': w / aw,
                    'fetch_x1_ratio': f / ar, 'fetch_x2_ratio': 2 * f / ar,
                    'traffic_bytes_per_launch (2 x FETCH + WRITE)': 2 * f + w, 'ratio_to_algorithmic': (2 * f + w) / (ar + aw),
                    'note': 'stores are 16 B/lane (WRITE_SIZE calibrated: exact).  k_project_r reads Z with 16-B-per-lane buffer loads '
                            '(the calibrated case: FETCH_SIZE x 2); the staged kernel read it 8 B per lane (uncalibrated: both readings given)'}
    if len(sys.argv) > 4:
        # store-free K1 (the beta-Cores gradient loop): a separate pair of passes over tools/k1_bench.py --store-free
        f2, w2 = rows(sys.argv[3]), rows(sys.argv[4])
        f, n, kn = per_launch(f2, 'FETCH_SIZE', 'k_project_r<1, 6, 4, false>', 100000)
        w, _, _ = per_launch(w2, 'WRITE_SIZE', 'k_project_r<1, 6, 4, false>', 100000)
        if f is not None:
            ar = 8. * N * (D + 1)
            out['k_project_r store-free (beta-likelihood)'] = {
                'kernel': kn, 'launches': n, 'FETCH_SIZE_bytes': f, 'WRITE_SIZE_bytes': w, 'algorithmic_bytes (8*N*Dz)': ar,
                'traffic_bytes_per_launch (2 x FETCH + WRITE)': 2 * f + w, 'ratio_to_algorithmic': (2 * f + w) / ar,
                'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python3 tools/k1_bench.py --rows 10000000 --models linreg_beta --store-free --reps 3'}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main()
