#!/usr/bin/env python3
"""Digests of rocprofv3 CSV output for profiles/:
    profile_digest.py by-grid <kernel_trace.csv>           kernel durations split by (kernel, grid size): the default bench
                                                           run launches the same kernels at several problem sizes
    profile_digest.py counters <counter_collection.csv>    per (kernel, grid size, counter): dispatches and mean value
Only this library's kernels (k_*) are kept."""
import collections
import csv
import sys


def by_grid(path):
    acc = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            kn = r['Kernel_Name']
            if 'k_' not in kn or kn.startswith('void at::') or 'rocblas' in kn:
                continue
            acc[(kn, int(r.get('Grid_Size') or r['Grid_Size_X']), int(r.get('Workgroup_Size') or r['Workgroup_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    # a grid-stride kernel launched with the same (resident) grid at two problem sizes: split where durations fall
    # into two clusters more than 3x apart
    split = {}
    for key, v in acc.items():
        sv = sorted(v)
        cut = next((i for i in range(1, len(sv)) if sv[i] > 3 * sv[i - 1] and i >= 3 and len(sv) - i >= 3), None)
        if cut is None:
            split[key + ('',)] = v
        else:
            split[key + ('shorter launches',)] = [x for x in v if x < sv[cut]]
            split[key + ('longer launches',)] = [x for x in v if x >= sv[cut]]
    w = csv.writer(sys.stdout)
    w.writerow(['Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'Calls', 'AverageNs', 'MinNs', 'MaxNs', 'Note'])
    for (kn, g, wg, note), v in sorted(split.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([kn, g, wg, len(v), '%.1f' % (sum(v) / len(v)), min(v), max(v), note])


def counters(path):
    acc = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            kn = r['Kernel_Name']
            if 'k_' not in kn or kn.startswith('void at::') or 'rocblas' in kn:
                continue
            acc[(kn, int(r['Grid_Size']), r['VGPR_Count'], r['LDS_Block_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
    w = csv.writer(sys.stdout)
    w.writerow(['Kernel_Name', 'Grid_Size', 'VGPR_Count', 'LDS_Block_Size', 'Counter_Name', 'dispatches', 'mean_Counter_Value'])
    for k, v in sorted(acc.items()):
        if len(v) and k[1] >= 100000:
            w.writerow(list(k) + [len(v), '%.6g' % (sum(v) / len(v))])


if __name__ == '__main__':
    {'by-grid': by_grid, 'counters': counters}[sys.argv[1]](sys.argv[2])
