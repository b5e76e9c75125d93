#!/usr/bin/env python3
"""Per-instantiation means of the SQ counters of k_project launches (or, with PMC_KERNEL=<name part> in the environment, of any
other kernel: PMC_KERNEL=k_sweep_i4) from rocprofv3 --pmc output directories.

    python tools/pmc_k1_digest.py <dir> [<dir> ...]        # each the -d directory of one counter pass

Prints CSV: kernel (template arguments), launches, then mean per launch of every counter found."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    want = os.environ.get('PMC_KERNEL', 'k_project')
    min_grid = int(os.environ.get('PMC_MIN_GRID', '0'))
    for d in sys.argv[1:]:
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = row.get('Kernel_Name', '')
                    if want not in k or int(row.get('Grid_Size', '0') or 0) < min_grid:
                        continue
                    m = re.search(r'(k_project(?:_r)?<[^>]*>)', k) if want == 'k_project' else re.search(r'(%s[^(]*)' % re.escape(want), k)
                    key = m.group(1) if m else k
                    acc[key][row['Counter_Name']].append((row.get('Dispatch_Id'), float(row['Counter_Value'])))
    names = sorted({c for k in acc for c in acc[k]})
    print(','.join(['kernel: k_project<model,NT,KC,JT,RAW,TL,STORE> | k_project_r<model,NT,TL,STORE>', 'launches'] + names))
    for k in sorted(acc):
        n = 0
        vals = []
        for c in names:
            per = defaultdict(float)
            for disp, v in acc[k].get(c, []):
                per[disp] += v
            n = max(n, len(per))
            vals.append('%.6g' % (sum(per.values()) / max(len(per), 1)) if per else '')
        print(','.join(['"%s"' % k, str(n)] + vals))


if __name__ == '__main__':
    main()
