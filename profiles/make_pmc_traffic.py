"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/r01_pmc_traffic.json.

usage: python profiles/make_pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <config key>

The conv family = every launch of gather_gemm_kernel / wgrad_gemm_kernel / wgrad_kw_kernel (the kernels bench.py's
roofline object prices).  Bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and, on
gfx950, FETCH_SIZE reports half of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section).
"""
import collections
import csv
import gzip
import json
import os
import sys

FAMILY = ('gather_gemm_kernel', 'wgrad_gemm_kernel', 'wgrad_kw_kernel')


def fold(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    op = gzip.open if path.endswith('.gz') else open
    with op(path, 'rt') as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter or not any(k in r['Kernel_Name'] for k in FAMILY):
                continue
            a = acc[r['Kernel_Name']]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    return acc


def main():
    fetch, write, key = sys.argv[1:4]
    fa, wa = fold(fetch, 'FETCH_SIZE'), fold(write, 'WRITE_SIZE')
    per, launches, total = {}, 0, 0.0
    for k in sorted(fa):
        n = fa[k][0]
        if wa[k][0] != n:
            raise SystemExit('launch counts differ between the passes for %s: %d vs %d' % (k, n, wa[k][0]))
        per[k] = {'launches': n, 'fetch_kb_raw_avg': fa[k][1] / n, 'write_kb_avg': wa[k][1] / n}
        launches += n
        total += (2 * fa[k][1] + wa[k][1]) * 1024
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'r01_pmc_traffic.json')
    rec = json.load(open(out_path)) if os.path.exists(out_path) else {'configs': {}}
    rec['how'] = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace only) -- python3 bench.py '
                  '--steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-roofline; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 '
                  '(gfx950: FETCH_SIZE reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM); '
                  'Infinity-Cache hits are included in the counters; folded by profiles/make_pmc_traffic.py')
    rec['configs'][key] = {'conv_family_launches': launches, 'conv_family_bytes_per_launch': round(total / launches),
                           'per_kernel': per}
    json.dump(rec, open(out_path, 'w'), indent=1)
    print(key, 'launches', launches, 'bytes/launch', round(total / launches))


if __name__ == '__main__':
    main()
