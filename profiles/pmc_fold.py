"""Fold rocprofv3 passes of ONE bench.py command into profiles/<tag>_pmc_traffic.json:

    python profiles/pmc_fold.py <tag> <config key> <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv> [algo.json]

  * kernel trace (``--kernel-trace``): per-kernel launch count and duration;
  * two separate ``--pmc`` passes (FETCH_SIZE, WRITE_SIZE; each with ``--kernel-trace`` only): HBM-side bytes per launch =
    (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- both counters are in KiB and, on gfx950, FETCH_SIZE reports half of a wide
    coalesced streaming read (MI355X_MICROARCH.md, section HBM); Infinity-Cache hits are included in the counters.
Per kernel family: launches, measured bytes, duration, counter-based GB/s (bytes / duration) and, when algo.json
(profiles/membound_algo.py) is given, the algorithmic bytes and the traffic / algorithmic ratio."""
import collections
import csv
import gzip
import json
import os
import re
import sys

FAMILIES = [            # first match wins
    ('conv_mfma', ('gather_gemm_kernel', 'wgrad_gemm_kernel', 'wgrad_kw_kernel', 'wgrad_kw2_kernel', 'wgrad_kw_group_kernel', 'gather_fp8_kernel', 'pgemm_kernel',
                   'wgrad_group_kernel', 'wgrad_group256_kernel', 'wgrad_kw8_kernel', 'wgrad_kw28_kernel')),
    ('slab_reduce', ('slab_reduce_kernel', 'slab_reduce_group_kernel')),
    ('bn_fwd', ('bn_stats_kernel', 'bn_apply_kernel', 'bn_finalize_kernel', 'bn_finalize_wide_kernel', 'bn_relu_maxpool_kernel')),
    ('bn_bwd', ('bn_bwd_resident_kernel', 'bn_bwd_reduce_kernel', 'bn_bwd_apply_kernel', 'bn_bwd_finalize_kernel')),
    ('kl_loss', ('kl_heatmap_kernel', 'kl_heatmap_reg_kernel')),
    ('argmax', ('argmax2d_kernel', 'argmax2d_reg_kernel')),
    ('softargmax', ('softargmax_kernel', 'softargmax_reg_kernel')),
    ('pseudo_label', ('pseudo_label_kernel', 'pseudo_label_reg_kernel')),
    ('pointwise21', ('pw_k2c', 'pw_c2k', 'pw_wgrad', 'hm_rowsum')),
    ('fp8_quantize', ('quantize_fp8_kernel', 'pack_weights_fp8_kernel', 'fp8_update_scale_kernel')),
    ('optimizer', ('sgd_kernel', 'pack_weights')),
    ('pool_layout', ('maxpool', 'nchw_to_nhwc', 'nhwc_to_nchw')),
]


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return 'other'


def _open(path):
    return gzip.open(path, 'rt') if path.endswith('.gz') else open(path)


def fold_counter(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with _open(path) as f:
        for r in csv.DictReader(f):
            if r['Counter_Name'] != counter:
                continue
            a = acc[r['Kernel_Name']]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    return acc


def fold_trace(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with _open(path) as f:
        for r in csv.DictReader(f):
            a = acc[r['Kernel_Name']]
            a[0] += 1
            a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    return acc


def main():
    tag, key, trace, fetch, write = sys.argv[1:6]
    algo = json.load(open(sys.argv[6])) if len(sys.argv) > 6 else {}
    tr, fa, wa = fold_trace(trace), fold_counter(fetch, 'FETCH_SIZE'), fold_counter(write, 'WRITE_SIZE')
    fams = collections.defaultdict(lambda: dict(launches=0, bytes=0.0, seconds=0.0, kernels={}))
    for k in sorted(tr):
        n, sec = tr[k]
        if k not in fa or k not in wa:
            continue
        if fa[k][0] != n or wa[k][0] != n:
            raise SystemExit('launch counts differ between the passes for %s: trace %d fetch %d write %d' % (k, n, fa[k][0], wa[k][0]))
        b = (2 * fa[k][1] + wa[k][1]) * 1024
        f = fams[family(k)]
        f['launches'] += n; f['bytes'] += b; f['seconds'] += sec
        short = re.sub(r'\(.*', '', k)[:110]
        f['kernels'][short] = dict(launches=n, bytes_per_launch=round(b / n), us_per_launch=round(sec / n * 1e6, 2),
                                   GBps=round(b / sec / 1e9, 1))
    out = {}
    for name, f in fams.items():
        rec = dict(launches=f['launches'], bytes_per_launch=round(f['bytes'] / f['launches']), ms_total=round(f['seconds'] * 1e3, 3),
                   counter_GBps=round(f['bytes'] / f['seconds'] / 1e9, 1), frac_of_8TBps=round(f['bytes'] / f['seconds'] / 8e12, 4),
                   kernels=f['kernels'])
        a = algo.get('families', {}).get(name)
        if a and algo.get('iterations_in_trace'):
            ab = a['bytes_per_iteration'] * algo['iterations_in_trace']
            rec.update(algorithmic_bytes_per_launch=round(ab / f['launches']), traffic_over_algorithmic=round(f['bytes'] / ab, 3),
                       algorithmic_GBps=round(ab / f['seconds'] / 1e9, 1))
        out[name] = rec
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '%s_pmc_traffic.json' % tag)
    doc = json.load(open(path)) if os.path.exists(path) else {'configs': {}}
    doc['how'] = ('three runs of the same command under rocprofv3 (--kernel-trace; --pmc FETCH_SIZE --kernel-trace; --pmc WRITE_SIZE '
                  '--kernel-trace): bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md section HBM), '
                  'durations from the un-countered kernel trace; folded by profiles/pmc_fold.py')
    doc['configs'][key] = dict(command=algo.get('command'), git_head=algo.get('git_head'), families=out,
                               conv_family_launches=out.get('conv_mfma', {}).get('launches'),
                               conv_family_bytes_per_launch=out.get('conv_mfma', {}).get('bytes_per_launch'))
    json.dump(doc, open(path, 'w'), indent=1)
    for name, r in sorted(out.items(), key=lambda kv: -kv[1]['ms_total']):
        print('%-14s launches %6d  %9.1f MB/launch  %8.2f ms  %7.1f GB/s (counter)%s' % (
            name, r['launches'], r['bytes_per_launch'] / 1e6, r['ms_total'], r['counter_GBps'],
            '  traffic/algorithmic %.2f' % r['traffic_over_algorithmic'] if 'traffic_over_algorithmic' in r else ''))


if __name__ == '__main__':
    main()
