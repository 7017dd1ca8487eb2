"""In-situ layer table: joins the library's per-launch log (bench.py --launch-log: label, algorithmic FLOPs / bytes of every conv-family,
BatchNorm and slab-reduce launch of ONE iteration, in launch order) with a rocprofv3 kernel trace of graph-replayed iterations of the
same configuration (kernel durations in situ, no event overhead).  Launch j of a family in the log is launch j of that family in
every traced iteration: the iteration is a fixed sequence of launches.

    python profiles/insitu_table.py KERNEL_TRACE.csv LAUNCH_LOG.json ITERS > profiles/rNN_insitu_layer_table.txt

ITERS = number of timed iterations at the END of the trace to average over (bench.py --steps; run it with --no-eval --no-roofline
--no-cpu-baseline so that the trace ends with the timed replays).  Floors: max(FLOP / 1.75 PFLOP/s, bytes / 6.3 TB/s, 5 us)."""
import csv
import json
import re
import sys
import collections

trace, logf, iters = sys.argv[1], sys.argv[2], int(sys.argv[3])
MFMA, HBM = 1.75e15, 6.3e12

# kernel-name pattern of each logged launch, by label prefix
PAT = [('fwd8', 'gather_fp8_kernel'), ('dgrad8', 'gather_fp8_kernel'), ('fwd', ('gather_gemm_kernel', 'pgemm_kernel')),
       ('dgrad', ('gather_gemm_kernel', 'pgemm_kernel')), ('hm1x1', 'gather_gemm_kernel'), ('wgrad_kw_group', 'wgrad_kw_group_kernel'),
       ('wgrad_kw2', 'wgrad_kw2_kernel'), ('wgrad_kw', 'wgrad_kw_kernel'), ('wgrad_group256', 'wgrad_group256_kernel'), ('wgrad_group', 'wgrad_group_kernel'),
       ('wgrad8', 'wgrad_kw'), ('wgrad', 'wgrad_gemm_kernel'),
       ('bn_stats', 'bn_stats_kernel'), ('bn_finalize', 'bn_finalize'), ('bn_apply', 'bn_apply_kernel'), ('bn_relu_maxpool', 'bn_relu_maxpool_kernel'),
       ('bn_bwd_res', 'bn_bwd_resident_kernel'), ('bn_bwd_reduce', 'bn_bwd_reduce_kernel'), ('bn_bwd_finalize', 'bn_bwd_finalize_kernel'),
       ('bn_bwd_apply', 'bn_bwd_apply_kernel'), ('slab_reduce_group', 'slab_reduce_group_kernel'), ('slab_reduce', 'slab_reduce_kernel')]
FAMILY_KERNELS = {0: ('gather_gemm_kernel', 'gather_fp8_kernel', 'pgemm_kernel', 'wgrad_gemm_kernel', 'wgrad_kw_kernel', 'wgrad_kw2_kernel',
                      'wgrad_group_kernel', 'wgrad_group256_kernel', 'wgrad_kw_group_kernel', 'wgrad_kw8_kernel', 'wgrad_kw28_kernel'),
                  1: ('bn_stats_kernel', 'bn_finalize_kernel', 'bn_finalize_wide_kernel', 'bn_apply_kernel', 'bn_relu_maxpool_kernel',
                      'bn_bwd_resident_kernel', 'bn_bwd_reduce_kernel', 'bn_bwd_finalize_kernel', 'bn_bwd_apply_kernel'),
                  2: ('slab_reduce_kernel', 'slab_reduce_group_kernel')}


def base(name):
    m = re.match(r'_Z\d+([A-Za-z0-9_]+?)I', name)
    if m:
        return m.group(1)
    return re.sub(r'<.*|\(.*', '', re.sub(r'^void ', '', name)).strip()


log = json.load(open(logf))['launches']
rows = []
for r in csv.DictReader(open(trace)):
    rows.append((int(r['Start_Timestamp']), base(r['Kernel_Name']), r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3,
                 int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)))
rows.sort()
out = []
for fam in (0, 1, 2):
    ents = [e for e in log if e['family'] == fam]
    if not ents:
        continue
    ks = [r for r in rows if r[1] in FAMILY_KERNELS[fam]]
    if fam == 1:      # the bias column sum borrows bn_bwd_reduce_kernel<T, XHAT = false, 0> and is not a BatchNorm launch
        ks = [r for r in ks if not (r[1] == 'bn_bwd_reduce_kernel' and 'Lb0ELi0E' in r[2])]
    n = len(ents)
    if len(ks) < n * iters:
        print('# family %d: trace holds %d launches, fewer than %d x %d -- skipped' % (fam, len(ks), iters, n)); continue
    ks = ks[-n * iters:]
    bad = 0
    for j, e in enumerate(ents):
        want = next((k for p, k in PAT if e['label'].startswith(p)), '?')
        durs, grids = [], set()
        for it in range(iters):
            r = ks[it * n + j]
            if not any(w in r[1] for w in ((want,) if isinstance(want, str) else want)):
                bad += 1
            durs.append(r[3]); grids.add(r[4])
        out.append((fam, e['label'], ks[(iters - 1) * n + j][1], sorted(grids), sum(durs) / len(durs), e['flops'], e['bytes'], e['us']))
    if bad:
        print('# family %d: %d of %d joined launches have an unexpected kernel name -- the log and the trace are not the same sequence' % (fam, bad, n * iters))

agg = collections.OrderedDict()
for fam, lab, kern, grids, us, fl, by, ev in out:
    a = agg.setdefault((fam, lab, kern), [0, 0.0, fl, by, grids, 0.0])
    a[0] += 1; a[1] += us; a[5] += ev
tab = []
for (fam, lab, kern), (n, us, fl, by, grids, ev) in agg.items():
    mean = us / n
    floor = max(fl / MFMA * 1e6, by / HBM * 1e6, 5.0)
    tab.append((fam, (mean - floor) * n * 1e-3, lab, kern, n, mean, floor, fl, by, grids, ev / n))
names = {0: 'conv family (MFMA implicit GEMM)', 1: 'BatchNorm', 2: 'weight-gradient slab reductions'}
for fam in (0, 1, 2):
    t = sorted([r for r in tab if r[0] == fam], key=lambda r: -r[1])
    if not t:
        continue
    tot = sum(r[4] * r[5] for r in t) * 1e-3
    flo = sum(r[4] * r[6] for r in t) * 1e-3
    print('\n== %s: %d launches / iteration, %.3f ms in situ, floor %.3f ms, gap %.3f ms   (mean of %d iterations)' %
          (names[fam], sum(r[4] for r in t), tot, flo, tot - flo, iters))
    print('%-62s %-22s %3s %8s %8s %8s %9s %9s %8s' % ('layer', 'kernel [grid]', 'n', 'us', 'floor', 'gap ms', 'TFLOP/s', 'GB/s', 'event us'))
    for fam_, gap, lab, kern, n, mean, floor, fl, by, grids, ev in t:
        print('%-62s %-22s %3d %8.1f %8.1f %8.3f %9.0f %9.0f %8.1f' % (lab[:62], (kern.replace('_kernel', '') + ' ' + str(grids[0] if len(grids) == 1 else grids))[:22],
                                                                 n, mean, floor, gap, fl / mean / 1e6, by / mean / 1e3, ev))
