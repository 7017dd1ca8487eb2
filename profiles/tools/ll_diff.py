import json, sys, collections
def load(p):
    d = json.load(open(p)); ov = d['event_dispatch_overhead_us']
    agg = collections.OrderedDict()
    for e in d['launches']:
        if e['family'] != 0: continue
        a = agg.setdefault(e['label'], [0, 0.0]); a[0] += 1; a[1] += e['us'] - ov
    return agg
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = []
for k in a:
    if k in b and a[k][0] == b[k][0]:
        ua, ub = a[k][1] / a[k][0], b[k][1] / b[k][0]
        rows.append(((ub - ua) * a[k][0], k, a[k][0], ua, ub))
rows.sort()
for d, k, n, ua, ub in rows[:14] + rows[-8:]:
    print('%+8.1f us/iter  n=%2d  %7.1f -> %7.1f  %s' % (d, n, ua, ub, k))
print('total', sum(r[0] for r in rows))
