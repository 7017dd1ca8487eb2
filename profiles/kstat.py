"""Per-kernel / per-grid summary of a rocprofv3 kernel trace.  usage: kstat.py trace.csv iters [name-filter]"""
import csv, sys, re, statistics as st, collections
path, iters = sys.argv[1], int(sys.argv[2]); flt = sys.argv[3] if len(sys.argv) > 3 else ''
def short(n):
    m = re.match(r'_Z\d+([a-z0-9_]+?)I', n)
    if m: 
        t = re.search(r'Li(\d+)ELi(\d+)ELb(\d)ELi(\d)ELi(\d)ELb(\d)', n)
        return m.group(1) + ('<%s,%s,sc%s,hm%s>' % (t.group(1), t.group(2), t.group(3), t.group(6)) if t else '')
    return re.sub(r'\(.*', '', re.sub(r'^void ', '', n))[:60]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    n = short(r['Kernel_Name'])
    if flt and flt not in n: continue
    g = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']) // int(r['Workgroup_Size_Y']), int(r['Grid_Size_Z']))
    rows[(n, g)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = collections.defaultdict(float)
out = []
for (n, g), v in rows.items():
    out.append((sum(v) / iters / 1e3, n, g, len(v) / iters, st.median(v), min(v)))
    tot[n] += sum(v) / iters / 1e3
out.sort(reverse=True)
print('%-40s %-18s %7s %9s %9s %9s' % ('kernel', 'grid', 'n/iter', 'med us', 'min us', 'ms/iter'))
for ms, n, g, c, med, mn in out[:int(60)]:
    print('%-40s %-18s %7.1f %9.1f %9.1f %9.3f' % (n, g, c, med, mn, ms))
print('--- totals'); 
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:25]: print('%-50s %8.3f' % (n, v))
print('ALL', sum(tot.values()))
