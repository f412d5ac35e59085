"""Which kernels run side by side: from a rocprofv3 kernel trace, the time the device runs 0 / 1 / 2+ kernels, the milliseconds each
pair of chain kernels spends together, average durations and the hardware queues each kernel was dispatched on.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o kt -- python3 tools/ab_two_threads.py only2
    python3 tools/overlap_analysis.py DIR"""
import csv, glob, sys
from collections import Counter, defaultdict
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:28], r.get('Queue_Id', '?')))
rows.sort()
gray = [r for r in rows if r[2].startswith('k_gray_c3')]
t_lo = gray[len(gray) // 3][0]                       # the later two thirds: timed steps
sel = [r for r in rows if r[0] >= t_lo]
t_hi = max(r[1] for r in sel)
ev = []
for i, r in enumerate(sel):
    ev.append((r[0], 1, i)); ev.append((r[1], 0, i))
ev.sort()
live, last, depth_t, pair_t = set(), t_lo, Counter(), Counter()
for t, kind, i in ev:
    dt = t - last
    if dt > 0:
        depth_t[min(len(live), 3)] += dt
        names = sorted(sel[j][2] for j in live)
        for a in range(len(names)):
            for b in range(a + 1, len(names)):
                pair_t[(names[a], names[b])] += dt
    last = t
    if kind:
        live.add(i)
    else:
        live.discard(i)
span = t_hi - t_lo
print('span %.1f ms: idle %.2f, one kernel %.2f, two %.2f, three+ %.2f' % tuple(x / 1e6 for x in (span, depth_t[0], depth_t[1], depth_t[2], depth_t[3])))
print('together (ms):')
for (a, b), t in pair_t.most_common(14):
    print('  %8.2f  %-28s %s' % (t / 1e6, a, b))
dur, qs = defaultdict(list), defaultdict(Counter)
for r in sel:
    dur[r[2]].append(r[1] - r[0]); qs[r[2]][r[3]] += 1
print('kernel: launches, mean ms, total ms, queues')
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print('  %-28s %5d  %7.3f  %8.2f  %s' % (k, len(v), sum(v) / len(v) / 1e6, sum(v) / 1e6, dict(qs[k])))
