"""GPU idle time inside bench.py's timed steps from a rocprofv3 kernel trace: union of the kernel intervals of the last
steps vs the span they cover, and the largest gaps with the kernels around them.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o kt -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-e2e
    python3 tools/gap_analysis.py DIR"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40]))
rows.sort()
# the timed steps: the last 3/4 of the chain launches (k_gray_c3 marks a unit)
gray = [r for r in rows if 'k_gray' in r[2]]
n = len(gray)
t_lo = gray[n // 4][0]
sel = [r for r in rows if r[0] >= t_lo]
t_hi = max(r[1] for r in sel)
busy, cur_s, cur_e, gaps = 0, sel[0][0], sel[0][1], []
prev = sel[0]
for r in sel[1:]:
    if r[0] > cur_e:
        busy += cur_e - cur_s
        gaps.append((r[0] - cur_e, prev[2], r[2]))
        cur_s, cur_e = r[0], r[1]
    else:
        cur_e = max(cur_e, r[1])
    if r[1] >= cur_e:
        prev = r
busy += cur_e - cur_s
span = t_hi - t_lo
print('span %.2f ms, GPU busy (union of kernels) %.2f ms = %.1f %%, idle %.2f ms in %d gaps' % (span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6, len(gaps)))
from collections import Counter
c = Counter()
for g, a, b in gaps:
    c[(a.split('(')[0], b.split('(')[0])] += g
for (a, b), g in c.most_common(12):
    print('  %8.3f ms  after %-36s before %s' % (g / 1e6, a, b))
