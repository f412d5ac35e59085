"""Condense the rocprofv3 output of tools/profile_round.sh into the per-kernel tables kept in profiles/."""
import csv, glob, os, re, sys
from collections import defaultdict

tag = sys.argv[1]
out = 'gpurun_out'


def short(name):
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\(.*$', '', name)


# kernel-trace statistics: copied as produced (one row per kernel)
for f in glob.glob('%s/%s_kt/**/*kernel_stats.csv' % (out, tag), recursive=True):
    with open(f) as src, open('%s/%s_kernel_stats.csv' % (out, tag), 'w') as dst:
        dst.write(src.read())

# PMC passes: mean counter value per launch and kernel
rows = []
for d in sorted(glob.glob('%s/%s_pmc_*' % (out, tag))):
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = (short(r['Kernel_Name']), r['Counter_Name'])
                acc[k][0] += 1
                acc[k][1] += float(r['Counter_Value'])
    for (k, c), (n, v) in sorted(acc.items()):
        rows.append((k, c, n, v / n))
with open('%s/%s_pmc.csv' % (out, tag), 'w') as fh:
    fh.write('kernel,counter,launches,mean_per_launch\n')
    for r in rows:
        fh.write('%s,%s,%d,%.1f\n' % r)
print('wrote %s/%s_kernel_stats.csv and %s/%s_pmc.csv (%d rows)' % (out, tag, out, tag, len(rows)))
