"""Condense the rocprofv3 output of tools/profile_round.sh into the per-kernel tables kept in profiles/
(<tag>_kernel_stats.csv, <tag>_pmc.csv) and into the <workload> entry of pmc_current.json that bench.py reads:
per kernel the mean HBM counters per launch (FETCH_SIZE / WRITE_SIZE in KB, as rocprofv3 reports them), the VALU
wave-instruction count per launch and the image-px one launch processes (from the bench line of the same tag)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

tag = sys.argv[1]
workload = sys.argv[2] if len(sys.argv) > 2 else 'genome'
out = 'gpurun_out'
SHORT = {'k_canny_f32': 'canny', 'k_canny_pipe': 'canny_f64', 'k_canny_pipe_list': 'canny_redo', 'k_canny': 'canny', 'k_gray': 'gray', 'k_gray_c3': 'gray', 'k_lines': 'lines', 'k_pvalue': 'pvalue',
         'k_stripiness': 'stripiness', 'k_frame_prep': 'frame_prep', 'k_score_wave': 'score'}


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
if not rows and os.path.exists('%s/%s_pmc.csv' % (out, tag)):
    # the raw passes are gone (profile_round.sh deletes them): rebuild pmc_current.json from the condensed table, e.g.
    # after a later run with other bench arguments has overwritten the workload's entry
    with open('%s/%s_pmc.csv' % (out, tag)) as fh:
        for r in csv.DictReader(fh):
            rows.append((r['kernel'], r['counter'], int(r['launches']), float(r['mean_per_launch'])))
else:
    with open('%s/%s_pmc.csv' % (out, tag), 'w') as fh:
        fh.write('kernel,counter,launches,mean_per_launch\n')
        for r in rows:
            fh.write('"%s",%s,%d,%.1f\n' % r)
print('wrote %s/%s_kernel_stats.csv and %s/%s_pmc.csv (%d rows)' % (out, tag, out, tag, len(rows)))

# bench.py's view: per launch means keyed by the library's own kernel names
try:
    line = json.loads(open('%s/%s_bench.json' % (out, tag)).read().strip().splitlines()[-1])
except (OSError, ValueError, IndexError):
    line = None
sys.path.insert(0, '.')
from stripenn_amd import hip as _hip
entry = {'tag': tag, 'src_sha': _hip.source_hash(), 'kernels': {}}      # the sources the counters were collected on
for k, c, n, v in rows:
    base = re.sub(r'<.*$', '', k)
    if base not in SHORT:
        continue
    e = entry['kernels'].setdefault(SHORT[base], {})
    key = {'FETCH_SIZE': 'fetch_kb', 'WRITE_SIZE': 'write_kb', 'SQ_INSTS_VALU': 'valu_insts',
           'SQ_ACTIVE_INST_VALU': 'valu_active', 'SQ_BUSY_CYCLES': 'busy_cycles', 'SQ_WAVE_CYCLES': 'wave_cycles',
           'SQ_WAIT_ANY': 'wait_any'}.get(c)
    if key:
        e[key] = v
        e.setdefault('launches_profiled', n)
if line is not None:
    r = line['roofline']
    img_px_step = line['config']['contact_px_per_step'] * 6.0
    for name, e in entry['kernels'].items():
        if name in ('canny', 'gray', 'lines'):
            e['image_px'] = img_px_step / r['launches_per_step']        # the three chain kernels launch together
path = '%s/pmc_current.json' % out
cur = {}
for p in ('profiles/pmc_current.json', path):
    try:
        cur.update(json.load(open(p)))
    except (OSError, ValueError):
        pass
cur[workload] = entry
json.dump(cur, open(path, 'w'), indent=1, sort_keys=True)
print('wrote', path, '(copy it to profiles/pmc_current.json)')
