# per-kernel durations of the chr16-size chain (tools/probe_chain.py) from a rocprofv3 kernel trace; run on the GPU box
R=$(pwd); cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --output-format csv --kernel-trace --stats -d $R/gpurun_out/kt_probe -o kt -- python3 $R/tools/probe_chain.py > $R/gpurun_out/kt_probe.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob('$R/gpurun_out/kt_probe/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Name'][:60]
        if any(k in n for k in ('canny','gray','lines')):
            print(n, r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'total_ms', round(float(r['TotalDurationNs'])/1e6,2))
PY
rm -rf $R/gpurun_out/kt_probe
