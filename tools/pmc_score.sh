# SQ counters of the score kernels alone (tools/probe_scorekernels.py), one PMC pass: bash tools/pmc_score.sh
R=$(pwd); mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
PYTHONPATH=$R timeout 300 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU -d $R/gpurun_out/pmc_score -o pmc -- python3 $R/tools/probe_scorekernels.py 4 > $R/gpurun_out/pmc_score.log 2>&1
python3 - <<PY
import csv,glob
acc={}
for f in glob.glob('$R/gpurun_out/pmc_score/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        if 'score' in k or 'pvalue' in k or 'stripiness' in k:
            a=acc.setdefault((k,r['Counter_Name']),[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
for (k,c),(n,v) in sorted(acc.items()): print('%-62s %-18s n=%d mean %.0f' % (k,c,n,v/n))
PY
rm -rf $R/gpurun_out/pmc_score
