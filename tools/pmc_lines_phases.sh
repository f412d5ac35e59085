# SQ counters of k_lines truncated after phase N (timing-only library libstp_ablate_stops.so: cd stripenn_amd/csrc && make ablate), chr16 chain:
#   1 load | 2 + hysteresis | 3 + vertical lines, 3-column OR | 4 + block scan | 7 + paint (first direction) | 8 + refine |
#   9 + column statistics | 6 everything but grouping and totals | 5 everything but the totals | 0 the whole kernel
# One PMC pass per stop (never combined with a trace domain).
R=$(pwd); cd /tmp; export TMPDIR=/tmp
export PYTHONPATH=$R STP_LIB=$R/stripenn_amd/libstp_ablate_stops.so
for n in 1 2 3 4 7 8 9 6 5 0; do
  export STP_LINES_STOP=$n
  rm -rf $R/gpurun_out/plp
  timeout 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $R/gpurun_out/plp -o pmc -- python3 $R/tools/probe_chain.py > $R/gpurun_out/plp.log 2>&1
  python3 - <<PY
import csv,glob
acc={}
for f in glob.glob('$R/gpurun_out/plp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_lines' in r['Kernel_Name']:
            k=r['Counter_Name']; a=acc.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
print('stop $n', ' '.join('%s=%.4g' % (k, v/n) for k,(n,v) in sorted(acc.items())))
PY
  rm -rf $R/gpurun_out/plp
done
