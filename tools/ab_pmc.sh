# A/B of library builds on the SAME box with SQ counters of k_canny_pipe (chr16-size step, tools/probe_chain.py):
#   bash tools/ab_pmc.sh stripenn_amd/libstp_ab_old.so stripenn_amd/libstripenn_hip.so
# One PMC pass per library and counter group (never combined with a trace domain).
R=$(pwd); cd /tmp; export TMPDIR=/tmp
export PYTHONPATH=$R
for l in "$@"; do
  export STP_LIB=$R/$l
  tag=$(basename $l .so)
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    rm -rf $R/gpurun_out/abp_$tag
    timeout 200 rocprofv3 --output-format csv --pmc $grp -d $R/gpurun_out/abp_$tag -o pmc -- python3 $R/tools/probe_chain.py > $R/gpurun_out/abp_$tag.log 2>&1
    python3 - <<PY
import csv,glob
acc={}
for f in glob.glob('$R/gpurun_out/abp_$tag/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'canny' in r['Kernel_Name']:
            k=r['Counter_Name']; a=acc.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
print('$tag', ' '.join('%s=%.4g' % (k, v/n) for k,(n,v) in sorted(acc.items())))
PY
    rm -rf $R/gpurun_out/abp_$tag
  done
  grep canny $R/gpurun_out/abp_$tag.log | tail -1
done
