# memory-side counters of the Canny kernel on the chr16 chain (one pass per counter set); run on the GPU box
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $set | cut -c1-10 | tr ' ' _)
  PYTHONPATH=$R timeout 200 rocprofv3 --output-format csv --pmc $set -d $R/gpurun_out/pm_$tag -o pmc -- python3 $R/tools/probe_chain.py > $R/gpurun_out/pm_$tag.log 2>&1
  python3 - <<PY
import csv,glob
acc={}
for f in glob.glob('$R/gpurun_out/pm_$tag/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_canny_f32' in r['Kernel_Name']:
            k=r['Counter_Name']; a=acc.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
print({k: round(v/n) for k,(n,v) in sorted(acc.items())} or open('$R/gpurun_out/pm_$tag.log').read()[-300:])
PY
  rm -rf $R/gpurun_out/pm_$tag
done
