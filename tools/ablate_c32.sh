# Time and VALU / wait counters of k_canny_f32 per phase (vertical pass; + horizontal pass; + magnitudes and candidate
# collection; + certified class test; whole kernel) and of k_canny_pipe beside it: timing-only ablation builds, one
# plain run (kernel time from the library's own event timers) and one PMC pass each, chr16-size chain.
# Run on the GPU box from the repo root: bash tools/ablate_c32.sh
ls stripenn_amd/libstp_ablate_c32_1.so > /dev/null 2>&1 || make -s -C stripenn_amd/csrc ablate32 || exit 1
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for l in libstp_ablate_c32_1 libstp_ablate_c32_2 libstp_ablate_c32_3 libstp_ablate_c32_4 libstripenn_hip exact; do
  lib=$R/stripenn_amd/$l.so; mode=f32
  if [ $l = exact ]; then lib=$R/stripenn_amd/libstripenn_hip.so; mode=exact; fi
  PYTHONPATH=$R STP_CANNY=$mode STP_LIB=$lib timeout 200 python3 $R/tools/probe_chain.py > $R/gpurun_out/abl32_$l.time 2>&1
  for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' _)
    PYTHONPATH=$R STP_CANNY=$mode STP_LIB=$lib timeout 200 rocprofv3 --output-format csv --pmc $set -d $R/gpurun_out/abl32_${l}_$tag -o pmc -- python3 $R/tools/probe_chain.py > $R/gpurun_out/abl32_$l.log 2>&1
  done
  python3 - <<PY
import csv,glob
acc={}
for f in glob.glob('$R/gpurun_out/abl32_${l}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_canny_f32' in r['Kernel_Name'] or 'k_canny_pipe<' in r['Kernel_Name']:      # (not the list kernel's empty launches)
            k=r['Counter_Name']; a=acc.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
print('$l', {k: round(v/n) for k,(n,v) in sorted(acc.items())})
PY
  tr ' ' '\n' < $R/gpurun_out/abl32_$l.time | grep canny
  rm -rf $R/gpurun_out/abl32_${l}_*
done
