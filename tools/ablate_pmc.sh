# VALU instruction counts of k_canny_pipe per phase (vertical pass; + horizontal pass; + magnitudes and candidate
# collection; whole kernel): timing-only ablation builds + one PMC pass each.
# Run on the GPU box from the repo root: bash tools/ablate_pmc.sh
make -s -C stripenn_amd/csrc ablate
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for l in libstp_ablate_p1 libstp_ablate_p12 libstp_ablate_p123 libstripenn_hip; do
  PYTHONPATH=$R STP_LIB=$R/stripenn_amd/$l.so timeout 200 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $R/gpurun_out/abl_$l -o pmc -- python3 $R/tools/probe_chain.py > $R/gpurun_out/abl_$l.log 2>&1
  python3 - <<PY
import csv,glob
acc={}
for f in glob.glob('$R/gpurun_out/abl_$l/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'canny' in r['Kernel_Name']:
            k=r['Counter_Name']; a=acc.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
print('$l', {k:(n, v/n) for k,(n,v) in acc.items()})
PY
  grep canny $R/gpurun_out/abl_$l.log | tail -1
done
