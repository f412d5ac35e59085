# Matrix-pipe counters of the MFMA microbenchmark (tools/ubench_mfma.hip), one PMC pass: do vector and matrix instructions ever
# execute together on gfx950's f32 MFMA?   bash tools/pmc_mfma.sh   (output: gpurun_out/r05_pmc_mfma.txt)
R=$(pwd); mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --output-format csv --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_mfma -o pmc -- $R/tools/bin/ubench_mfma > $R/gpurun_out/pmc_mfma.log 2>&1
python3 - <<PY
import csv,glob
rows={}
for f in glob.glob('$R/gpurun_out/pmc_mfma/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.setdefault((int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0][:28], r['Workgroup_Size']), {})[r['Counter_Name']]=float(r['Counter_Value'])
print('# dispatches of tools/ubench_mfma in launch order (k_mix<VOP>: workgroup of 256 (nM + nV) threads; every configuration is launched twice)')
for (d,k,wg),c in sorted(rows.items()):
    print('%3d %-28s wg %-5s' % (d,k,wg), ' '.join('%s=%.0f' % (n.replace('SQ_',''), v) for n,v in sorted(c.items())))
PY
rm -rf $R/gpurun_out/pmc_mfma
