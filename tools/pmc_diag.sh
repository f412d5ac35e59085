# Where k_canny_f32 waits: SQ / TCP / TCC counters of the chr16-size chain, one rocprofv3 pass per counter set, for each library
# given (default: the shipped one).  Run on the GPU box from the repo root:
#   bash tools/pmc_diag.sh [kernel-name-substring] [lib.so ...]      (output: gpurun_out/pmc_diag_<lib>.txt)
K=${1:-k_canny_f32}; shift
LIBS=${@:-libstripenn_hip.so}
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for l in $LIBS; do
  mkdir -p $R/gpurun_out; out=$R/gpurun_out/pmc_diag_${K}_$(basename $l .so).txt; : > $out
  PYTHONPATH=$R STP_LIB=$R/stripenn_amd/$l timeout 200 python3 $R/tools/probe_chain.py | tail -1 >> $out
  n=0
  for set in \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" \
    "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM" \
    "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR" \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
    "TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
    "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum TD_TD_BUSY_sum" \
    "GRBM_GUI_ACTIVE GRBM_COUNT TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
    n=$((n+1)); d=$R/gpurun_out/pmcd_$n
    PYTHONPATH=$R STP_LIB=$R/stripenn_amd/$l timeout 200 rocprofv3 --output-format csv --pmc $set -d $d -o pmc -- python3 $R/tools/probe_chain.py > $d.log 2>&1
    python3 - >> $out <<PY
import csv,glob
acc={}
for f in glob.glob('$d/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if '$K' in r['Kernel_Name'] and '_list' not in r['Kernel_Name']:
            k=r['Counter_Name']; a=acc.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=float(r['Counter_Value'])
if acc:
    for k,(c,v) in sorted(acc.items()): print('%-40s %16.0f  (mean of %d launches)' % (k, v/c, c))
else:
    print('no data for: $set'); print(open('$d.log').read()[-400:])
PY
    rm -rf $d $d.log
  done
  echo "== $l"; cat $out
done
