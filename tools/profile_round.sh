#!/bin/bash
# Measurement artefacts of one round, run on the GPU box from the repo root:
#   bash tools/profile_round.sh r01d
# Writes gpurun_out/<tag>_*: the bench JSON line, the rocprofv3 kernel-trace statistics of the same
# command and -- in separate passes, never combined with a trace domain -- the PMC counters
# (FETCH_SIZE, WRITE_SIZE, SQ instruction / busy counters).  tools/summarize_profile.py turns the raw
# CSVs into the small per-kernel tables kept under profiles/.
set -u
TAG=${1:-r01x}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
timeout 240 python3 bench.py --steps 5 --warmup 1 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_kt -o kt -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --output-format csv --pmc $C -d $OUT/${TAG}_pmc_$C -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_$C.log 2>&1
done
timeout 240 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/${TAG}_pmc_SQ -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_SQ.log 2>&1
timeout 240 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY -d $OUT/${TAG}_pmc_SQ2 -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_SQ2.log 2>&1
cd $R
python3 tools/summarize_profile.py $TAG
