#!/bin/bash
# Measurement artefacts of one round, run on the GPU box from the repo root:
#   bash tools/profile_round.sh r02a [genome|chr16]
# Writes gpurun_out/<tag>_*: the bench JSON line, the rocprofv3 kernel-trace statistics of the same
# workload and -- in separate passes, never combined with a trace domain -- the PMC counters
# (FETCH_SIZE, WRITE_SIZE, SQ instruction / busy counters).  tools/summarize_profile.py turns the raw
# CSVs into the small per-kernel tables kept under profiles/ and into profiles/pmc_current.json, which
# bench.py reads for `roofline.traffic` and the FP64-VALU issue fraction.
set -u
TAG=${1:-r02x}
WL=${2:-genome}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p $OUT
STEPS=5; PSTEPS=1
if [ "$WL" = "chr16" ]; then STEPS=20; PSTEPS=2; fi
B="$R/bench.py --workload $WL ${STP_PROFILE_BENCH_ARGS:-}"
timeout 600 python3 $B --steps $STEPS --warmup 1 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
B="$B --no-extras"
timeout 400 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/${TAG}_kt -o kt -- python3 $B --steps $STEPS --warmup 1 --no-cpu-baseline --no-e2e > $OUT/${TAG}_kt.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --output-format csv --pmc $C -d $OUT/${TAG}_pmc_$C -o pmc -- python3 $B --steps $PSTEPS --warmup 0 --no-cpu-baseline --no-e2e > $OUT/${TAG}_pmc_$C.log 2>&1 || exit 1
done
timeout 400 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/${TAG}_pmc_SQ -o pmc -- python3 $B --steps $PSTEPS --warmup 0 --no-cpu-baseline --no-e2e > $OUT/${TAG}_pmc_SQ.log 2>&1 || exit 1
timeout 400 rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/${TAG}_pmc_SQ2 -o pmc -- python3 $B --steps $PSTEPS --warmup 0 --no-cpu-baseline --no-e2e > $OUT/${TAG}_pmc_SQ2.log 2>&1 || exit 1
cd $R
python3 tools/summarize_profile.py $TAG $WL
# the raw traces are large (gpurun copies back at most 64 MiB): keep the summaries only
rm -rf $OUT/${TAG}_kt $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE $OUT/${TAG}_pmc_SQ $OUT/${TAG}_pmc_SQ2
