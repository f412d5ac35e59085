# emulated shares of 8 under unit-plan choices (bench.py units_for): plan / piece / order
for cfg in "equal 128 file" "equal 128 size" "taper 128 file" "equal 100 file" "equal 170 file" "taper 170 file"; do
  set -- $cfg
  for r in 0 2 5 6; do
    STP_BENCH_PLAN=$1 STP_BENCH_PIECE=$2 STP_BENCH_ORDER=$3 python3 bench.py --steps 20 --emulate-rank $r/8 --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(\"plan $1 piece $2 order $3 share $r/8: %.2f ms/step (chain %.2f)\" % (d[\"ms_per_step\"], d[\"roofline\"][\"chain\"][\"kernels_ms_per_step\"][\"chain_wall\"]))"
  done
done
