# emulated shares of 8 under host-side choices (bench.py), steps pipelined, units of <= 204 frames: score thread / hardware queues
cd $GRAFT_REPO_ROOT
for cfg in "0 4" "1 4" "1 8"; do
  set -- $cfg
  for r in 0 1 2 3 4 5 6 7; do
    STP_BENCH_SCORE_THREAD=$1 GPU_MAX_HW_QUEUES=$2 python3 bench.py --steps 60 --warmup 3 --emulate-rank $r/8 --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); c=d[\"config\"]; print(\"score_thread $1 queues $2 share $r/8: %.2f ms/step (chain %.2f, host wait %.2f blocked %.2f)\" % (d[\"ms_per_step\"], d[\"roofline\"][\"chain\"][\"kernels_ms_per_step\"][\"chain_wall\"], c[\"host_wait_ms_per_step\"], c[\"host_blocked_ms_per_step\"]))"
  done
done
