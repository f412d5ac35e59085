cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for P in 0 1; do
  echo "PIPELINE_STEPS=$P whole: $(STP_BENCH_PIPELINE_STEPS=$P python3 bench.py --steps 20 --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['value'])")"
  for r in 0 3 5 7; do
    echo "PIPELINE_STEPS=$P share $r/8: $(STP_BENCH_PIPELINE_STEPS=$P python3 bench.py --steps 60 --warmup 3 --emulate-rank $r/8 --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'])")"
  done
done; done
