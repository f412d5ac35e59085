# A/B of builds of the library on the SAME box with the genome step of bench.py (10 steps, no CPU baseline / e2e):
#   bash tools/ab_bench.sh stripenn_amd/libA.so stripenn_amd/libB.so ...
R=$(pwd)
for rep in 1 2; do
  for l in "$@"; do
    STP_LIB=$R/$l python3 bench.py --steps 10 --no-cpu-baseline --no-e2e --allow-stp-lib 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['chain']['kernels_ms_per_step']
print('$l: %.0f Mpx/s  %.2f ms/step  host_wait %.1f  ' % (d['value'], d['ms_per_step'], d['config']['host_wait_ms_per_step']) + ' '.join('%s=%.2f' % (a, b) for a, b in k.items()))"
  done
done
