cd $GRAFT_REPO_ROOT
run() { python3 bench.py --steps 20 --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['chain']['kernels_ms_per_step']; c=d['config']
print('$1: %.0f Mpx/s  %.2f ms/step  host_wait %.1f blocked %.1f ' % (d['value'], d['ms_per_step'], c['host_wait_ms_per_step'], c['host_blocked_ms_per_step']) + ' '.join('%s=%.2f' % (a, b) for a, b in k.items()))"; }
for rep in 1 2 3; do
run default
STP_BENCH_ORDER=interleave run interleave
STP_BENCH_ORDER=interleave STP_BENCH_FLIGHT=3 run interleave_flight3
STP_BENCH_ORDER=file run file_order
done
