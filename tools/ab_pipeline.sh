R=$(pwd)
run() { python3 bench.py --steps 10 --no-cpu-baseline --no-e2e --no-extras 2>/dev/null | python3 -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['chain']['kernels_ms_per_step']
print('$1: %.0f Mpx/s  %.2f ms/step  host_wait %.1f  ' % (d['value'], d['ms_per_step'], d['config']['host_wait_ms_per_step']) + ' '.join('%s=%.2f' % (a, b) for a, b in k.items()))"; }
for rep in 1 2; do
run default
STP_BENCH_FLIGHT=3 run flight3
STP_BENCH_SCORE_THREAD=1 run score_thread
STP_BENCH_SCORE_THREAD=1 STP_BENCH_FLIGHT=3 run score_thread_flight3
STP_BENCH_ORDER=file run file_order
STP_BENCH_PIPELINE_STEPS=1 run pipelined_steps
done
