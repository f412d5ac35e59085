for reuse in 1 0; do
for n in 1 2 3 0; do
  STP_REUSE=$reuse STP_LIB=$PWD/stripenn_amd/libstp_ablate_stops.so STP_LINES_STOP=$n timeout -k 10 200 python3 bench.py --no-extras --no-score --no-cpu-baseline --no-e2e --allow-stp-lib --steps 5 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['chain']['kernels_ms_per_step']; print('reuse $reuse stop $n: lines %.2f canny %.2f ms / step (step %.1f)' % (k['lines'], k['canny'], d['ms_per_step']))" || exit 1
done
done
