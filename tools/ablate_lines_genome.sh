#!/bin/bash
# k_lines truncated after phase N (timing-only library, make ablate) on the genome workload: lines ms per step by phase
#   1 load | 2 + hysteresis | 3 + vertical lines, 3-column OR | 4 + block scan | 7 + paint (first direction) | 8 + refine |
#   9 + column statistics | 6 everything but grouping and totals | 5 everything but the totals | 0 the whole kernel
for n in 1 2 3 4 7 8 9 6 5 0; do
  STP_LIB=$PWD/stripenn_amd/libstp_ablate_stops.so STP_LINES_STOP=$n timeout -k 10 200 python3 bench.py --no-extras --no-score --allow-stp-lib --steps 3 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['chain']['kernels_ms_per_step']; print('stop $n: lines %.2f ms / step (step %.1f)' % (k['lines'], d['ms_per_step']))" || exit 1
done
