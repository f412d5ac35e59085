cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ov; rm -rf $O; mkdir -p $O
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --output-format csv -d $O/t2 -o kt -- python3 tools/ab_two_threads.py only2 > $O/t2.log 2>&1 && python3 tools/overlap_analysis.py $O/t2 > $O/t2.txt 2>&1
GPU_MAX_HW_QUEUES=4 rocprofv3 --kernel-trace --output-format csv -d $O/t2q4 -o kt -- python3 tools/ab_two_threads.py only2 > $O/t2q4.log 2>&1 && python3 tools/overlap_analysis.py $O/t2q4 > $O/t2q4.txt 2>&1
STP_BENCH_CTX=2 GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --output-format csv -d $O/c2 -o kt -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline --no-e2e > $O/c2.log 2>&1 && python3 tools/overlap_analysis.py $O/c2 > $O/c2.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/c1 -o kt -- python3 bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline --no-e2e > $O/c1.log 2>&1 && python3 tools/overlap_analysis.py $O/c1 > $O/c1.txt 2>&1
rm -rf $O/t2 $O/t2q4 $O/c2 $O/c1
tail -n 30 $O/*.txt; tail -n 3 $O/t2.log $O/t2q4.log
