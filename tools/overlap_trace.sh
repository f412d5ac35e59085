# kernel traces of the genome step under host-side choices, condensed by tools/overlap_analysis.py and tools/gap_analysis.py:
#   bash tools/overlap_trace.sh
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/ov; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p1 -o kt -- python3 bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-e2e > $O/p1.log 2>&1 && python3 tools/overlap_analysis.py $O/p1 > $O/p1.txt 2>&1 && python3 tools/gap_analysis.py $O/p1 >> $O/p1.txt 2>&1
rm -rf $O/p1
cat $O/p1.txt
