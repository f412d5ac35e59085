R=$(pwd); out=$R/gpurun_out/r06_soak_long.log
sha=$(python3 -c "from stripenn_amd import hip; print(hip.source_hash())")
echo "# Long soak of the final round-6 sources $sha (grey buffers poisoned before every launch in the first two blocks)" > $out
run() { echo "# $*" >> $out; timeout -k 10 $1 "${@:2}" 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|np.nanmedian\|^  g = " | tail -3 >> $out; }
STP_TEST_POISON_GRAY=1 run 500 python3 tools/soak_overlap.py 401000 900
STP_TEST_POISON_GRAY=1 run 400 python3 tools/soak_fuzz.py 411000 1100
run 200 python3 tools/soak_c32_gpu.py 530 80
run 260 python3 tools/soak_score.py 9900 500
cat $out
