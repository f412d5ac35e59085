R=$(pwd); out=$R/gpurun_out/r06_soak_long.log
sha=$(python3 -c "from stripenn_amd import hip; print(hip.source_hash())")
echo "# Long soak of the final round-6 sources $sha (grey buffers poisoned before every launch in the first two blocks)" > $out
run() { echo "# $*" >> $out; timeout -k 10 $1 "${@:2}" 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|np.nanmedian\|^  g = " | tail -3 >> $out; }
STP_TEST_POISON_GRAY=1 run 400 python3 tools/soak_overlap.py 301000 700
STP_TEST_POISON_GRAY=1 run 300 python3 tools/soak_fuzz.py 311000 900
run 200 python3 tools/soak_c32_gpu.py 430 60
run 200 python3 tools/soak_score.py 9500 400
cat $out
