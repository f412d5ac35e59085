# Randomised GPU soaks on the round-6 kernels (run on the GPU box from the repo root; output: gpurun_out/r06_soak.log)
R=$(pwd); out=$R/gpurun_out/r06_soak.log; mkdir -p $R/gpurun_out
sha=$(python3 -c "from stripenn_amd import hip; print(hip.source_hash())")
echo "# Randomised GPU soaks on the round-6 kernels (image symmetry in k_canny_f32, frame overlap in k_canny_f32 / k_lines), sources $sha" > $out
export STP_FAULT_LOG=$R/gpurun_out/r06_fault.log
run() { echo "# $*" >> $out; timeout -k 10 $1 "${@:2}" 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|np.nanmedian\|^  g = " | tail -4 >> $out; }
run 330 python3 tools/soak_overlap.py ${1:-31000} 260
run 200 python3 tools/soak_fuzz.py ${2:-41000} 400
run 100 python3 tools/soak_c32_gpu.py 230 30
run 250 python3 tools/soak_pipeline.py ${3:-53000} 160
cat $out
