# Randomised GPU soaks on the round-5 kernels (run on the GPU box from the repo root; output: gpurun_out/r05_soak.log)
R=$(pwd); out=$R/gpurun_out/r05_soak.log; mkdir -p $R/gpurun_out
sha=$(python3 -c "from stripenn_amd import hip; print(hip.source_hash())")
echo "# Randomised GPU soaks on the round-5 kernels (LDS-free wave transposes in k_lines, interior flat rule + word-column-major class planes in the Canny kernels, k_score_wave with symmetric reads, stp_band_pack_csr), sources $sha" > $out
export STP_FAULT_LOG=$R/gpurun_out/r05_fault.log
run() { echo "# $*" >> $out; timeout -k 10 $1 "${@:2}" 2>&1 | grep -v "amdgpu.ids\|RuntimeWarning\|np.nanmedian\|^  g = " | tail -4 >> $out; }
run 280 python3 tools/soak_fuzz.py 21000 600
run 200 python3 tools/soak_misc.py 5000 150
run 120 python3 tools/soak_c32_gpu.py 130 40
run 250 python3 tools/soak_score.py 2000 300
run 330 python3 tools/soak_pipeline.py 23000 220
STP_SCORE=block run 100 python3 tools/soak_score.py 3000 60
cat $out
