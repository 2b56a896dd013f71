# Where does the factor 1.8-2.3 over the reference's own fp32 noise on WELL-CONDITIONED weights come from (cfg 1 R18 raw /
# damped, cfg 2 / 3 damped: 5e-5 ... 9e-5 absolute, far inside the flat 1e-3)?  Same test, kernel families switched.
# Output: gpurun_out/parity_attrib_damped.log
mkdir -p gpurun_out
OUT=gpurun_out/parity_attrib_damped.log
: > $OUT
K='inference and (cfg1 or (damped and (cfg2 or cfg3)))'
run() {  # label, env...
  L=$1; shift
  echo "=== $L" >> $OUT
  env "$@" python -m pytest tests/test_fullsize_golden_gpu.py -q -s -k "$K" 2>&1 | grep -E "vs the reference|passed|failed|Error" | sed -e 's/; probs.*//' -e 's/480x853 vs the reference: |logit| max [0-9.]*; //' >> $OUT
}
run "shipped (F(3x3)/F(4x4) auto, folded direct chains, inference folding)" VSPW_X=1
run "Winograd off (direct kernels everywhere)" VSPW_WINOGRAD=0
run "F(2x2) everywhere" VSPW_WINO_TILE=2
run "F(3x3) everywhere" VSPW_WINO_TILE=3
run "direct 3x3 chains NOT folded" VSPW_DIRECT_FOLD=0
cat $OUT
# two-level accumulation of the K >= 2*chunk pointwise GEMMs (diagnostic build: python tools/diag/build_variant.py chunk -DVSPW_WITH_ACCUM_CHUNK)
if [ -f cvpr2021_vspw_implement_amd/lib/libvspw_hip_chunk.so ]; then
  for c in 256 64; do
    L="pointwise GEMM chains of $c (diagnostic build)"
    echo "=== $L" >> $OUT
    VSPW_HIP_LIB=$PWD/cvpr2021_vspw_implement_amd/lib/libvspw_hip_chunk.so VSPW_ACCUM_CHUNK=$c python -m pytest tests/test_fullsize_golden_gpu.py -q -s -k "$K" 2>&1 | grep -E "vs the reference|passed|failed|Error" | sed -e 's/; probs.*//' -e 's/480x853 vs the reference: |logit| max [0-9.]*; //' >> $OUT
  done
  tail -12 $OUT
fi
