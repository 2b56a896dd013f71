# Attribution of the raw-weight parity excess (|hip - ref64| / |ref32 - ref64|) on cfg 2/3/4: Winograd on / off /
# direct kernels with two-level (k-blocked) accumulation.  Needs lib/libvspw_hip_chunk8.so
# (python tools/diag/build_variant.py chunk8 -DVSPW_NT_CHUNK=8).  Output: gpurun_out/parity_attrib.log
mkdir -p gpurun_out
OUT=gpurun_out/parity_attrib.log
: > $OUT
K='raw and (cfg2 or cfg3 or cfg4)'
run() {  # label, env...
  L=$1; shift
  echo "=== $L" >> $OUT
  env "$@" python -m pytest tests/test_fullsize_golden_gpu.py -q -s -k "$K" 2>&1 | grep -E "vs the reference|grad |passed|failed|Error" >> $OUT
}
run "winograd on (shipped)" VSPW_WINOGRAD=1
run "winograd off (direct kernels)" VSPW_WINOGRAD=0
run "direct kernels + k-blocked accumulation (chunk 256)" VSPW_WINOGRAD=0 VSPW_HIP_LIB=$PWD/cvpr2021_vspw_implement_amd/lib/libvspw_hip_chunk8.so
run "winograd on + k-blocked accumulation in the direct / pointwise kernels" VSPW_WINOGRAD=1 VSPW_HIP_LIB=$PWD/cvpr2021_vspw_implement_amd/lib/libvspw_hip_chunk8.so
cat $OUT
