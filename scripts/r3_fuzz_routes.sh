#!/bin/bash
# The randomised stream campaign (phase 2) over the alternative host routes of the 'B' coder.
mkdir -p gpurun_out/r3_fuzz_routes
bad=0
for cfg in "BWTC_HIP_MODELS=host" "BWTC_HIP_W_ENGINES=0" "BWTC_HIP_SIMD=0" "BWTC_HIP_MODELS=host BWTC_HIP_FUSED=1" "BWTC_HIP_TEST_MODELS_FALLBACK=1" "BWTC_HIP_WAVELET=host"; do
  name=$(echo "$cfg" | tr ' =' '__')
  env $cfg timeout -k 10 300 python scripts/fuzz_gpu_parity.py 90 424242 2 > gpurun_out/r3_fuzz_routes/$name.log 2>&1 || bad=1
  echo "$cfg: $(grep -E '^phase 2|^mismatches' gpurun_out/r3_fuzz_routes/$name.log | tr '\n' ' ')"
  grep -q "Memory access fault" gpurun_out/r3_fuzz_routes/$name.log && exit 9
done
exit $bad
