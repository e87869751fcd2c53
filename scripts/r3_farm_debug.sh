#!/bin/bash
# Phase timings of the farm's workers (BWTC_HIP_DEBUG): 8 blocks of 256 MiB over two contexts on one device.
set -e
out=gpurun_out/r3_farm_timing
mkdir -p $out
python - <<'P'
import sys
sys.path.insert(0, ".")
from bwtc_amd import synth
with open("/tmp/farm_in.bin", "wb") as f:
    blocks = [synth.gen_text(268435456, 100 + i).tobytes() for i in range(4)]
    for r in range(2):
        for b in blocks:
            f.write(b)
P
exe=bwtc_amd/host/compress
BWTC_HIP_DEBUG=1 timeout -k 10 300 $exe -v 1 -m 1451 -e B --devices 0,0 /tmp/farm_in.bin /tmp/farm_two.bwtc 2> $out/two_debug.log
grep -E "^farm:|Compressed" $out/two_debug.log
BWTC_HIP_DEBUG=1 BWTC_HIP_MODELS=host timeout -k 10 300 $exe -v 1 -m 1451 -e B --devices 0,0 /tmp/farm_in.bin /tmp/farm_two.bwtc 2> $out/two_host_debug.log
grep -E "^farm:|Compressed" $out/two_host_debug.log
