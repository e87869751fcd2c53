#!/bin/bash
# Times `compress` on a stream of 16 blocks of 256 MiB (4 distinct synthetic text blocks, four times):
# one context, two contexts on the one device with the models on the device / on the worker threads.
set -e
out=gpurun_out/r3_farm_timing
mkdir -p $out
python - <<'P'
import sys
sys.path.insert(0, ".")
from bwtc_amd import synth
with open("/tmp/farm_in.bin", "wb") as f:
    blocks = [synth.gen_text(268435456, 100 + i).tobytes() for i in range(4)]
    for r in range(4):
        for b in blocks:
            f.write(b)
P
exe=bwtc_amd/host/compress
for cfg in "one:" "two:--devices 0,0" "two_hostmodels:--devices 0,0"; do
  name=${cfg%%:*}; args=${cfg#*:}
  if [ "$name" = two_hostmodels ]; then export BWTC_HIP_MODELS=host; fi
  timeout -k 10 300 $exe -v 1 -m 1451 -e B $args /tmp/farm_in.bin /tmp/farm_$name.bwtc 2> $out/$name.log
  tail -1 $out/$name.log
done
unset BWTC_HIP_MODELS
cmp /tmp/farm_one.bwtc /tmp/farm_two.bwtc && cmp /tmp/farm_one.bwtc /tmp/farm_two_hostmodels.bwtc && echo "streams equal" | tee $out/equal.log
