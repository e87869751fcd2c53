#!/bin/bash
# `compress` end to end on a long stream (32 blocks of 256 MiB = 8.6 GB, four distinct synthetic text blocks
# repeated): the default loop of one context against the block farm with one context (--devices 0).
set -e
out=gpurun_out/r3_cli_timing
mkdir -p $out
python - <<'P'
import sys
sys.path.insert(0, ".")
from bwtc_amd import synth
with open("/tmp/cli_in.bin", "wb") as f:
    blocks = [synth.gen_text(268435456, 100 + i).tobytes() for i in range(4)]
    for r in range(8):
        for b in blocks:
            f.write(b)
P
exe=bwtc_amd/host/compress
for cfg in "loop:" "farm1:--devices 0" "loop_again:"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout -k 10 300 $exe -v 1 -m 1451 -e B $args /tmp/cli_in.bin /tmp/cli_$name.bwtc 2> $out/$name.log
  echo "$name: $(tail -1 $out/$name.log)"
done
cmp /tmp/cli_loop.bwtc /tmp/cli_farm1.bwtc && echo "streams equal"
