#!/bin/sh
# Development: does a second context on the same GPU (its kernels in the gaps of the first one's)
# raise the throughput of `compress`?  A file of 24 text blocks of 256 MiB, compressed with one,
# two and three contexts on device 0; the streams must be the same bytes.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
F=/tmp/probe_text.bin
python3 - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
from bwtc_amd import hip
size = 256 << 20
buf = np.empty(size, np.uint8)
with open("$F", "wb") as f:
    for j in range(24):
        hip.synth_into("t", 3 + (j % 6), buf)
        f.write(buf.tobytes())
PY
ls -la $F
for d in 0 0,0 0,0,0; do
  t0=$(date +%s.%N)
  bwtc_amd/host/compress -m 1452 -s 8 --devices $d $F /tmp/probe_$d.bwtc 2>&1 | tail -2
  t1=$(date +%s.%N)
  python3 -c "print('devices $d: %.2f s wall' % ($t1 - $t0))"
done
sha256sum /tmp/probe_0.bwtc /tmp/probe_0,0.bwtc /tmp/probe_0,0,0.bwtc | cut -c1-16
rm -f $F /tmp/probe_*.bwtc
