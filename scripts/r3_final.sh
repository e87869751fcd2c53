#!/bin/bash
# Round-3 closing measurements: the GPU suite, the bench lines kept under profiles/, the profile.
set -o pipefail
O=gpurun_out/r3_final
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 500 python bench.py > $O/default.json 2> $O/default.err || { tail -5 $O/default.err; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/steps20_warmup5.json 2> $O/steps20.err || exit 1
timeout -k 10 300 python bench.py --steps 400 --no-cpu-baseline > $O/steps400.json 2> $O/steps400.err || exit 1
timeout -k 10 300 python bench.py --depth 32 --steps 100 --no-cpu-baseline > $O/depth32_steps100.json 2> $O/depth32.err || exit 1
timeout -k 10 300 python bench.py --coder H --no-cpu-baseline > $O/coderH.json 2> $O/coderH.err || exit 1
python - <<'P'
import json
for n in ("default", "steps20_warmup5", "steps400", "depth32_steps100", "coderH"):
    d = json.load(open("gpurun_out/r3_final/%s.json" % n))
    print(n, d["value"], d["ms_per_step"], d.get("gpu_ms_per_step"), d.get("closed_loop", {}).get("MBps"), d.get("host_core_s_per_block"), d.get("cpu_baseline", {}).get("value"))
P
scripts/profile_r03.sh > $O/profile.log 2>&1 || { tail -5 $O/profile.log; exit 1; }
echo profile done
