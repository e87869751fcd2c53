#!/bin/bash
# Does a second, independent stream of kernels fill the gaps of the first?  The transform alone, one
# process against two processes at once on the one GPU (no host coding involved).
mkdir -p gpurun_out/r3_two
timeout -k 10 200 python bench.py --bwt-only --steps 150 --no-cpu-baseline > gpurun_out/r3_two/one.json 2> gpurun_out/r3_two/one.err || exit 1
timeout -k 10 300 python bench.py --bwt-only --steps 300 --no-cpu-baseline > gpurun_out/r3_two/a.json 2> gpurun_out/r3_two/a.err &
pa=$!
timeout -k 10 300 python bench.py --bwt-only --steps 300 --no-cpu-baseline > gpurun_out/r3_two/b.json 2> gpurun_out/r3_two/b.err &
pb=$!
wait $pa; ra=$?
wait $pb; rb=$?
[ $ra -eq 0 ] && [ $rb -eq 0 ] || exit 1
python - <<'P'
import json
for n in ("one", "a", "b"):
    d = json.load(open("gpurun_out/r3_two/%s.json" % n))
    print(n, d["ms_per_step"], d["value"])
P
