#!/bin/bash
# the head build on a fresh box: the whole GPU suite, smoke(), the driver's bench command, and a bounded
# randomised campaign over the sorter's phases (3: repetitive inputs, 6: the long-key route's switches)
# -> gpurun_out/r05_head/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/r05_head
mkdir -p $O
timeout -k 10 780 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1 || { tail -30 $O/gputest.log; exit 1; }
tail -2 $O/gputest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/n1_driver_command.json 2> $O/n1_driver_command.err || { tail -20 $O/n1_driver_command.err; exit 1; }
cat $O/n1_driver_command.json
# the transform alone with the process held to four CPUs (elapsed - device time per block)
timeout -k 10 200 python -c "import os; a=sorted(os.sched_getaffinity(0))[:4]; os.sched_setaffinity(0,a); os.execvp('python',['python','bench.py','--steps','20','--warmup','3','--bwt-only','--no-cpu-baseline'])" > $O/n1_bwt_only_256_4cpus.json 2> $O/n1_bwt_only_256_4cpus.err || exit 1
python -c "import json; d=json.load(open('$O/n1_bwt_only_256_4cpus.json')); print('bwt-only on 4 CPUs: ms_per_step', d['ms_per_step'], 'gpu_ms_per_step', d['gpu_ms_per_step'], 'device_ms_bwt', d['device_ms_bwt'], 'frac', d['roofline']['frac'])"
timeout -k 10 200 python scripts/fuzz_gpu_parity.py ${FUZZ_S:-30} 50505 36 > $O/fuzz.log 2>&1 || { tail -20 $O/fuzz.log; exit 1; }
grep -E "^phase|^mismatches" $O/fuzz.log
echo done
