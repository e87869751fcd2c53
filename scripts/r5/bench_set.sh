#!/bin/bash
# the bench lines kept under profiles/r05_*: the driver's command, a 64-step run, 'H' coder, the transform alone at
# 256 MiB (C3 generator, real text, the .py corpus) and 1 GiB (C5), and the 2-rank rehearsal over gloo on the one GPU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/r05_bench
mkdir -p $O
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/n1_driver_command.json 2> $O/n1_driver_command.err || exit 1
timeout -k 10 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/n1_steps100.json 2> $O/n1_steps100.err || exit 1
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --coder H --no-cpu-baseline > $O/n1_coderH.json 2> $O/n1_coderH.err || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --bwt-only --no-cpu-baseline > $O/n1_bwt_only_256.json 2> $O/n1_bwt_only_256.err || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --bwt-only --workload realtext --no-cpu-baseline > $O/n1_bwt_only_realtext.json 2> $O/n1_bwt_only_realtext.err || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --bwt-only --workload pycorpus --no-cpu-baseline > $O/n1_bwt_only_pycorpus.json 2> $O/n1_bwt_only_pycorpus.err || exit 1
timeout -k 10 400 python bench.py --steps 8 --warmup 2 --bwt-only --size-mib 1024 --no-cpu-baseline > $O/n1_c5_1GiB_bwt_only.json 2> $O/n1_c5.err || exit 1
timeout -k 10 400 python bench.py --steps 12 --warmup 3 --workload pycorpus --no-cpu-baseline > $O/n1_B_pycorpus.json 2> $O/n1_B_pycorpus.err || exit 1
BWTC_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/2ranks_one_gpu_gloo_rehearsal.json 2> $O/2ranks.err || exit 1
VERIFY=1 REPS=3 timeout -k 10 400 python3 scripts/r5/workloads.py 256 c3 realtext pycorpus dna random period9 zeros reptext > $O/workloads_bwt_256MiB.jsonl 2> $O/workloads.err || exit 1
echo done
