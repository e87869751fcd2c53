#!/usr/bin/env python3
"""Headline benchmark: MB/s compressed (BWT + encode) on a 256 MiB block, one block per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (BWTManager::doTransform + the entropy front-end that is
built, see config.stages) over one synthetic block that is already resident in HBM.  Blocks
are independent (PrecompressorBlock::sliceIntoBlocks), so rank r works on its own block
(C4: generator seed 30 + r) and no data-path collective exists; torch.distributed is used
only for the barrier and the max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line.

The default coder is the reference's default, 'B' (BASELINE config: "Wavelet coder"): the
device half of a step (transform, run scanner, stream kernels) takes 80 ms, the host half
(adaptive models + range coders, serial by format) 1.6 core-seconds, so the K steps overlap --
every block is completely encoded inside the timed region (drain() collects the last ones).
`single_block_ms` is one block alone; `other_coder` is the all-GPU 'H' route on the same block.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def usable_cpus():
    """CPUs this job may keep busy: affinity mask, cut down to the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(size_mib_sample, seed, coder):
    """Times the CPU path on a bounded sample of the same workload (rank 0, N=1 only).
    BWT = the reference's own divbwtf when oracle/_ref was built ("reference"), else the
    oracle's restatement ("port"); the entropy stage is the oracle's port either way."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from bwtc_amd import synth
    d = synth.gen_text(size_mib_sample << 20, seed)
    use_ref = oracle_lib.ref() is not None
    t0 = time.perf_counter()
    bwt, lf, fr = (oracle_lib.ref_bwt_block if use_ref else oracle_lib.oracle_bwt_block)(d, 8)
    t_bwt = time.perf_counter() - t0
    t_enc = 0.0
    if coder in ("H", "B"):
        t1 = time.perf_counter()
        if coder == "H":
            oracle_lib.oracle_huffman_encode_block(bwt, lf, fr)
        else:
            oracle_lib.oracle_wavelet_encode_block(bwt, lf, fr)
        t_enc = time.perf_counter() - t1
    mb = d.size / 1e6
    return {
        "value": round(mb / (t_bwt + t_enc), 3), "unit": "MB/s", "cores": 1,
        "kind": "reference" if use_ref else "port",
        "bwt_only_MBps": round(mb / t_bwt, 3),
        "sample": "%d MiB of the same text generator (seed %d), 8 starting points: BWT by %s, "
                  "1 thread%s" % (size_mib_sample, seed,
                                  "the reference's divbwtf (oracle/_ref)" if use_ref
                                  else "oracle/bwtc_oracle.c",
                                  "; '%s' encode by the oracle port" % coder if coder else ""),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size-mib", type=int, default=256)
    ap.add_argument("--coder", choices=["B", "H"], default="B",
                    help="entropy coder of the step: B = wavelet (BASELINE config), H = Huffman")
    ap.add_argument("--depth", type=int, default=16,
                    help="'B': blocks under way at once (device half of block i+1 overlaps the host half of block i)")
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from bwtc_amd import hip, synth
    from bwtc_amd.farm import Farm
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one GPU per rank; the modulo only matters when ranks are rehearsed on a smaller box
    gpu = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(gpu)
    dev = torch.device("cuda", gpu)
    # "nccl" is RCCL on ROCm; BWTC_BENCH_BACKEND=gloo rehearses the N>1 path without it
    farm = Farm(backend=os.environ.get("BWTC_BENCH_BACKEND", "nccl"), device=dev)
    rank, world = farm.rank, farm.world

    size = args.size_mib << 20
    # one block per GPU: rank r owns block r of the job (C3 at N=1, C4 seeds 30.. otherwise)
    assert farm.my_blocks(world) == [rank]
    seed = 3 if world == 1 else 30 + rank
    host = synth.gen_text(size, seed)
    d_in = torch.from_numpy(host).to(dev)
    d_out = torch.empty_like(d_in)
    os.environ.setdefault("BWTC_HIP_WAVELET_DEPTH", str(max(1, args.depth)))
    ctx = hip.Context(gpu, size)
    d_comp = torch.empty(ctx.compress_bound(size), dtype=torch.uint8, device=dev)
    h_comp = np.empty(ctx.compress_bound(size), np.uint8)      # the 'B' coder runs on host threads
    cores = usable_cpus()
    threads = max(1, min(64, cores // max(world, 1)))      # the GPU-feeding thread sleeps in its waits
    torch.cuda.synchronize()
    comp = [0]

    # 'B': the device work of a block takes a tenth of the time its range coder needs on the
    # host, so the steps overlap -- a step does the device half of its block and queues the
    # host half on the context's worker threads (bwtc_hip_wavelet_encode_device_begin); every
    # block is finished (..._end) inside the timed region, at the latest by drain().
    depth = max(1, args.depth)
    ring = [np.empty(ctx.compress_bound(size), np.uint8) for _ in range(depth)] if args.coder == "B" else []
    pending = []

    def collect():
        comp[0] = ctx.wavelet_encode_end(pending.pop(0))

    def drain():
        while pending:
            collect()

    issued = [0]

    def step_for(coder):
        def step():
            lf, freqs = ctx.bwt_block_device(d_in.data_ptr(), d_out.data_ptr(), size, 8)
            if coder == "H":
                comp[0] = ctx.huffman_encode_device(d_out.data_ptr(), size, lf, freqs, d_comp.data_ptr())
            elif not ring:
                ctx.wavelet_reset()
                comp[0] = ctx.wavelet_encode_device(d_out.data_ptr(), size, lf, freqs, h_comp, threads)
            else:
                if len(pending) >= depth:
                    collect()
                ctx.wavelet_reset()
                pending.append(ctx.wavelet_encode_device_begin(d_out.data_ptr(), size, lf, freqs,
                                                               ring[issued[0] % depth], threads))
                issued[0] += 1
        return step

    step = step_for(args.coder)

    # one block alone, start to finish (untimed): the latency a single block sees
    single_ms = None
    if args.coder == "B":
        step(); drain()
        t1 = time.perf_counter()
        step(); drain()
        single_ms = 1e3 * (time.perf_counter() - t1)
    for _ in range(args.warmup):
        step()
    drain()
    ctx.reset_kernel_timers()
    elapsed = farm.timed(step, args.steps, 0, drain)
    comp_bytes = comp[0]

    if rank == 0:
        kt = ctx.kernel_timers()
        st = ctx.stats()
        total_mb = world * args.steps * size / 1e6
        roof = None
        # HBM bytes per launch of the same kernel on the same workload, from the committed PMC
        # passes (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 runs, gfx950 correction applied
        # by scripts/pmc_summary.py).  Only quoted for the workload it was measured on.
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic_text256.json")
        if args.size_mib == 256 and os.path.exists(pmc):
            for name, v in json.load(open(pmc))["kernels"].items():
                if "k_radix_scatter<unsigned long>" in name:
                    traffic = int(v["hbm_bytes_per_launch_avg"])
        if kt["scatter_launches"]:
            achieved = kt["scatter_bytes"] / (kt["scatter_ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "k_radix_scatter<u64> (+<u32> initial passes)",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "traffic_source": "profiles/r01_pmc_traffic_text256.json" if traffic else None,
                    "launches": kt["scatter_launches"],
                    "avg_launch_us": round(1e3 * kt["scatter_ms"] / kt["scatter_launches"], 1),
                    "algorithmic_bytes_per_launch": int(kt["scatter_bytes"] / kt["scatter_launches"])}
        out = {
            "metric": "MB/s compressed (BWT+encode) on 256 MiB block",
            "value": round(total_mb / elapsed, 2), "unit": "MB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64 integer", "data": "synthetic",
            "config": {"workload": "C3: %d MiB enwik8-style synthetic text block per GPU "
                                   "(splitmix64 token generator, seed %s), 8 starting points"
                                   % (args.size_mib, "3" if world == 1 else "30+rank"),
                       "stages": "BWT (suffix sort + BWT + LFpowers + freqs) on the GPU + " +
                                 ("'B' wavelet coder: run scanner, tree bit vectors, traversal order and gap "
                                  "flags on the GPU; adaptive models (parallel over tree nodes) and the range "
                                  "coder (one serial chain per section, by the format) on %d host threads"
                                  % threads if args.coder == "B" else
                                  "'H' run-length/Huffman coder on the GPU"),
                       "coder": args.coder,
                       "blocks_per_gpu": 1, "parallelism": "block farm, no collective",
                       "blocks_under_way": depth if args.coder == "B" else 1},
            "device_ms_bwt": round(st.ms_total, 3), "rounds": st.rounds,
            "R_eff": round(st.active_sum / max(st.n, 1), 3),
            "compressed_bytes": int(comp_bytes),
            "single_block_ms": round(single_ms, 1) if single_ms else None,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample_mib, seed, args.coder)
        # the other coder on the same block, two untimed-region steps, for orientation
        other = "H" if args.coder == "B" else "B"
        ostep = step_for(other)
        ostep(); drain()
        t1 = time.perf_counter()
        ostep(); drain()
        out["other_coder"] = {"coder": other, "MBps": round(size / 1e6 / (time.perf_counter() - t1), 2),
                              "compressed_bytes": int(comp[0])}
        print(json.dumps(out), flush=True)
    farm.close()


if __name__ == "__main__":
    main()
