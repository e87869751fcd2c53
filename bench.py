#!/usr/bin/env python3
"""Headline benchmark: MB/s compressed (BWT + encode) on 256 MiB blocks, one GPU per rank.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one synthetic block, measured as SURVEY.md 8(d) defines
it: from the block resident in page-locked HOST memory to its compressed record in host memory.
  * Every step takes a different block (C3 generator, seeds 3, 4, ... cycled over --blocks
    distinct blocks; C4 seeds 30 + rank + 8 j when there are several ranks).
  * The upload of block i+1 (copy stream) overlaps the kernels of block i (compute stream), two
    device input buffers.
  * 'B' (the reference's default coder, BASELINE config 3): the device half of a block (BWT, run
    scanner, stream kernels) is followed by a host half (adaptive models + range coders, serial
    by format) on the context's worker threads, so blocks are pipelined `--depth` deep.  The
    timed region is the STEADY STATE of that pipeline: the pipeline is filled before the clock
    starts (max(W, depth) untimed blocks), every timed step begins one block and collects the
    finished record of the oldest one, and what is still under way after step K is collected
    after the clock stops.  K blocks enter and K records leave inside the region; fill and drain
    times are reported separately (`fill_ms`, `drain_ms`).  The model state runs on from block
    to block (one stream, no reset).
  * 'H': everything on the GPU, the record is copied to host memory inside the step.
  * --bwt-only: the transform alone (output stays in HBM); with --size-mib 1024 this is
    BASELINE config 5.
Blocks are independent, so rank r works on its own blocks and no data-path collective exists;
torch.distributed is used only for the barrier and the max-over-ranks of the elapsed time.
Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
GUIDE_COPY_GBS = 6290.0  # MI355X_MICROARCH.md: what a float4 copy kernel reaches (79 % of the spec peak)


def _rss_gb():
    """[resident memory of this process now, its high-water mark so far], GB."""
    try:
        f = dict(l.split(":", 1) for l in open("/proc/self/status") if l.startswith(("VmRSS", "VmHWM")))
        return [round(int(f[k].split()[0]) * 1024 / 1e9, 1) for k in ("VmRSS", "VmHWM")]
    except Exception:
        return None


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


def _ranges(cpus):
    """[0, 1, 2, 3, 8, 9] -> "0-3,8-9"."""
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else "%d-%d" % (cpus[i], cpus[j]))
        i = j + 1
    return ",".join(out)


def mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def has_avx512():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                f = line.split()
                return all(x in f for x in ("avx512f", "avx512bw", "avx512vl", "avx512dq"))
    except OSError:
        pass
    return False


def cpu_baseline(size_mib, seed, coder, runs):
    """The CPU path on the same block, one thread (the reference cannot use more,
    Compressor.cpp:67-70): BWT = the reference's own divbwtf when oracle/_ref was built
    ("reference"), else the oracle's restatement ("port"); the entropy stage is the oracle's
    literal restatement of the reference coder either way.  Rank 0, N=1 only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from bwtc_amd import hip
    d = hip.synth_into("t", seed, np.empty(size_mib << 20, np.uint8))
    use_ref = oracle_lib.ref() is not None
    bwt_s, enc_s = [], []
    for _ in range(max(1, runs)):
        t0 = time.perf_counter()
        bwt, lf, fr = (oracle_lib.ref_bwt_block if use_ref else oracle_lib.oracle_bwt_block)(d, 8)
        bwt_s.append(time.perf_counter() - t0)
        t1 = time.perf_counter()
        if coder == "H":
            oracle_lib.oracle_huffman_encode_block(bwt, lf, fr)
        elif coder == "B":
            oracle_lib.oracle_wavelet_encode_block(bwt, lf, fr)
        enc_s.append(time.perf_counter() - t1)
    mb = d.size / 1e6
    tot = [a + b for a, b in zip(bwt_s, enc_s)]
    return {
        "value": round(mb / min(tot), 3), "unit": "MB/s", "cores": 1,
        "kind": "reference" if use_ref else "port",
        "median_MBps": round(mb / statistics.median(tot), 3), "runs": len(tot),
        "bwt_only_MBps": round(mb / min(bwt_s), 3),
        "bwt_only_median_MBps": round(mb / statistics.median(bwt_s), 3),
        "sample": "the whole %d MiB block of the same generator (seed %d), 8 starting points, %d run(s), best "
                  "(median beside it): BWT by %s, 1 thread%s"
                  % (size_mib, seed, len(tot),
                     "the reference's divbwtf (oracle/_ref)" if use_ref else "oracle/bwtc_oracle.c",
                     "; '%s' encode by the oracle's restatement of the reference coder" % coder if coder in ("H", "B") else ""),
    }


def _cpu_c4_worker(cpu, size_mib, seed, coder):
    """One of the pinned single-thread runs of cpu_baseline_c4 (a process of its own: bench.py
    --cpu-c4-worker cpu size seed coder): its own block, its own core; prints seconds and the kind."""
    os.sched_setaffinity(0, {cpu})
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from bwtc_amd import hip
    d = hip.synth_into("t", seed, np.empty(size_mib << 20, np.uint8))
    use_ref = oracle_lib.ref() is not None
    t0 = time.perf_counter()
    bwt, lf, fr = (oracle_lib.ref_bwt_block if use_ref else oracle_lib.oracle_bwt_block)(d, 8)
    if coder == "H":
        oracle_lib.oracle_huffman_encode_block(bwt, lf, fr)
    elif coder == "B":
        oracle_lib.oracle_wavelet_encode_block(bwt, lf, fr)
    print("%.4f %s" % (time.perf_counter() - t0, "reference" if use_ref else "port"), flush=True)


def cpu_baseline_c4(size_mib, coder, cpus, n=8):
    """SURVEY.md 8(d), the CPU analogue of one block per GPU: n independent single-thread runs of the
    CPU path, each on its own block (C4 seeds 30...) and pinned to its own core, at the same time."""
    import subprocess
    cpus = sorted(cpus)[:n]
    if len(cpus) < n:
        return {"error": "only %d CPUs usable, %d needed" % (len(cpus), n)}
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-c4-worker", str(cpus[i]), str(size_mib),
                               str(30 + i), coder or "-"], stdout=subprocess.PIPE, text=True) for i in range(n)]
    res = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
            sec, kind = out.split()[-2:]
            res.append((float(sec), kind))
        except (subprocess.TimeoutExpired, ValueError):
            p.kill()
            return {"error": "a CPU run did not finish"}
    wall = time.perf_counter() - t0
    mb = n * (size_mib << 20) / 1e6
    return {"value": round(mb / max(r[0] for r in res), 3), "unit": "MB/s", "cores": n, "kind": res[0][1],
            "per_run_s": [round(r[0], 2) for r in res], "wall_s": round(wall, 2),
            "sample": "%d blocks of %d MiB (C4 seeds 30..%d), one single-thread run each, pinned to CPUs %s, all at once; "
                      "value = all their bytes / the slowest run" % (n, size_mib, 30 + n - 1, _ranges(cpus))}


def _cpu_seconds():
    """user + system CPU seconds of this process so far (all threads)."""
    import resource
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


def _thread_cpu():
    """{tid: (name, CPU seconds)} of this process's threads."""
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(tid)] = (name, (int(rest[11]) + int(rest[12])) / tck)
        except (OSError, ValueError):
            pass
    return out


def _cgroup_throttle():
    """(periods, throttled periods, throttled seconds) of this process's cgroup, or None."""
    try:
        d = {}
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, _, v = line.partition(" ")
            d[k] = int(v)
        return (d.get("nr_periods", 0), d.get("nr_throttled", 0), d.get("throttled_usec", 0) / 1e6)
    except (OSError, ValueError):
        return None


def main():
    if len(sys.argv) == 6 and sys.argv[1] == "--cpu-c4-worker":
        _cpu_c4_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size-mib", type=int, default=256)
    ap.add_argument("--coder", choices=["B", "H"], default="B",
                    help="entropy coder of the step: B = wavelet (BASELINE config), H = Huffman")
    ap.add_argument("--bwt-only", action="store_true", help="the transform alone (config 5 with --size-mib 1024)")
    ap.add_argument("--depth", type=int, default=0,
                    help="'B': blocks under way at once.  0 = what the product picks for a stream as long as the one "
                         "this run codes (compress.cpp: 128 / 96 blocks -- the fused lane engines -- only for streams "
                         "of 256 blocks and more, when the host has AVX-512 and 40 / 24 GB per rank; else 20, 16 below 40 GB)")
    ap.add_argument("--blocks", type=int, default=4, help="distinct input blocks, cycled")
    ap.add_argument("--workload", default="c3",
                    help="extras only (the headline is c3): realtext, pycorpus, realtext_rep, dna, random, zeros, period9, reptext -- "
                         "the inputs of scripts/r5/workloads.py instead of the C3 generator's blocks")
    ap.add_argument("--cpu-runs", type=int, default=3, help="runs of the one-thread CPU baseline (best and median reported)")
    ap.add_argument("--cpu-size-mib", type=int, default=0, help="CPU baseline block (0 = same as --size-mib)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-c4", action="store_true",
                    help="also time 8 pinned single-thread CPU runs at once (the CPU analogue of the 8-GPU row): cpu_baseline_c4")
    ap.add_argument("--corpus-blocks", type=int, default=16,
                    help="BWTC_CORPUS=<file>: how many blocks of --size-mib of it are coded (reported as `corpus`)")
    args = ap.parse_args()
    coder = None if args.bwt_only else args.coder

    import torch
    from bwtc_amd import hip
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
    world_hint = max(1, int(os.environ.get("WORLD_SIZE", "1")))
    # The depth the product would pick for this stream (bwtc_amd/host/compress.cpp): the fused host
    # engines keep a text block under way for seconds, so the deep pipelines (128 / 96 blocks) are
    # only for streams of 256 blocks and more; everything else runs 20 deep (16 with less than 40 GB).
    mem_per_rank = mem_available_gb() / world_hint
    stream_blocks = max(args.warmup, 20) + args.steps
    # (deep pipelines are for the host-model routes; with the models on the GPU 13-14 blocks carry the
    # rate and 20 absorb the spread of the host half)
    deep = stream_blocks >= 256 and has_avx512() and os.environ.get("BWTC_HIP_MODELS") == "host"
    auto_depth = 128 if deep and mem_per_rank >= 40 else 96 if deep and mem_per_rank >= 24 else 24 if mem_per_rank >= 48 else 20 if mem_per_rank >= 40 else 16
    # A rank of a full node may get less than the 40 GB and 16 CPUs the 'B' pipeline likes (8 ranks: 320 GB, 128 CPUs):
    # the run then goes on with what there is -- fewer blocks under way (a block under way holds about 1.5 GB of
    # page-locked staging and records), fewer worker threads -- and says so in the line (`degraded`).
    degraded = []
    if coder == "B" and mem_per_rank < 40:
        auto_depth = max(4, min(auto_depth, int((mem_per_rank - 4.0) / 1.5)))
        degraded.append("%.0f GB of host memory per rank (40 wanted): %d blocks under way" % (mem_per_rank, auto_depth))
    depth = (args.depth if args.depth > 0 else auto_depth) if coder == "B" else 1
    os.environ.setdefault("BWTC_HIP_WAVELET_DEPTH", str(depth))
    cores = usable_cpus()
    threads = max(1, min(64, cores // max(world, 1)))      # the GPU-feeding thread sleeps in its waits
    if coder == "B" and cores // max(world_hint, 1) < 16:
        degraded.append("%d usable CPUs per rank (16 wanted): the host half will bind the step" % (cores // max(world_hint, 1)))
    if os.environ.get("BWTC_BENCH_THREADS"):
        threads = max(1, int(os.environ["BWTC_BENCH_THREADS"]))
    ctx = hip.Context(gpu, size)
    # Each rank's workers, the thread that feeds its GPU and (by first touch) the page-locked
    # buffers they share stay on the NUMA node of the rank's GPU, on CPUs no other rank uses.
    numa_node = ctx.numa_node()
    my_cpus = farm.cpu_slice(numa_node) if os.environ.get("BWTC_BENCH_AFFINITY", "1") != "0" else []
    if os.environ.get("BWTC_BENCH_CPULIST"):             # experiments: an explicit list, e.g. "64-79,192"
        my_cpus = []
        for part in os.environ["BWTC_BENCH_CPULIST"].split(","):
            a, _, b = part.partition("-")
            my_cpus += list(range(int(a), int(b or a) + 1))
    if my_cpus:
        ctx.set_worker_cpus(my_cpus)
        os.sched_setaffinity(0, my_cpus)
        threads = max(1, min(threads, len(my_cpus)))

    # distinct blocks in page-locked host memory: C3 seeds 3, 4, ... (one rank) / C4 seeds 30 + rank + 8 j
    nblk = max(1, args.blocks)
    seeds = [(3 + j) if world == 1 else (30 + rank + 8 * j) for j in range(nblk)]
    if args.workload == "c3":
        pool = [hip.synth_into("t", s, ctx.host_alloc(size)) for s in seeds]
    else:
        sys.path.insert(0, os.path.join(ROOT, "scripts", "r5"))
        import workloads
        data, workload_what = workloads.gen(args.workload, size)
        pool = [ctx.host_alloc(size)]
        pool[0][:] = data
        nblk = 1
    d_in = [ctx.dmalloc(size + 64), ctx.dmalloc(size + 64)]
    d_out = ctx.dmalloc(size + 64)
    bound = ctx.compress_bound(size)
    d_comp = ctx.dmalloc(bound) if coder == "H" else None
    ring = [np.empty(bound, np.uint8) for _ in range(depth + 1)] if coder == "B" else []
    h_rec = ctx.host_alloc(bound) if coder == "H" else None      # page-locked: the record comes down in one DMA

    pending = []                 # tickets of blocks under way ('B')
    comp = [0]
    issued = [0]
    clock = {"gpu_s": 0.0, "collect_s": 0.0}

    def collect():
        t = time.perf_counter()
        comp[0] = ctx.wavelet_encode_end(pending.pop(0))
        clock["collect_s"] += time.perf_counter() - t

    def drain():
        while pending:
            collect()

    def upload(i):
        ctx.to_device_async(d_in[i % 2], pool[i % nblk])

    feeder_cpu = {}                                      # BWTC_BENCH_THREAD_CPU=1: CPU seconds of this thread per call site

    def cpu_of(name, fn, *a):
        if os.environ.get("BWTC_BENCH_THREAD_CPU") != "1":
            return fn(*a)
        c0, w0 = time.thread_time(), time.perf_counter()
        r = fn(*a)
        v = feeder_cpu.setdefault(name, [0.0, 0.0])
        v[0] += time.thread_time() - c0
        v[1] += time.perf_counter() - w0
        return r

    def step():
        i = issued[0]
        if coder == "B" and len(pending) >= depth:
            cpu_of("collect", collect)                   # the oldest block's record (host memory)
        t = time.perf_counter()
        cpu_of("copy_wait", ctx.copy_wait)               # block i has landed (its copy ran under block i-1's kernels)
        cpu_of("upload", upload, i + 1)                  # block i+1 goes up while block i is transformed
        lf, freqs = cpu_of("bwt_block_device", ctx.bwt_block_device, d_in[i % 2], d_out, size, 8)
        if coder == "H":
            comp[0] = ctx.huffman_encode_device(d_out, size, lf, freqs, d_comp)
            rc = ctx.lib.bwtc_hip_memcpy_to_host(ctx.handle, h_rec.ctypes.data, d_comp, comp[0])
            if rc:
                raise hip.BwtcHipError("bwtc_hip_memcpy_to_host failed with code %d" % rc)
        elif coder == "B":
            pending.append(cpu_of("wavelet_begin", ctx.wavelet_encode_device_begin, d_out, size, lf, freqs, ring[i % len(ring)], threads))
        clock["gpu_s"] += time.perf_counter() - t
        issued[0] += 1

    rss0 = _rss_gb()
    # ---- untimed: H2D alone, one block alone, pipeline fill ---------------------------------
    t0 = time.perf_counter()
    upload(0)
    ctx.copy_wait()
    h2d_ms = 1e3 * (time.perf_counter() - t0)
    step(); drain()                                      # first use: allocations, worker pool
    issued[0] = 0
    upload(0)
    t0 = time.perf_counter()
    step(); drain()
    single_ms = 1e3 * (time.perf_counter() - t0)
    ctx.wavelet_reset()
    issued[0] = 0
    upload(0)
    t0 = time.perf_counter()
    fill_blocks = max(args.warmup, depth if coder == "B" else 0)
    for _ in range(fill_blocks):
        step()                                           # nothing is collected here: the pipeline fills
    # ... and runs until it is in its steady state (full steps: one block begun, one record
    # collected): a stream's first seconds -- page-locked buffers being made, lanes filling with the
    # sections of several blocks -- are its start-up, reported in fill_ms and in the closed loop
    settle = 2 * depth if coder == "B" else 0
    for _ in range(settle):
        step()
    fill_blocks += settle
    # The depth the stream has shown to need (the library's own figure: block latency over block interval, a quarter
    # more, plus two) replaces the memory-derived default for the timed region -- fewer blocks under way hold less
    # page-locked memory, and what an 8-GPU node has to provide per rank is this number, not the default.
    depth_default = depth
    depth_measured = ctx.wavelet_depth_needed() if coder == "B" else 0
    if coder == "B" and args.depth == 0 and depth_measured and os.environ.get("BWTC_BENCH_FIXED_DEPTH") != "1":
        depth = max(8, min(depth_default, depth_measured))
        while len(pending) > depth:
            collect()
        ctx.wavelet_set_depth(depth)                     # staging beyond the need goes back to the system (what 8 ranks hold is 8 x this)
    fill_ms = 1e3 * (time.perf_counter() - t0)
    rss = {"before_first_block": rss0, "after_fill": _rss_gb()}
    ctx.reset_kernel_timers()
    clock["gpu_s"] = clock["collect_s"] = 0.0
    m0, c0, b0 = ctx.wavelet_host_clock()
    f0 = ctx.wavelet_host_progress()[1]
    dev_ms = []

    trace_rss = os.environ.get("BWTC_BENCH_RSS_TRACE") == "1"

    def timed_step():
        step()
        dev_ms.append(ctx.stats().ms_total)
        if trace_rss and len(dev_ms) % 25 == 0:
            print("step %d: rss/hwm %s" % (len(dev_ms), _rss_gb()), file=sys.stderr, flush=True)

    # ---- timed: K steps, each begins one block and (B) collects one ------------------------------
    cpu0, thr0 = _cpu_seconds(), _cgroup_throttle()
    tc0 = _thread_cpu() if os.environ.get("BWTC_BENCH_THREAD_CPU") == "1" else None
    elapsed = farm.timed(timed_step, args.steps, 0, None)
    cpu1, thr1 = _cpu_seconds(), _cgroup_throttle()
    if tc0 is not None:
        tc1 = _thread_cpu()
        rows = sorted(((tc1[t][1] - tc0.get(t, ("", 0.0))[1], t, tc1[t][0]) for t in tc1), reverse=True)
        print("feeder thread, CPU s / wall s per call site (whole run): %s" % {k: [round(x, 3) for x in v] for k, v in feeder_cpu.items()}, file=sys.stderr)
        print("CPU seconds per thread inside the timed region (%.3f s wall):" % elapsed, file=sys.stderr)
        for dt, t, name in rows:
            if dt > 0.0:
                print("  tid %d %-16s %.3f" % (t, name, dt), file=sys.stderr)
    rss["after_timed_region"] = _rss_gb()
    m1, c1, b1 = ctx.wavelet_host_clock()
    f1 = ctx.wavelet_host_progress()[1]
    block_latency = ctx.wavelet_latency()                # begun -> record finished, mean over the stream so far
    gpu_s, collect_s = clock["gpu_s"], clock["collect_s"]            # of the timed region only
    t0 = time.perf_counter()
    drain()
    drain_ms = 1e3 * (time.perf_counter() - t0)
    rss["after_drain"] = _rss_gb()
    comp_bytes = comp[0]
    # records the host workers finished inside the timed region: the rate the host half really
    # sustained (the records COLLECTED there may all have been finished during the fill)
    finished = (f1 - f0) if coder == "B" else args.steps
    # what every rank got and did (a skewed NUMA slice or a starved rank shows here, not in the maximum alone)
    per_rank = {
        "host_threads": farm.all_gather_int(threads if coder == "B" else 0),
        "numa_node": farm.all_gather_int(numa_node),
        "cpus_in_slice": farm.all_gather_int(len(my_cpus)),
        "staging_peak_MB": farm.all_gather_int(hip.host_staging_bytes()[1] // 1000000),
        "gpu_us_per_step": farm.all_gather_int(1e6 * gpu_s / args.steps),
        "collect_wait_us_per_step": farm.all_gather_int(1e6 * collect_s / args.steps),
        "records_finished_in_region": farm.all_gather_int(finished),
        "depth_used": farm.all_gather_int(depth),
        "depth_needed": farm.all_gather_int(ctx.wavelet_depth_needed() if coder == "B" else 1),
        "process_cpu_ms_per_step": farm.all_gather_int(1e3 * (cpu1 - cpu0) / args.steps),
    }
    finished = int(-farm.max(-finished))                 # the slowest rank's count

    # a real corpus (enwik8 / enwik9 ...) when one is supplied, cut into blocks of
    # the same size, as ONE closed stream: first block begun -> last record collected.  The
    # synthetic generator has 64 tokens; this is the protection against tuning to it.
    kt_main, st_main = ctx.kernel_timers(), ctx.stats()    # of the timed region's blocks, before another stream runs
    corpus = None
    cpath = os.environ.get("BWTC_CORPUS")
    if cpath and rank == 0 and coder in ("B", "H"):
        try:
            fsize = os.path.getsize(cpath)
            nb = max(1, min(args.corpus_blocks, (fsize + size - 1) // size))
            cblocks = []
            with open(cpath, "rb") as f:
                for _ in range(nb):
                    raw = f.read(size)
                    if not raw:
                        break
                    b = ctx.host_alloc(len(raw))
                    b[:] = np.frombuffer(raw, np.uint8)
                    cblocks.append(b)
            ctx.wavelet_reset()
            cring = [np.empty(ctx.compress_bound(size), np.uint8) for _ in range(depth + 1)]
            cpend, cbytes, t0 = [], 0, time.perf_counter()
            ctx.to_device_async(d_in[0], cblocks[0])
            for i, blk in enumerate(cblocks):
                if coder == "B" and len(cpend) >= depth:
                    cbytes += ctx.wavelet_encode_end(cpend.pop(0))
                ctx.copy_wait()
                if i + 1 < len(cblocks):
                    ctx.to_device_async(d_in[(i + 1) % 2], cblocks[i + 1])
                lf, freqs = ctx.bwt_block_device(d_in[i % 2], d_out, blk.size, 8)
                if coder == "B":
                    cpend.append(ctx.wavelet_encode_device_begin(d_out, blk.size, lf, freqs, cring[i % (depth + 1)], threads))
                else:
                    n = ctx.huffman_encode_device(d_out, blk.size, lf, freqs, d_comp)
                    rc = ctx.lib.bwtc_hip_memcpy_to_host(ctx.handle, h_rec.ctypes.data, d_comp, n)
                    if rc:
                        raise hip.BwtcHipError("bwtc_hip_memcpy_to_host failed with code %d" % rc)
                    cbytes += n
            while cpend:
                cbytes += ctx.wavelet_encode_end(cpend.pop(0))
            wall = time.perf_counter() - t0
            total = sum(b.size for b in cblocks)
            corpus = {"path": os.path.basename(cpath), "bytes": int(total), "blocks": len(cblocks),
                      "compressed_bytes": int(cbytes), "ratio": round(cbytes / max(total, 1), 4),
                      "MBps": round(total / 1e6 / wall, 2), "wall_s": round(wall, 3),
                      "what": "the file's first blocks as one closed stream (first block begun -> last record collected), "
                              "the stream's own model state from block to block"}
        except (OSError, hip.BwtcHipError) as ex:
            corpus = {"path": cpath, "error": str(ex)}

    # the spec peak beside what a streaming copy reaches on this box -- the library's own
    # 16-byte-per-lane copy kernel through the C ABI (bwtc_hip_copy_probe), and the guide's figure for the same kind
    # of kernel (6.29 TB/s).  (Until round 4 this was torch.Tensor.copy_ on uint8, which reaches 4.7 TB/s and flattered
    # every fraction taken of it.)
    copy_gbs = None
    if rank == 0:
        try:
            copy_gbs = ctx.copy_probe(1 << 30, 5)
        except Exception:                                                    # never a reason to lose the bench line
            copy_gbs = None

    if rank == 0:
        kt, st = kt_main, st_main
        total_mb = world * args.steps * size / 1e6
        r_eff = st.active_sum / max(st.n, 1)
        roof = None
        # HBM bytes per launch of the same kernel on the same workload, from the committed PMC
        # passes (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 runs, gfx950 correction applied
        # by scripts/pmc_summary.py).  Only quoted for the workload it was measured on.
        traffic, traffic_src = None, None
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_text256.json")))
        if args.size_mib == 256 and args.workload == "c3" and pmcs:
            # all scatter launches of the transform (the same mix the probe averages over)
            tot_b, tot_n = 0.0, 0
            pmc_kernels = json.load(open(pmcs[-1]))["kernels"]
            has_long = any("k_radix_scatter<unsigned long" in k and "unsigned short, " in k for k in pmc_kernels)
            for name, v in pmc_kernels.items():
                # the launches the probe times: the suffix sorter's full-size passes (round 4: the long-key sort's,
                # whose items carry a 16-bit value and a second key word; not the 'B' coder's keys-only step sort,
                # not the few-thousand-item passes of the text rounds)
                is_long = "k_radix_scatter<unsigned long" in name and any("unsigned short, %d," % m in name for m in (1, 2, 3))
                is_old = "k_radix_scatter<" in name and "unsigned short" not in name and ", true, unsigned int, 0, 16>" not in name
                if (is_long or (is_old and not has_long and v["hbm_bytes_per_launch_avg"] > 1e8 and "r04" not in os.path.basename(pmcs[-1]))):
                    tot_b += v["hbm_bytes_per_launch_avg"] * v["launches"]
                    tot_n += v["launches"]
            if tot_n:
                traffic = int(tot_b / tot_n)
                traffic_src = os.path.relpath(pmcs[-1], ROOT)
        if kt["scatter_launches"]:
            achieved = kt["scatter_bytes"] / (kt["scatter_ms"] * 1e-3) / 1e9
            bwt_ms = statistics.mean(dev_ms)
            # the whole transform by SURVEY.md 8(d)'s counting rule: N (97 + 100 R_eff) bytes
            whole_bytes = st.n * (97.0 + 100.0 * r_eff)
            long_route = bool(getattr(st, "route", 0) & 1)
            own_bytes = int(getattr(st, "alg_bytes", 0))
            roof = {"bound": "hbm",
                    "kernel": ("k_radix_scatter<u64 key, u16 + u32 values> (the %d passes of the long-key suffix sort, 14-byte items)"
                               % int(round(getattr(st, "sort_pass_items", 0) / max(1, st.n)) or 9)
                               if long_route else "k_radix_scatter<u64> (+<u32> passes of the suffix sorter)"),
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "device_copy_GBps": round(copy_gbs, 1) if copy_gbs else None,
                    "device_copy_kernel": "bwtc_hip_copy_probe: 1 GiB, one 16-byte item per thread, read + written bytes",
                    "guide_copy_GBps": GUIDE_COPY_GBS,
                    "frac_of_device_copy": round(achieved / copy_gbs, 4) if copy_gbs else None,
                    "frac_of_guide_copy": round(achieved / GUIDE_COPY_GBS, 4),
                    "traffic": traffic,
                    "traffic_source": traffic_src,
                    "launches": kt["scatter_launches"],
                    "avg_launch_us": round(1e3 * kt["scatter_ms"] / kt["scatter_launches"], 1),
                    "algorithmic_bytes_per_launch": int(kt["scatter_bytes"] / kt["scatter_launches"]),
                    "whole_transform": {"algorithmic_bytes": int(whole_bytes), "device_ms": round(bwt_ms, 3),
                                        "achieved": round(whole_bytes / (bwt_ms * 1e-3) / 1e9, 1),
                                        "frac": round(whole_bytes / (bwt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        "rule": "SURVEY.md 8(d): N (97 + 100 R_eff) bytes / device time of the transform",
                                        # the same counting rule (every array a kernel reads or writes, once) applied
                                        # to the kernels that actually ran: the survey's formula prices doubling rounds
                                        # over 4-byte keys, and a design that needs no rounds has R_eff near 0
                                        "own_algorithmic_bytes": own_bytes,
                                        "own_frac": round(own_bytes / (bwt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if own_bytes else None,
                                        "route": int(getattr(st, "route", 0))}}
        step_ms = 1e3 * elapsed / args.steps
        # what binds the step: the host half binds it when records come back late (collect wait, fewer records finished
        # than blocks begun), when the cgroup's CPU quota throttled the process for more than a tenth of the region, or
        # when the process used 95 % of the CPUs it may use for the whole region
        throttled_s = (thr1[2] - thr0[2]) if thr0 and thr1 else 0.0
        cpu_per_step = (cpu1 - cpu0) / args.steps
        cpu_saturated = cpu_per_step >= 0.95 * cores * (elapsed / args.steps)
        quota_bound = throttled_s > 0.1 * elapsed
        gpu_ms = 1e3 * gpu_s / args.steps
        wait_ms = 1e3 * collect_s / args.steps
        blocks_done = max(1, max(b1 - b0, f1 - f0))
        what = ("BWT only" if coder is None else "BWT+encode")
        # What the region PROVES: `steps` blocks were begun in it, `finished` records were finished by
        # the host half in it.  The value is the smaller of the two rates; the closed loop (every
        # block of the stream, fill and drain included) is reported beside it.
        counted = min(args.steps, finished)
        value = world * counted * size / 1e6 / elapsed
        all_blocks = fill_blocks + args.steps
        wall_s = (fill_ms + drain_ms) * 1e-3 + elapsed
        closed = world * all_blocks * size / 1e6 / wall_s
        # blocks a stream needs before its closed-loop rate is within 10 % of `value`
        overhead_s = max(0.0, wall_s - all_blocks * elapsed / args.steps)
        stream_for_rate = int(9.0 * overhead_s / (elapsed / args.steps)) + 1
        out = {
            "metric": "MB/s compressed (%s) on %d MiB block" % (what, args.size_mib)
                      if coder else "MB/s transformed (BWT only) on %d MiB block" % args.size_mib,
            "value": round(value, 2), "unit": "MB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8/u32/u64 integer", "data": "synthetic",
            "config": {"workload": ("%s: %d MiB enwik8-style synthetic text blocks (splitmix64 token generator, "
                                    "seeds %s, %d distinct blocks cycled), 8 starting points, one block per step per GPU"
                                    % ("C5" if coder is None and args.size_mib >= 1024 else "C3" if world == 1 else "C4",
                                       args.size_mib, "3.." if world == 1 else "30+rank+8j", nblk))
                                   if args.workload == "c3" else
                                   ("EXTRA (not the headline workload) %s: %d MiB, %s; the one block every step"
                                    % (args.workload, args.size_mib, workload_what)),
                       "timed_region": "block in page-locked host memory -> " +
                                       ("transformed block in HBM" if coder is None else "record in host memory") +
                                       "; upload of block i+1 overlaps the kernels of block i" +
                                       ("; %d-deep block pipeline: K blocks begun and K records collected inside the "
                                        "region, value counts min(K, records the host finished inside it); fill and "
                                        "drain outside, closed loop reported beside" % depth if coder == "B" else ""),
                       "stages": "BWT (suffix sort + BWT + LFpowers + freqs) on the GPU" +
                                 (" + 'B' wavelet coder: run scanner, tree bit vectors, traversal order and gap "
                                  "flags on the GPU; adaptive models and the range coder (one serial chain per "
                                  "section, by the format) on %d host threads" % threads if coder == "B" else
                                  " + 'H' run-length/Huffman coder on the GPU, record copied to the host" if coder == "H" else ""),
                       "coder": coder, "blocks_per_gpu_per_step": 1, "parallelism": "block farm, no collective",
                       "blocks_under_way": depth},
            "single_block_ms": round(single_ms, 1), "single_block_MBps": round(size / 1e6 / (single_ms * 1e-3), 1), "h2d_ms": round(h2d_ms, 2),
            "fill_ms": round(fill_ms, 1), "drain_ms": round(drain_ms, 1),
            "gpu_ms_per_step": round(gpu_ms, 2), "collect_wait_ms_per_step": round(wait_ms, 2),
            "issue_rate_MBps": round(total_mb / elapsed, 2),
            "host_blocks_finished_in_region": finished,
            "closed_loop": {"blocks": all_blocks, "wall_s": round(wall_s, 3), "MBps": round(closed, 2),
                            "what": "every block of the stream / (fill + timed region + drain)"},
            "stream_blocks_for_rate": stream_for_rate,
            "host_mem_per_rank_gb": round(mem_per_rank, 1),
            "host_staging_peak_gb_rank0": round(hip.host_staging_bytes()[1] / 1e9, 2),
            "host_rss_peak_gb_rank0": round(__import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6, 1),
            "host_rss_gb_rank0": rss,
            "host_bound": bool(coder == "B" and (wait_ms > 0.1 * step_ms or finished < args.steps or quota_bound or cpu_saturated)),
            "host_bound_why": ([w for w, c in (("collect wait above a tenth of a step", wait_ms > 0.1 * step_ms),
                                               ("fewer records finished in the region than blocks begun", finished < args.steps),
                                               ("cgroup CPU quota throttled the process for more than a tenth of the region", quota_bound),
                                               ("the process used 95 % of its usable CPUs over the whole region", cpu_saturated)) if c]
                               if coder == "B" else []),
            # a region shorter than three block latencies mostly collects records that were begun before it
            "region_over_block_latency": round(elapsed / block_latency, 2) if coder == "B" and block_latency else None,
            "region_short": bool(coder == "B" and block_latency and elapsed < 3.0 * block_latency),
            "host_threads_per_rank": threads if coder == "B" else 0, "host_cpus_usable": cores,
            "host_cpus_of_rank0": _ranges(my_cpus), "numa_node_of_rank0": numa_node,
            "per_rank": [{"rank": r, "host_threads": per_rank["host_threads"][r], "numa_node": per_rank["numa_node"][r],
                          "cpus_in_slice": per_rank["cpus_in_slice"][r],
                          "staging_peak_gb": round(per_rank["staging_peak_MB"][r] / 1e3, 2),
                          "gpu_ms_per_step": round(per_rank["gpu_us_per_step"][r] / 1e3, 2),
                          "collect_wait_ms_per_step": round(per_rank["collect_wait_us_per_step"][r] / 1e3, 2),
                          "records_finished_in_region": per_rank["records_finished_in_region"][r],
                          "depth_used": per_rank["depth_used"][r], "depth_needed": per_rank["depth_needed"][r],
                          "process_cpu_s_per_step": round(per_rank["process_cpu_ms_per_step"][r] / 1e3, 3),
                          "host_bound": bool(coder == "B" and (per_rank["collect_wait_us_per_step"][r] > 100.0 * step_ms
                                                               or per_rank["records_finished_in_region"][r] < args.steps))}
                         for r in range(world)],
            "depth_default": depth_default, "depth_measured_after_fill": depth_measured,
            "block_latency_ms": round(1e3 * block_latency, 1) if coder == "B" else 0.0,
            "depth_needed": int(-(-(1e3 * block_latency if block_latency else single_ms) // max(step_ms, 1e-3))) if coder == "B" else 1,
            "host_core_s_per_block": round(((m1 - m0) + (c1 - c0)) / blocks_done, 3) if coder == "B" else 0.0,
            # what the kernel charged this process inside the timed region (all threads: workers, the thread
            # that feeds the GPU, the runtime's helpers), per step; and whether the cgroup's CPU quota bit
            "process_cpu_s_per_step": round((cpu1 - cpu0) / args.steps, 3),
            "cgroup_throttled": ({"periods": thr1[0] - thr0[0], "throttled_periods": thr1[1] - thr0[1],
                                  "throttled_s": round(thr1[2] - thr0[2], 3)} if thr0 and thr1 else None),
            "host_model_s_per_block": round((m1 - m0) / blocks_done, 3) if coder == "B" else 0.0,
            "host_coder_s_per_block": round((c1 - c0) / blocks_done, 3) if coder == "B" else 0.0,
            "device_ms_bwt": round(statistics.mean(dev_ms), 3), "rounds": st.rounds,
            "R_eff": round(r_eff, 3),
            "compressed_bytes": int(comp_bytes),
            "roofline": roof,
        }
        if coder == "B" and cores // max(world, 1) < 16:
            out["warning"] = ("only %d host threads per rank: the 'B' coder's host half needs about 16 per GPU "
                              "to keep up with the device half" % threads)
        out["degraded"] = degraded
        if corpus is not None:
            out["corpus"] = corpus
        if args.cpu_c4 and world == 1:
            out["cpu_baseline_c4"] = cpu_baseline_c4(args.cpu_size_mib or args.size_mib, coder or "", my_cpus or sorted(os.sched_getaffinity(0)))
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_size_mib or args.size_mib, seeds[0], coder or "", args.cpu_runs)
        print(json.dumps(out), flush=True)
    farm.close()
    ctx.close()


if __name__ == "__main__":
    main()
