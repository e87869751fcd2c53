"""Block farm over the GPUs of one node: one process per GPU, one context per process, rank r
takes BWT blocks r, r + world, ... (PrecompressorBlock::sliceIntoBlocks makes the blocks
independent, PrecompressorBlock.cpp:123-134; every encoded block carries its own 48-bit
length, Compressor.hpp:66-71).  The data path has NO collective; torch.distributed is used for
the barrier, the max/sum of scalars and (optionally) collecting the encoded records in
block order on rank 0.  Works with backend "nccl" (= RCCL, GPU box) and "gloo" (CPU tests).
"""
import os
import time

import torch
import torch.distributed as dist


class Farm:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = device
        self.backend = backend
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if backend is None:
                backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
            self.backend = backend
            kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world, **kw)

    # ---- work distribution -------------------------------------------------------------
    def my_blocks(self, n_blocks):
        """Indices of the blocks this rank transforms (round-robin, as SURVEY.md 8e)."""
        return list(range(self.rank, n_blocks, self.world))

    # ---- scalar reductions / barrier -----------------------------------------------------
    def _tensor(self, v, dtype=torch.float64):
        dev = self.device if (self.backend == "nccl" and self.device is not None) else torch.device("cpu")
        return torch.tensor([v], dtype=dtype, device=dev)

    def barrier(self):
        if self.world > 1:
            dist.barrier()
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize()

    def max(self, v):
        if self.world == 1:
            return float(v)
        t = self._tensor(v)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, v):
        if self.world == 1:
            return float(v)
        t = self._tensor(v)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    # ---- host CPUs -------------------------------------------------------------------------------
    def all_gather_int(self, v):
        """Every rank's value of a small integer, in rank order."""
        if self.world == 1:
            return [int(v)]
        dev = self.device if (self.backend == "nccl" and self.device is not None) else torch.device("cpu")
        mine = torch.tensor([int(v)], dtype=torch.int64, device=dev)
        parts = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine)
        return [int(p.item()) for p in parts]

    def cpu_slice(self, numa_node=-1):
        """CPUs for this rank's worker threads: the ranks whose GPUs sit on the same NUMA node split
        that node's CPUs (all usable CPUs when the node is unknown) into disjoint contiguous
        slices, in rank order.  One process per GPU on ONE host, as the bench and the farm run."""
        from bwtc_amd import hip
        nodes = self.all_gather_int(numa_node)
        same = [r for r in range(self.world) if nodes[r] == nodes[self.rank]]
        return hip.host_cpu_slice(numa_node, same.index(self.rank), len(same))

    # ---- timing contract of bench.py -----------------------------------------------------
    def timed(self, step, steps, warmup, drain=None):
        """warmup untimed steps, then `steps` steps bracketed by barrier+synchronize on both
        sides; returns the MAX over ranks of the elapsed seconds.  `drain` (optional) completes
        work the steps left under way; it runs after the warmup and, inside the timed region,
        after the last step."""
        for _ in range(warmup):
            step()
        if drain:
            drain()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if drain:
            drain()
        self.barrier()
        return self.max(time.perf_counter() - t0)

    # ---- result collection (host side, not part of the timed data path) -------------------
    def gather_records(self, records, n_blocks):
        """records: {block index: bytes} produced by this rank.  Returns on rank 0 the list of
        all n_blocks records in block order (None elsewhere)."""
        if self.world == 1:
            return [records[i] for i in range(n_blocks)]
        out = [None] * self.world if self.rank == 0 else None
        dist.gather_object(records, out, dst=0)
        if self.rank != 0:
            return None
        merged = {}
        for part in out:
            merged.update(part)
        assert sorted(merged) == list(range(n_blocks)), "a block was lost or duplicated"
        return [merged[i] for i in range(n_blocks)]

    def close(self):
        if self.world > 1 and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


def frame_stream(coder, records, block_sizes):
    """Compressor::compress framing (Compressor.cpp:55-58,104-115) around already encoded BWT
    blocks: coder letter, then per block a precompressor-block header (packed original size,
    packed slice count = 1, empty grammar byte) + the record, then the 0x00 terminator."""
    def packed(v):
        b = bytearray()
        while True:
            x = v & 0x7F
            v >>= 7
            b.append(x | (0x80 if v else 0))
            if not v:
                return bytes(b)
    out = bytearray(coder.encode())
    for rec, size in zip(records, block_sizes):
        out += packed(size) + packed(1) + b"\x00" + bytes(rec)
    out += b"\x00"
    return bytes(out)
