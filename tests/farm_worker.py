"""Worker for tests/test_farm.py: one rank of a world_size-N gloo farm on the CPU.  The block
transform is replaced by a stand-in (the farm logic is what is under test); the real one needs
a GPU and is covered by the -m gpu tests."""
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bwtc_amd.farm import Farm  # noqa: E402


def main():
    out_path, n_blocks = sys.argv[1], int(sys.argv[2])
    farm = Farm(backend="gloo")
    mine = farm.my_blocks(n_blocks)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (farm.rank + 1))      # rank-dependent step time

    elapsed = farm.timed(step, steps=3, warmup=1)
    records = {i: zlib.compress(("block %d" % i).encode() * 10) for i in mine}
    ordered = farm.gather_records(records, n_blocks)
    total_units = farm.sum(len(mine))
    # host CPUs of this rank's workers: every rank reports the same NUMA node here (a made-up one
    # for half of the ranks when FARM_TWO_NODES is set), the slices must not overlap
    node = farm.rank % 2 if os.environ.get("FARM_TWO_NODES") else -1
    if os.environ.get("FARM_NODE_OF_RANK"):            # a made-up topology: "0,0,0,0,1,1,1,1"
        node = int(os.environ["FARM_NODE_OF_RANK"].split(",")[farm.rank])
    cpus = farm.cpu_slice(node)
    os.sched_setaffinity(0, cpus or os.sched_getaffinity(0))
    res = {"rank": farm.rank, "world": farm.world, "mine": mine, "elapsed": elapsed,
           "calls": len(calls), "total_units": total_units, "cpus": cpus, "node": node,
           "affinity_now": sorted(os.sched_getaffinity(0))}
    if farm.rank == 0:
        res["ordered"] = [zlib.decompress(r).decode()[:7] for r in ordered]
    with open("%s.%d" % (out_path, farm.rank), "w") as f:
        json.dump(res, f)
    farm.close()


if __name__ == "__main__":
    main()
