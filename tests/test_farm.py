"""CPU test of the N>1 path: a world_size-2 (and 3) gloo farm.  Checks the block assignment
(every block exactly once, round-robin), the timing contract (warmup untimed, K timed steps,
MAX over ranks) and the in-order collection of the encoded records."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_blocks", [(2, 5), (3, 8)])
def test_gloo_farm(tmp_path, world, n_blocks):
    out = str(tmp_path / "res")
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "farm_worker.py"), out,
                                       str(n_blocks)], env=env))
    for p in procs:
        assert p.wait(timeout=120) == 0
    res = [json.load(open("%s.%d" % (out, r))) for r in range(world)]
    seen = sorted(i for r in res for i in r["mine"])
    assert seen == list(range(n_blocks))
    for r in res:
        assert r["mine"] == list(range(r["rank"], n_blocks, world))
        assert r["calls"] == 4                       # 1 warmup + 3 timed
        assert r["total_units"] == n_blocks
        # the slowest rank sleeps 0.01 * world per step; MAX over ranks is what everyone reports
        assert r["elapsed"] >= 3 * 0.01 * world * 0.9
        assert abs(r["elapsed"] - res[0]["elapsed"]) < 1e-9
    assert res[0]["ordered"] == ["block %d" % i for i in range(n_blocks)]


@pytest.mark.parametrize("world,two_nodes", [(2, False), (3, False), (2, True)])
def test_gloo_farm_ranks_get_disjoint_cpu_slices(tmp_path, world, two_nodes):
    """Every rank pins its worker threads (and itself) to its own slice of the host's CPUs: the
    ranks of one NUMA node split that node's CPUs; no CPU is handed out twice."""
    out = str(tmp_path / "res")
    port = _free_port()
    usable = sorted(os.sched_getaffinity(0))
    if len(usable) < world:
        pytest.skip("fewer CPUs than ranks")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if two_nodes:
            env["FARM_TWO_NODES"] = "1"
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "farm_worker.py"), out, "4"], env=env))
    for p in procs:
        assert p.wait(timeout=120) == 0
    res = [json.load(open("%s.%d" % (out, r))) for r in range(world)]
    seen = []
    for r in res:
        assert r["cpus"], "a rank got no CPU"
        assert r["affinity_now"] == sorted(r["cpus"])          # the rank really runs there
        assert set(r["cpus"]) <= set(usable)
        seen += r["cpus"]
    if not two_nodes:
        assert len(seen) == len(set(seen)), "a CPU was handed to two ranks"
        assert abs(len(res[0]["cpus"]) - len(res[-1]["cpus"])) <= 1
    else:
        # ranks on different (made-up) nodes: each is alone on its node; a node the system does not
        # know falls back to all usable CPUs, a known one is cut to its CPUs
        for r in res:
            assert len(r["cpus"]) >= 1


def test_gloo_farm_eight_ranks_on_a_made_up_two_node_topology(tmp_path):
    """The scaling row's shape without its hardware: 8 ranks, GPUs 0-3 on NUMA node 0 and 4-7 on node 1 (a made-up
    topology: BWTC_HIP_NODE<k>_CPULIST files give each node half of this box's CPUs).  Every rank's slice lies on its
    GPU's node, the four ranks of a node split it without overlap and within one CPU of evenly, nobody is left without
    a CPU -- and the farm's other collectives (barrier, max, gather in block order) work at world size 8."""
    usable = sorted(os.sched_getaffinity(0))
    if len(usable) < 8:
        pytest.skip("fewer than 8 CPUs")
    half = len(usable) // 2
    lists = [usable[:half], usable[half:]]
    for k, cpus in enumerate(lists):
        (tmp_path / ("node%d" % k)).write_text(",".join(str(c) for c in cpus) + "\n")
    out = str(tmp_path / "res")
    port = _free_port()
    world, n_blocks = 8, 19
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FARM_NODE_OF_RANK="0,0,0,0,1,1,1,1",
                   BWTC_HIP_NODE0_CPULIST=str(tmp_path / "node0"), BWTC_HIP_NODE1_CPULIST=str(tmp_path / "node1"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "farm_worker.py"), out, str(n_blocks)], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    res = [json.load(open("%s.%d" % (out, r))) for r in range(world)]
    seen = []
    for r in res:
        node = 0 if r["rank"] < 4 else 1
        assert r["node"] == node
        assert r["cpus"] and set(r["cpus"]) <= set(lists[node]), (r["rank"], r["cpus"])
        assert r["mine"] == list(range(r["rank"], n_blocks, world))
        seen += r["cpus"]
    assert len(seen) == len(set(seen)), "a CPU was handed to two ranks"
    for node in (0, 1):
        sizes = [len(r["cpus"]) for r in res if (r["rank"] >= 4) == bool(node)]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == len(lists[node])
    assert res[0]["ordered"] == [("block %d" % i)[:7] for i in range(n_blocks)]      # (the worker keeps seven characters)


def test_frame_stream_matches_oracle_framing(oracle):
    import numpy as np
    from bwtc_amd.farm import frame_stream
    data = np.frombuffer(b"abracadabra" * 50, np.uint8)
    blocks = [data[:300], data[300:]]
    recs = []
    for b in blocks:
        bwt, lf, fr = oracle.oracle_bwt_block(b, 4)
        recs.append(oracle.oracle_huffman_encode_block(bwt, lf, fr).tobytes())
    stream = frame_stream("H", recs, [b.size for b in blocks])
    assert stream == oracle.oracle_compress_H(data, 300, 4).tobytes()
    assert oracle.oracle_decompress_H(np.frombuffer(stream, np.uint8), data.size + 8).tobytes() == data.tobytes()
