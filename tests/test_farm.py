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
