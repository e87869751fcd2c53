"""CPU tests of the product's host-side 'H' coder logic (entropy_host.cpp, reached through
the host-only C-ABI entry points) against the oracle and the reference's known answers."""
import json
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _p(a):
    return a.ctypes.data_as(__import__("ctypes").c_void_p)


def _host():
    from bwtc_amd import hip
    return hip.load()


def _rand_freqs(rng):
    kind = rng.integers(0, 5)
    f = np.zeros(256, np.uint64)
    nsym = int(rng.integers(1, 257))
    syms = rng.choice(256, nsym, replace=False)
    if kind == 0:
        f[syms] = rng.integers(1, 10, nsym)
    elif kind == 1:
        f[syms] = rng.integers(1, 1 << 28, nsym)
    elif kind == 2:
        f[syms] = 1
    elif kind == 3:   # Fibonacci-like: deep trees
        a, b = 1, 1
        for s in syms[:40]:
            f[s] = a
            a, b = b, a + b
    else:
        f[syms] = (2 ** rng.integers(0, 20, nsym)).astype(np.uint64)
    return f


def test_huffman_lengths_codes_shape_match_oracle(oracle):
    H, L = _host(), oracle.lib()
    rng = np.random.default_rng(5)
    for _ in range(400):
        f = _rand_freqs(rng)
        cl_h = np.zeros(256, np.uint8)
        H.bwtc_hip_host_huffman_lengths(_p(f), _p(cl_h))
        cl_o = np.zeros(256, np.uint32)
        L.orc_huffman_lengths(_p(f), _p(cl_o))
        assert (cl_h == cl_o).all()
        code_h = np.zeros(256, np.uint32)
        code_o = np.zeros(256, np.uint32)
        H.bwtc_hip_host_huffman_codes(_p(cl_h), _p(code_h))
        L.orc_huffman_codes(_p(cl_o), _p(code_o))
        assert (code_h == code_o).all()
        bits = np.zeros(16384, np.uint8)
        nb = L.orc_serialize_shape(_p(cl_o), _p(bits))
        want = np.packbits(bits[:nb]).tobytes()
        out = np.zeros(4096, np.uint8)
        n = H.bwtc_hip_host_serialize_shape(_p(cl_h), _p(out), out.size)
        assert out[:n].tobytes() == want
        # Kraft equality: a complete prefix code
        used = cl_h[cl_h > 0].astype(np.float64)
        if used.size > 1:
            assert abs(np.sum(2.0 ** -used) - 1.0) < 1e-9


def test_huffman_lengths_known_answers():
    H = _host()
    ka = json.load(open(os.path.join(G, "utils_known_answers.json")))
    for c in ka["huffman_lengths"]:
        f = np.zeros(256, np.uint64)
        for k, v in c["freqs"].items():
            f[ord(k)] = v
        cl = np.zeros(256, np.uint8)
        H.bwtc_hip_host_huffman_lengths(_p(f), _p(cl))
        order = sorted(c["freqs"].items(), key=lambda kv: (kv[1], ord(kv[0])))
        assert [int(cl[ord(k)]) for k, _ in order] == c["sorted_lengths"], c["ref"]


def test_sections_and_block_header_match_oracle(oracle):
    H, L = _host(), oracle.lib()
    rng = np.random.default_rng(9)
    for _ in range(200):
        f = np.zeros(256, np.uint32)
        nsym = int(rng.integers(1, 257))
        f[rng.choice(256, nsym, replace=False)] = rng.integers(1, int(rng.choice([50, 9999, 10001, 10 ** 6])), nsym)
        a = np.zeros(256, np.uint32)
        b = np.zeros(256, np.uint32)
        na = H.bwtc_hip_host_sections(_p(f), _p(a))
        nb = L.orc_sections(_p(f), _p(b))
        assert na == nb and (a[:na] == b[:nb]).all() and int(a[:na].sum()) == int(f.sum())
        n_lf = int(rng.integers(1, 257))
        lf = rng.integers(0, 1 << 31, n_lf).astype(np.uint32)
        oa = np.zeros(2048, np.uint8)
        ob = np.zeros(2048, np.uint8)
        ka = H.bwtc_hip_host_bwtblock_header(_p(lf), n_lf, _p(oa), oa.size)
        kb = L.orc_write_bwtblock_header(_p(lf), n_lf, _p(ob))
        assert ka == kb and (oa[:ka] == ob[:kb]).all()
