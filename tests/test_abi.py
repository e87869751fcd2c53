"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads without a
GPU and exports every symbol include/bwtc_hip.h declares; argument checks that need no
device work behave as documented."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "bwtc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bwtc_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from bwtc_amd import hip
    lib = hip.load()
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    assert sorted(hip.EXPORTS) == names


def test_pure_host_entry_points():
    from bwtc_amd import hip
    lib = hip.load()
    assert b"gfx950" in lib.bwtc_hip_version()
    # BWTBlock.cpp:104-108 after the BWTManager.cpp:60-64 clamp
    assert lib.bwtc_hip_n_lf(256, 8) == 1
    assert lib.bwtc_hip_n_lf(257, 8) == 8
    assert lib.bwtc_hip_n_lf(1 << 20, 0) == 1
    assert lib.bwtc_hip_n_lf(1 << 20, 300) == 256
    assert lib.bwtc_hip_workspace_bytes(1 << 20) > 40 * (1 << 20)
    # bad arguments -> -1, no device needed
    assert lib.bwtc_hip_create(0, 1 << 20, None) == -1
    assert lib.bwtc_hip_bwt_block(None, None, 0, None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    from bwtc_amd import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libbwtc_hip.so")
    with pytest.raises(hip.BwtcHipError):
        hip.load()


def test_product_never_imports_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "bwtc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".c")):
                t = open(os.path.join(d, f), errors="ignore").read()
                if "oracle" in t.lower():
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_synthetic_generator_matches_numpy_definition():
    """bwtc_hip_synth (C++, used by bench.py and the host tools) produces the bytes of
    bwtc_amd/synth.py (SURVEY.md 8d generators) for all three kinds, including ragged sizes."""
    import numpy as np
    from bwtc_amd import hip, synth
    for kind, gen, seed in (("t", synth.gen_text, 3), ("d", synth.gen_dna, 2), ("r", synth.gen_random_bytes, 1),
                            ("t", synth.gen_text, 37)):
        for n in (1, 7, 33, 1000, (1 << 20) + 3):
            a = hip.synth_into(kind, seed, np.empty(n, np.uint8))
            assert (a == gen(n, seed)).all(), (kind, seed, n)
    with pytest.raises(hip.BwtcHipError):
        hip.synth_into("x", 1, np.empty(4, np.uint8))
