"""GPU tests of the pair-replacing pre-stage (`--prepr p...`): the device sweeps behind
bwtc_hip_precompress against the oracle's restatement of preprocessors/PairReplacer.cpp + Grammar.cpp, and whole
streams of `compress --prepr` over the matrix of test/CompressorAndDecompressorTest.cpp:61-205 (prep "p" ...
"ppppp", coders 'H' and 'B', several precompressor blocks and slices) byte for byte against the oracle's stream,
decoded by this repository's `uncompress` and by the oracle's decoders."""
import os
import subprocess

import numpy as np
import pytest

from bwtc_amd import hip, synth
from test_host_logic import _prepr_inputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    with hip.Context(0, 16 << 20) as c:
        yield c


@pytest.mark.parametrize("name,data", _prepr_inputs(), ids=[n for n, _ in _prepr_inputs()])
def test_precompress_on_the_device_matches_the_oracle(ctx, oracle, name, data):
    for opts in ("p", "ppp", "ppppp"):
        g, og = hip.Grammar(), oracle.OracleGrammar()
        mine = ctx.precompress(g, opts, data)
        want = oracle.oracle_precompress(og, opts, data)
        assert mine.size == want.size and (mine == want).all(), (name, opts)
        assert g.write().tobytes() == og.write().tobytes(), (name, opts)
        back = g.postprocess(mine, data.size + 8)
        assert back.size == data.size and (back == data).all()


def test_precompress_of_a_16_MiB_text_block(ctx, oracle):
    """Many tiles, counters flushed by many workgroups, five rounds over the shrinking text."""
    data = synth.gen_text(16 << 20, 21)
    g, og = hip.Grammar(), oracle.OracleGrammar()
    mine = ctx.precompress(g, "ppppp", data)
    want = oracle.oracle_precompress(og, "ppppp", data)
    assert mine.size == want.size and (mine == want).all()
    assert mine.size < 0.6 * data.size
    assert g.write().tobytes() == og.write().tobytes()
    back = g.postprocess(mine, data.size + 8)
    assert back.size == data.size and (back == data).all()


def test_pair_statistics_of_long_runs_on_the_device(ctx, oracle):
    """A pair of equal bytes replaced along runs that cross many tiles: the runs' parity comes from their starts."""
    rng = np.random.default_rng(3)
    parts = []
    for k in rng.integers(1, 30000, 400):
        parts.append(np.full(int(k), 120, np.uint8))
        parts.append(rng.integers(97, 101, int(rng.integers(1, 4))).astype(np.uint8))
    data = np.concatenate(parts)
    for opts in ("p", "pp"):
        g, og = hip.Grammar(), oracle.OracleGrammar()
        mine = ctx.precompress(g, opts, data)
        want = oracle.oracle_precompress(og, opts, data)
        assert mine.size == want.size and (mine == want).all(), opts
        assert g.write().tobytes() == og.write().tobytes()


@pytest.mark.parametrize("coder", ["H", "B"])
@pytest.mark.parametrize("prep", ["p", "pp", "ppppp"])
def test_compress_cli_with_prepr_gives_the_oracles_stream(tmp_path, oracle, coder, prep):
    exe = os.path.join(ROOT, "bwtc_amd", "host", "compress")
    unexe = os.path.join(ROOT, "bwtc_amd", "host", "uncompress")
    assert os.path.exists(exe) and os.path.exists(unexe)
    data = np.concatenate([synth.gen_text(1_600_000, 3), synth.gen_random_bytes(50_000, 1), np.zeros(20_000, np.uint8),
                           synth.gen_text(700_000, 4)])
    src = tmp_path / "input.bin"
    dst = tmp_path / "input.bwtc"
    out = tmp_path / "output.bin"
    src.write_bytes(data.tobytes())
    # --mem 1: precompressor blocks of 740 000 bytes, BWT blocks of (1 000 000 - precompressed size) / 4.5
    r = subprocess.run([exe, "-m", "1", "-s", "8", "-e", coder, "--prepr", prep, str(src), str(dst)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    stream = np.frombuffer(dst.read_bytes(), np.uint8)
    want = oracle.oracle_compress_prepr(coder, prep, data, 1_000_000)
    assert stream.size == want.size and (stream == want).all()
    back = oracle.oracle_decompress_H(stream, data.size + 64) if coder == "H" else oracle.oracle_decompress_wavelet(stream, data.size + 64)
    assert back is not None and back.tobytes() == data.tobytes()
    r = subprocess.run([unexe, str(dst), str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data.tobytes()


def test_compress_cli_rejects_unknown_prepr_letters(tmp_path):
    exe = os.path.join(ROOT, "bwtc_amd", "host", "compress")
    src = tmp_path / "x"
    src.write_bytes(b"abc" * 100)
    r = subprocess.run([exe, "--prepr", "px", str(src), str(tmp_path / "y")], capture_output=True, text=True)
    assert r.returncode != 0
