"""GPU tests of the C++ host mirror (bwtc_amd/host): the C++ test program shaped after the
reference's InverseBwtTest / LFpowersTest / CompressorAndDecompressorTest, and the `compress`
command line tool, whose output must decode (here: with the oracle's decoder) to the input."""
import os
import subprocess

import numpy as np
import pytest

from bwtc_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror_program():
    exe = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")
    assert os.path.exists(exe), "build with __graft_entry__.build() / make -C bwtc_amd/host"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0
    assert "all tests passed" in r.stdout


def test_compress_cli_roundtrip(tmp_path, oracle):
    exe = os.path.join(ROOT, "bwtc_amd", "host", "compress")
    assert os.path.exists(exe)
    data = np.concatenate([synth.gen_text(3_000_000, 3), synth.gen_random_bytes(500_000, 1),
                           np.zeros(100_000, np.uint8)])
    src = tmp_path / "input.bin"
    dst = tmp_path / "input.bin.bwtc"
    src.write_bytes(data.tobytes())
    # --mem 10 -> BWT blocks of 1.85 MB -> two blocks
    r = subprocess.run([exe, "-m", "10", "-s", "8", "--bwt", "g", "-e", "H", "-v", "1", str(src), str(dst)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    stream = np.frombuffer(dst.read_bytes(), np.uint8)
    assert stream[0] == ord("H")
    back = oracle.oracle_decompress_H(stream, data.size + 8)
    assert back is not None and back.tobytes() == data.tobytes()
    assert stream.tobytes() == oracle.oracle_compress_H(data, int(10 * 1000000 * 0.185), 8).tobytes()
    # and back through this repository's own uncompress (host 'H' decoder + GPU inverse BWT)
    unexe = os.path.join(ROOT, "bwtc_amd", "host", "uncompress")
    out = tmp_path / "roundtrip.bin"
    r = subprocess.run([unexe, "-v", "1", str(dst), str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data.tobytes()
    # the reference's default coder 'B' (no -e): five blocks, overlapped by the Compressor loop
    dstb = tmp_path / "input.B.bwtc"
    r = subprocess.run([exe, "-m", "4", "-s", "8", "-v", "1", str(src), str(dstb)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    sb = dstb.read_bytes()
    assert sb[:1] == b"B"
    assert sb == oracle.oracle_compress_B(data, int(4 * 1000000 * 0.185), 8).tobytes()
    outb = tmp_path / "roundtrip.B.bin"
    r = subprocess.run([unexe, "-v", "1", str(dstb), str(outb)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert outb.read_bytes() == data.tobytes()
    # another model letter through both tools
    dstu = tmp_path / "input.u.bwtc"
    outu = tmp_path / "roundtrip.u.bin"
    r = subprocess.run([exe, "-m", "10", "-e", "u", str(src), str(dstu)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert dstu.read_bytes() == oracle.oracle_compress_wavelet("u", data, int(10 * 1000000 * 0.185), 8).tobytes()
    r = subprocess.run([unexe, str(dstu), str(outu)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert outu.read_bytes() == data.tobytes()
    r = subprocess.run([exe, "-e", "m", str(src), str(dstu)], capture_output=True, text=True)
    assert r.returncode != 0
    # rejected choices, as the reference's validators do (compress.cpp:86-96)
    r = subprocess.run([exe, "--bwt", "d", str(src), str(dst)], capture_output=True, text=True)
    assert r.returncode != 0


def test_cli_block_farm_over_two_contexts_gives_the_sequential_stream(tmp_path, oracle):
    """`compress --devices 0,0` (bwtc_hip_farm.hpp: one thread + context + page-locked staging ring
    per entry, in-order writer, the 'B' coder's carried model state handed from context to context
    at bwtc_hip_wavelet_encode_queue): the stream must be byte-equal to the one-device stream and
    to the oracle's sequential encoder, for both coders, with more blocks than contexts."""
    exe = os.path.join(ROOT, "bwtc_amd", "host", "compress")
    unexe = os.path.join(ROOT, "bwtc_amd", "host", "uncompress")
    data = np.concatenate([synth.gen_text(5_300_000, 41), synth.gen_dna(1_000_000, 7), synth.gen_text(2_100_000, 43)])
    src = tmp_path / "farm.bin"
    src.write_bytes(data.tobytes())
    block = int(1 * 1000000 * 0.185)                  # -m 1 -> 185 000-byte blocks: 46 blocks
    for enc in ("B", "H"):
        one = tmp_path / ("one.%s.bwtc" % enc)
        for devices, dst in (("0,0", tmp_path / ("two.%s.bwtc" % enc)), ("0,0,0", tmp_path / ("three.%s.bwtc" % enc))):
            r = subprocess.run([exe, "-m", "1", "-e", enc, "--devices", devices, str(src), str(dst)],
                               capture_output=True, text=True, timeout=600)
            assert r.returncode == 0, r.stderr
        r = subprocess.run([exe, "-m", "1", "-e", enc, str(src), str(one)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        want = (oracle.oracle_compress_B if enc == "B" else oracle.oracle_compress_H)(data, block, 8).tobytes()
        assert one.read_bytes() == want, enc
        assert (tmp_path / ("two.%s.bwtc" % enc)).read_bytes() == want, enc
        assert (tmp_path / ("three.%s.bwtc" % enc)).read_bytes() == want, enc
    # the 'B' models of a farmed stream run on the device (the state-dependent passes at _queue time,
    # the block joins the host pipeline from a stream callback): the same stream with the models on the
    # worker threads, and when every block's device result is declared flagged (host fallback)
    for extra in ({"BWTC_HIP_MODELS": "host"}, {"BWTC_HIP_TEST_MODELS_FALLBACK": "1"}):
        dst = tmp_path / "two.B.alt.bwtc"
        r = subprocess.run([exe, "-m", "1", "-e", "B", "--devices", "0,0", str(src), str(dst)],
                           capture_output=True, text=True, timeout=600, env=dict(os.environ, **extra))
        assert r.returncode == 0, r.stderr
        assert dst.read_bytes() == (tmp_path / "two.B.bwtc").read_bytes(), extra
    out = tmp_path / "farm.out"
    r = subprocess.run([unexe, str(tmp_path / "two.B.bwtc"), str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data.tobytes()


def test_cli_roundtrip_full_size_block_default_coder(tmp_path):
    """BASELINE's configuration end to end: one 256 MiB text block, the default 'B' coder,
    compress -> uncompress (host wavelet decoder + GPU inverse BWT) gives the input back.
    Size-independent property; no oracle involved."""
    exe = os.path.join(ROOT, "bwtc_amd", "host", "compress")
    unexe = os.path.join(ROOT, "bwtc_amd", "host", "uncompress")
    data = synth.gen_text(256 << 20, 3)
    src = tmp_path / "text256.bin"
    dst = tmp_path / "text256.bwtc"
    out = tmp_path / "text256.out"
    src.write_bytes(data.tobytes())
    # --mem 1452 -> BWT block 0.185 * 1452e6 = 268.6 MB >= 256 MiB: one block
    r = subprocess.run([exe, "-m", "1452", "-s", "8", "-v", "1", str(src), str(dst)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    packed = dst.stat().st_size
    assert 40_000_000 < packed < 60_000_000, packed
    r = subprocess.run([unexe, "-v", "1", str(dst), str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    back = np.fromfile(out, np.uint8)
    assert back.size == data.size and (back == data).all()
    # three blocks (two full, one partial) overlapped by the Compressor loop, model state carried
    # across the block borders, decoded by one decoder object
    for f in (src, dst, out):
        f.unlink()
    data = np.concatenate([data, synth.gen_text(300 << 20, 31)])
    src.write_bytes(data.tobytes())
    r = subprocess.run([exe, "-m", "1452", "-s", "8", str(src), str(dst)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([unexe, str(dst), str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    back = np.fromfile(out, np.uint8)
    assert back.size == data.size and (back == data).all()


def test_cli_roundtrip_1GiB_single_block_default_coder(tmp_path):
    """The largest block the format carries (1 GiB, N = 2^30 + 1) through the default coder:
    2.2 billion steps in the stream kernels, one serial decoder pass, GPU inverse transform."""
    exe = os.path.join(ROOT, "bwtc_amd", "host", "compress")
    unexe = os.path.join(ROOT, "bwtc_amd", "host", "uncompress")
    data = synth.gen_text(1 << 30, 5)
    src = tmp_path / "t1g.bin"
    dst = tmp_path / "t1g.bwtc"
    out = tmp_path / "t1g.out"
    data.tofile(src)
    r = subprocess.run([exe, "-m", "5806", "-v", "1", str(src), str(dst)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    assert "outside the stream kernels" not in r.stderr          # the device route took it
    r = subprocess.run([unexe, str(dst), str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    back = np.fromfile(out, np.uint8)
    assert back.size == data.size and (back == data).all()


def _real_text(limit):
    """Real (not generated) text: the file BWTC_CORPUS names when there is one (SURVEY.md 8(d):
    enwik8 / enwik9), else Python sources and documentation found on the box, concatenated in
    sorted order -- anything written by people, so that the coders also see inputs that the
    64-token generator never makes (other section sizes, other run statistics)."""
    path = os.environ.get("BWTC_CORPUS")
    if path and os.path.exists(path):
        with open(path, "rb") as f:
            return f.read(limit), os.path.basename(path)
    buf = bytearray()
    import sysconfig
    roots = [sysconfig.get_paths()["stdlib"], "/usr/share/common-licenses"]
    seen = 0
    for root in roots:                                   # bounded walk: the standard library only, no site-packages
        for dirpath, dirnames, filenames in os.walk(root):
            dirnames[:] = sorted(d for d in dirnames if d not in ("site-packages", "dist-packages", "__pycache__", "test", "tests"))
            for name in sorted(filenames):
                seen += 1
                if len(buf) >= limit or seen > 20000:
                    break
                if not (name.endswith(".py") or name.endswith(".txt") or root.endswith("licenses")):
                    continue
                try:
                    with open(os.path.join(dirpath, name), "rb") as f:
                        buf += f.read(limit - len(buf))
                except OSError:
                    pass
            if len(buf) >= limit or seen > 20000:
                break
    return bytes(buf), "local files"


def test_real_text_roundtrip_and_oracle_parity(tmp_path, oracle):
    """BWTC_CORPUS (or real text found on the box) through `compress` (default coder 'B', models
    on the GPU) and `uncompress`; the stream is also the oracle's, byte for byte."""
    data, name = _real_text(24 << 20)
    if len(data) < (1 << 20):
        pytest.skip("no real text to be found here")
    src = tmp_path / "corpus.bin"
    dst = tmp_path / "corpus.bwtc"
    out = tmp_path / "corpus.out"
    src.write_bytes(data)
    exe = os.path.join(ROOT, "bwtc_amd", "host", "compress")
    unexe = os.path.join(ROOT, "bwtc_amd", "host", "uncompress")
    r = subprocess.run([exe, "-m", "50", "-v", "1", str(src), str(dst)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    stream = dst.read_bytes()
    arr = np.frombuffer(data, np.uint8)
    assert stream == oracle.oracle_compress_B(arr, int(50 * 1000000 * 0.185), 8).tobytes(), name
    r = subprocess.run([unexe, str(dst), str(out)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == data
    print("real text (%s): %d -> %d bytes, ratio %.3f" % (name, len(data), len(stream), len(stream) / len(data)))
