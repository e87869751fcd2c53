"""ctypes bindings for the parity checker (oracle/liboracle.so and, when present,
oracle/_ref/libbwtc_ref.so = the reference's own libdivsufsort).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under bwtc_amd/ may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_vp = ctypes.c_void_p
_u32 = ctypes.c_uint32
_sz = ctypes.c_size_t


def _ptr(a):
    return a.ctypes.data_as(_vp)


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True,
                   stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        p = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(p):
            build()
        L = ctypes.CDLL(p)
        L.orc_n_lf.restype = _u32
        L.orc_n_lf.argtypes = [_u32, _u32]
        L.orc_suffix_array.argtypes = [_vp, _u32, _vp]
        L.orc_bwt_raw.argtypes = [_vp, _u32, _vp, _u32, _vp]
        L.orc_bwt_block.argtypes = [_vp, _u32, _u32, _vp, _vp, _vp]
        L.orc_inverse_bwt_block.argtypes = [_vp, _u32, _vp, _u32]
        L.orc_pack_integer.restype = ctypes.c_uint64
        L.orc_pack_integer.argtypes = [ctypes.c_uint64, _vp]
        L.orc_run_frequencies.restype = ctypes.c_uint64
        L.orc_run_frequencies.argtypes = [_vp, _vp, _sz]
        L.orc_runs_and_characters.restype = ctypes.c_uint64
        L.orc_runs_and_characters.argtypes = [_vp, _vp, _sz, _vp, _vp, _sz, _vp]
        L.orc_huffman_lengths.argtypes = [_vp, _vp]
        L.orc_huffman_codes.argtypes = [_vp, _vp]
        L.orc_binary_code.restype = _sz
        L.orc_binary_code.argtypes = [_sz, _sz, _sz, _vp]
        L.orc_binary_interpolative_code.restype = _sz
        L.orc_binary_interpolative_code.argtypes = [_vp, _sz, _sz, _vp]
        L.orc_serialize_shape.restype = _sz
        L.orc_serialize_shape.argtypes = [_vp, _vp]
        L.orc_sections.restype = _u32
        L.orc_sections.argtypes = [_vp, _vp]
        L.orc_write_bwtblock_header.restype = _sz
        L.orc_write_bwtblock_header.argtypes = [_vp, _u32, _vp]
        L.orc_huffman_encode_block.restype = _sz
        L.orc_huffman_encode_block.argtypes = [_vp, _u32, _vp, _u32, _vp, _vp, _sz]
        L.orc_compress_H.restype = _sz
        L.orc_compress_H.argtypes = [_vp, _sz, _sz, _u32, _vp, _sz]
        L.orc_decompress_H.restype = _sz
        L.orc_decompress_H.argtypes = [_vp, _sz, _vp, _sz]
        L.orc_compress_B.restype = _sz
        L.orc_compress_wavelet.restype = _sz
        L.orc_compress_wavelet.argtypes = [ctypes.c_char, _vp, _sz, _sz, _u32, _vp, _sz]
        L.orc_wavelet_encode_block_with.restype = _sz
        L.orc_wavelet_encode_block_with.argtypes = [ctypes.c_char, _vp, _u32, _vp, _u32, _vp, _vp, _sz]
        L.orc_compress_B.argtypes = [_vp, _sz, _sz, _u32, _vp, _sz]
        L.orc_wavelet_encode_block.restype = _sz
        L.orc_wavelet_encode_block.argtypes = [_vp, _u32, _vp, _u32, _vp, _vp, _sz]
        L.orc_compress_bound.restype = _sz
        L.orc_compress_bound.argtypes = [_sz]
        L.orc_decompress_wavelet.restype = _sz
        L.orc_decompress_wavelet.argtypes = [_vp, _sz, _vp, _sz, _vp]
        L.orc_min_heap_order.restype = None
        L.orc_min_heap_order.argtypes = [_vp, _vp, _sz, _vp]
        L.orc_create_huffman_shape.restype = _sz
        L.orc_create_huffman_shape.argtypes = [_vp, _vp, _vp, _vp, _vp]
        L.orc_gamma_code.restype = _sz
        L.orc_gamma_code.argtypes = [_sz, _vp]
        L.orc_fixed_integer_code.restype = _sz
        L.orc_fixed_integer_code.argtypes = [_u32, _u32, _vp]
        L.orc_wavelet_symbol_codes.restype = None
        L.orc_wavelet_symbol_codes.argtypes = [_vp, _vp, _vp]
        # pair-replacing pre-stage (oracle/prepr_oracle.cpp)
        L.orc_grammar_new.restype = _vp
        L.orc_grammar_new.argtypes = []
        L.orc_grammar_free.restype = None
        L.orc_grammar_free.argtypes = [_vp]
        L.orc_grammar_rules.restype = _u32
        L.orc_grammar_rules.argtypes = [_vp]
        L.orc_grammar_specials.restype = _u32
        L.orc_grammar_specials.argtypes = [_vp]
        L.orc_grammar_is_special.argtypes = [_vp, ctypes.c_uint]
        L.orc_grammar_frequencies.restype = None
        L.orc_grammar_frequencies.argtypes = [_vp, _vp]
        L.orc_grammar_write.restype = _sz
        L.orc_grammar_write.argtypes = [_vp, _vp, _sz]
        L.orc_grammar_read.restype = _sz
        L.orc_grammar_read.argtypes = [_vp, _vp, _sz]
        L.orc_grammar_span.restype = _sz
        L.orc_grammar_span.argtypes = [_vp, _sz]
        L.orc_grammar_add_rule.restype = None
        L.orc_grammar_add_rule.argtypes = [_vp, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint]
        L.orc_grammar_expand_alphabet.restype = _sz
        L.orc_grammar_expand_alphabet.argtypes = [_vp, _vp, _sz, _vp, _sz, _vp]
        L.orc_pair_statistics.restype = None
        L.orc_pair_statistics.argtypes = [_vp, _vp, _sz, _vp, _vp]
        L.orc_pair_replace_round.restype = _sz
        L.orc_pair_replace_round.argtypes = [_vp, _vp, _sz, _vp, _vp]
        L.orc_precompress.restype = _sz
        L.orc_precompress.argtypes = [_vp, ctypes.c_char_p, _vp, _sz]
        L.orc_postprocess.restype = _sz
        L.orc_postprocess.argtypes = [_vp, _vp, _sz, _vp, _sz]
        L.orc_postprocess_bytes.restype = _sz
        L.orc_postprocess_bytes.argtypes = [_vp, _sz, _vp, _sz, _vp, _sz]
        L.orc_compress_prepr.restype = _sz
        L.orc_compress_prepr.argtypes = [ctypes.c_char, ctypes.c_char_p, _vp, _sz, ctypes.c_uint64, _u32, _vp, _sz]
        _lib = L
    return _lib


def ref():
    """The reference's libdivsufsort (oracle/_ref), or None when it was not built."""
    global _ref
    if _ref is None:
        p = os.path.join(ORACLE_DIR, "_ref", "libbwtc_ref.so")
        if not os.path.exists(p):
            return None
        R = ctypes.CDLL(p)
        R.ref_bwt_raw.argtypes = [_vp, _u32, _vp, _u32, _vp]
        R.ref_bwt_block.argtypes = [_vp, _u32, _u32, _vp, _vp, _vp]
        _ref = R
    return _ref


def _bwt_block(fn, data, sp, pad=0xA5):
    data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    size = data.size
    buf = np.empty(size + 1, np.uint8)
    buf[:size] = data
    buf[size] = pad
    lf = np.zeros(256, np.uint32)
    nlf = _u32(0)
    freqs = np.zeros(256, np.uint32)
    fn(_ptr(buf), size, sp, _ptr(lf), ctypes.byref(nlf), _ptr(freqs))
    assert buf[size] == pad, "byte after the block must be preserved"
    return buf[:size].copy(), lf[:nlf.value].copy(), freqs


def oracle_bwt_block(data, sp=8):
    """(bwt bytes, LFpowers, freqs) per BWTManager::doTransform(block, freqs)."""
    return _bwt_block(lib().orc_bwt_block, data, sp)


def ref_bwt_block(data, sp=8):
    return _bwt_block(ref().ref_bwt_block, data, sp)


def oracle_inverse_bwt_block(bwt, lf):
    b = np.array(bwt, dtype=np.uint8, copy=True)
    lfa = np.array(lf, dtype=np.uint32)
    rc = lib().orc_inverse_bwt_block(_ptr(b), b.size, _ptr(lfa), lfa.size)
    return rc, b


def oracle_suffix_array(T):
    T = np.ascontiguousarray(T, dtype=np.uint8)
    sa = np.zeros(T.size, np.uint32)
    lib().orc_suffix_array(_ptr(T), T.size, _ptr(sa))
    return sa


def oracle_runs_and_characters(section):
    """(run_freqs[256], total runs, {run length: count}) of one section."""
    section = np.ascontiguousarray(section, dtype=np.uint8)
    rf = np.zeros(256, np.uint64)
    cap = section.size + 1
    dl = np.zeros(cap, np.uint32)
    dc = np.zeros(cap, np.uint32)
    npairs = ctypes.c_size_t(0)
    runs = lib().orc_runs_and_characters(_ptr(rf), _ptr(section), section.size, _ptr(dl), _ptr(dc), cap,
                                         ctypes.byref(npairs))
    return rf, int(runs), dict(zip(dl[:npairs.value].tolist(), dc[:npairs.value].tolist()))


def oracle_sections(freqs):
    freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
    out = np.zeros(256, np.uint32)
    n = lib().orc_sections(_ptr(freqs), _ptr(out))
    return out[:n].copy()


def oracle_huffman_encode_block(bwt, lf, freqs):
    bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
    lf = np.ascontiguousarray(lf, dtype=np.uint32)
    freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
    cap = lib().orc_compress_bound(bwt.size)
    out = np.zeros(cap, np.uint8)
    n = lib().orc_huffman_encode_block(_ptr(bwt), bwt.size, _ptr(lf), lf.size, _ptr(freqs),
                                       _ptr(out), cap)
    return out[:n].copy()


def oracle_compress_H(data, block_size, sp=8):
    data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    data = np.ascontiguousarray(data)
    nblocks = (data.size + block_size - 1) // max(block_size, 1) + 1
    cap = lib().orc_compress_bound(data.size) + nblocks * 8192
    out = np.zeros(cap, np.uint8)
    n = lib().orc_compress_H(_ptr(data), data.size, block_size, sp, _ptr(out), cap)
    return out[:n].copy()


def oracle_wavelet_encode_block(bwt, lf, freqs):
    bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
    lf = np.ascontiguousarray(lf, dtype=np.uint32)
    freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
    cap = lib().orc_compress_bound(bwt.size)
    out = np.zeros(cap, np.uint8)
    n = lib().orc_wavelet_encode_block(_ptr(bwt), bwt.size, _ptr(lf), lf.size, _ptr(freqs), _ptr(out), cap)
    return out[:n].copy()


def oracle_compress_B(data, block_size, sp=8):
    data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    data = np.ascontiguousarray(data)
    nblocks = (data.size + block_size - 1) // max(block_size, 1) + 1
    cap = lib().orc_compress_bound(data.size) + nblocks * 8192
    out = np.zeros(cap, np.uint8)
    n = lib().orc_compress_B(_ptr(data), data.size, block_size, sp, _ptr(out), cap)
    return out[:n].copy()


def oracle_compress_wavelet(coder, data, block_size, sp=8):
    """Whole stream with wavelet coder letter `coder` ('B', 'b', 'u'); None for 'm' / 'M'."""
    data = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else data
    data = np.ascontiguousarray(data)
    nblocks = (data.size + block_size - 1) // max(block_size, 1) + 1
    cap = lib().orc_compress_bound(data.size) + nblocks * 8192
    out = np.zeros(cap, np.uint8)
    n = lib().orc_compress_wavelet(coder.encode(), _ptr(data), data.size, block_size, sp, _ptr(out), cap)
    return out[:n].copy() if n else None


def oracle_wavelet_encode_block_with(coder, bwt, lf, freqs):
    bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
    lf = np.ascontiguousarray(lf, dtype=np.uint32)
    freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
    cap = lib().orc_compress_bound(bwt.size)
    out = np.zeros(cap, np.uint8)
    n = lib().orc_wavelet_encode_block_with(coder.encode(), _ptr(bwt), bwt.size, _ptr(lf), lf.size, _ptr(freqs),
                                            _ptr(out), cap)
    return out[:n].copy()


def oracle_decompress_H(stream, max_size):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    out = np.zeros(max(max_size, 1), np.uint8)
    n = lib().orc_decompress_H(_ptr(stream), stream.size, _ptr(out), out.size)
    if n == ctypes.c_size_t(-1).value:
        return None
    return out[:n].copy()


def oracle_decompress_wavelet(stream, max_size, want_bwt=False):
    """Decompressor::decompress over a 'B' / 'b' / 'u' stream by the oracle's literal decoder;
    None on malformed input.  want_bwt: also return the blocks' transformed bytes."""
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    out = np.zeros(max(max_size, 1), np.uint8)
    bwt = np.zeros(max(max_size, 1), np.uint8) if want_bwt else None
    n = lib().orc_decompress_wavelet(_ptr(stream), stream.size, _ptr(out), out.size,
                                     _ptr(bwt) if want_bwt else None)
    if n == ctypes.c_size_t(-1).value:
        return (None, None) if want_bwt else None
    return (out[:n].copy(), bwt[:n].copy()) if want_bwt else out[:n].copy()


# ---- pair-replacing pre-stage (`--prepr p...`; oracle/prepr_oracle.cpp) --------------------------------
class OracleGrammar:
    """bwtc::Grammar as the oracle restates it."""

    def __init__(self):
        self.h = lib().orc_grammar_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_grammar_free(self.h)
            self.h = None

    @property
    def rules(self):
        return int(lib().orc_grammar_rules(self.h))

    @property
    def specials(self):
        return int(lib().orc_grammar_specials(self.h))

    def is_special(self, c):
        return bool(lib().orc_grammar_is_special(self.h, int(c)))

    def frequencies(self):
        f = np.zeros(256, np.uint64)
        lib().orc_grammar_frequencies(self.h, _ptr(f))
        return f

    def write(self):
        out = np.zeros(1 << 16, np.uint8)
        n = lib().orc_grammar_write(self.h, _ptr(out), out.size)
        assert n != ctypes.c_size_t(-1).value
        return out[:n].copy()

    def read(self, raw):
        raw = np.ascontiguousarray(raw, np.uint8)
        return int(lib().orc_grammar_read(self.h, _ptr(raw), raw.size))


def oracle_pair_statistics(data, grammar=None):
    data = np.ascontiguousarray(data, np.uint8)
    f = np.zeros(256, np.uint64)
    pf = np.zeros(1 << 16, np.uint64)
    lib().orc_pair_statistics(grammar.h if grammar else None, _ptr(data), data.size, _ptr(f), _ptr(pf))
    return f, pf


def oracle_pair_replace_round(grammar, data):
    """One PairReplacer over `data` with `grammar` (updated): (replacements made, new bytes)."""
    data = np.ascontiguousarray(data, np.uint8)
    out = np.zeros(2 * data.size + 8, np.uint8)
    n = ctypes.c_size_t(0)
    rep = lib().orc_pair_replace_round(grammar.h, _ptr(data), data.size, _ptr(out), ctypes.byref(n))
    return int(rep), out[:n.value].copy()


def oracle_precompress(grammar, options, data):
    buf = np.ascontiguousarray(data, np.uint8).copy()
    n = lib().orc_precompress(grammar.h, options.encode(), _ptr(buf), buf.size)
    return buf[:n].copy()


def oracle_postprocess(grammar, data, max_size):
    data = np.ascontiguousarray(data, np.uint8)
    out = np.zeros(max_size, np.uint8)
    n = lib().orc_postprocess(grammar.h, _ptr(data), data.size, _ptr(out), out.size)
    if n == ctypes.c_size_t(-1).value:
        return None
    return out[:n].copy()


def oracle_compress_prepr(coder, options, data, mem_limit, sp=8):
    """Compressor::compress with `--prepr options --mem` (mem_limit in bytes, Compressor.cpp:77-97)."""
    data = np.ascontiguousarray(data, np.uint8)
    out = np.zeros(int(lib().orc_compress_bound(data.size)) * 2 + 4096, np.uint8)
    n = lib().orc_compress_prepr(coder.encode(), options.encode(), _ptr(data), data.size, int(mem_limit), sp, _ptr(out), out.size)
    assert n > 0
    return out[:n].copy()
