"""ctypes binding of libbwtc_hip.so (include/bwtc_hip.h).

This is the only way Python code in this repository reaches the GPU path.  There is no CPU
fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BWTC_HIP_LIB: another build of the library (development: tile-shape sweeps, scripts/dev/)
LIB_PATH = os.environ.get("BWTC_HIP_LIB") or os.path.join(_HERE, "lib", "libbwtc_hip.so")

_vp = ctypes.c_void_p
_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64


class BwtcHipError(RuntimeError):
    pass


class Stats(ctypes.Structure):
    _fields_ = [("n", _u32), ("rounds", _u32), ("active_sum", _u64), ("sort_pass_items", _u64),
                ("ms_total", ctypes.c_float), ("ms_sort", ctypes.c_float),
                ("route", _u32), ("finisher_entries", _u32), ("alg_bytes", _u64)]


class KernelTimers(ctypes.Structure):
    _fields_ = [("scatter_launches", _u64), ("scatter_bytes", _u64), ("scatter_ms", ctypes.c_double)]


EXPORTS = [
    "bwtc_hip_device_count", "bwtc_hip_version", "bwtc_hip_workspace_bytes", "bwtc_hip_create",
    "bwtc_hip_destroy", "bwtc_hip_stream", "bwtc_hip_get_stats", "bwtc_hip_set_profiling",
    "bwtc_hip_get_kernel_timers", "bwtc_hip_copy_probe", "bwtc_hip_test_gpu_lanes", "bwtc_hip_malloc", "bwtc_hip_free", "bwtc_hip_memcpy_to_device",
    "bwtc_hip_memcpy_to_host", "bwtc_hip_host_alloc", "bwtc_hip_host_free",
    "bwtc_hip_memcpy_to_device_async", "bwtc_hip_copy_wait", "bwtc_hip_wavelet_host_clock", "bwtc_hip_wavelet_host_progress", "bwtc_hip_wavelet_latency", "bwtc_hip_host_staging_bytes", "bwtc_hip_host_usable_cpus", "bwtc_hip_n_lf", "bwtc_hip_bwt",
    "bwtc_hip_bwt_block", "bwtc_hip_bwt_block_device", "bwtc_hip_inverse_bwt_block",
    "bwtc_hip_inverse_bwt_block_device", "bwtc_hip_compress_bound",
    "bwtc_hip_huffman_encode_device", "bwtc_hip_huffman_encode", "bwtc_hip_transform_and_encode",
    "bwtc_hip_wavelet_section_stats", "bwtc_hip_transform_and_encode_wavelet", "bwtc_hip_wavelet_encode",
    "bwtc_hip_wavelet_encode_device", "bwtc_hip_wavelet_encode_device_begin", "bwtc_hip_wavelet_encode_end",
    "bwtc_hip_wavelet_encode_device_prepare", "bwtc_hip_wavelet_encode_queue",
    "bwtc_hip_wavelet_depth", "bwtc_hip_wavelet_set_depth", "bwtc_hip_numa_node", "bwtc_hip_host_cpu_slice", "bwtc_hip_set_worker_cpus", "bwtc_hip_wavelet_reset", "bwtc_hip_wavelet_start", "bwtc_hip_host_wavelet_sections", "bwtc_hip_host_wavelet_streams", "bwtc_hip_host_wavelet_streams_lanes", "bwtc_hip_host_huffman_lengths", "bwtc_hip_host_huffman_codes", "bwtc_hip_host_serialize_shape",
    "bwtc_hip_host_sections", "bwtc_hip_host_bwtblock_header", "bwtc_hip_synth", "bwtc_hip_suffix_array",
    "bwtc_hip_test_sort_u32", "bwtc_hip_test_sort_u64", "bwtc_hip_test_scan_u32",
    "bwtc_hip_wavelet_depth_needed", "bwtc_hip_grammar_create", "bwtc_hip_grammar_destroy", "bwtc_hip_grammar_rules", "bwtc_hip_grammar_special_symbols",
    "bwtc_hip_grammar_is_special", "bwtc_hip_grammar_write", "bwtc_hip_grammar_read", "bwtc_hip_pair_replace_device",
    "bwtc_hip_precompress", "bwtc_hip_host_precompress", "bwtc_hip_postprocess",
]

_lib = None


def load():
    """Load libbwtc_hip.so; raises BwtcHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BwtcHipError(
            "%s not found: build it with `make -C bwtc_amd/csrc` (or __graft_entry__.build()); "
            "there is no CPU fallback" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.bwtc_hip_device_count.restype = ctypes.c_int
    L.bwtc_hip_version.restype = ctypes.c_char_p
    L.bwtc_hip_workspace_bytes.restype = _u64
    L.bwtc_hip_workspace_bytes.argtypes = [_u32]
    L.bwtc_hip_create.argtypes = [ctypes.c_int, _u32, ctypes.POINTER(_vp)]
    L.bwtc_hip_destroy.argtypes = [_vp]
    L.bwtc_hip_destroy.restype = None
    L.bwtc_hip_stream.restype = _vp
    L.bwtc_hip_stream.argtypes = [_vp]
    L.bwtc_hip_get_stats.argtypes = [_vp, ctypes.POINTER(Stats)]
    L.bwtc_hip_set_profiling.argtypes = [_vp, ctypes.c_int]
    L.bwtc_hip_get_kernel_timers.argtypes = [_vp, ctypes.POINTER(KernelTimers), ctypes.c_int]
    L.bwtc_hip_test_gpu_lanes.argtypes = [_vp, _vp, _u64, _vp, _u32, ctypes.c_int, _vp, _u64, _vp]
    L.bwtc_hip_copy_probe.argtypes = [_vp, _u64, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
    L.bwtc_hip_malloc.restype = _vp
    L.bwtc_hip_malloc.argtypes = [_vp, _u64]
    L.bwtc_hip_free.restype = None
    L.bwtc_hip_free.argtypes = [_vp, _vp]
    L.bwtc_hip_memcpy_to_device.argtypes = [_vp, _vp, _vp, _u64]
    L.bwtc_hip_memcpy_to_host.argtypes = [_vp, _vp, _vp, _u64]
    L.bwtc_hip_host_alloc.restype = _vp
    L.bwtc_hip_host_alloc.argtypes = [_vp, _u64]
    L.bwtc_hip_host_free.restype = None
    L.bwtc_hip_host_free.argtypes = [_vp, _vp]
    L.bwtc_hip_memcpy_to_device_async.argtypes = [_vp, _vp, _vp, _u64]
    L.bwtc_hip_copy_wait.argtypes = [_vp]
    L.bwtc_hip_wavelet_host_clock.argtypes = [_vp, _vp, _vp, _vp]
    L.bwtc_hip_wavelet_host_progress.argtypes = [_vp, _vp, _vp]
    L.bwtc_hip_wavelet_latency.argtypes = [_vp, _vp]
    L.bwtc_hip_host_staging_bytes.argtypes = [_vp, _vp]
    L.bwtc_hip_numa_node.argtypes = [_vp]
    L.bwtc_hip_host_cpu_slice.argtypes = [ctypes.c_int, _u32, _u32, _vp, _u32]
    L.bwtc_hip_set_worker_cpus.argtypes = [_vp, _vp, _u32]
    L.bwtc_hip_synth.argtypes = [ctypes.c_char, _u64, _u64, _vp]
    L.bwtc_hip_n_lf.restype = _u32
    L.bwtc_hip_n_lf.argtypes = [_u32, _u32]
    L.bwtc_hip_bwt.argtypes = [_vp, _vp, _u32, _vp, _u32, _vp]
    L.bwtc_hip_bwt_block.argtypes = [_vp, _vp, _u32, _vp, _u32, _vp]
    L.bwtc_hip_bwt_block_device.argtypes = [_vp, _vp, _vp, _u32, _vp, _u32, _vp]
    L.bwtc_hip_inverse_bwt_block.argtypes = [_vp, _vp, _u32, _vp, _u32]
    L.bwtc_hip_inverse_bwt_block_device.argtypes = [_vp, _vp, _vp, _u32, _vp, _u32]
    L.bwtc_hip_compress_bound.restype = _u64
    L.bwtc_hip_compress_bound.argtypes = [_u32]
    L.bwtc_hip_huffman_encode_device.argtypes = [_vp, _vp, _u32, _vp, _u32, _vp, _vp, _u64,
                                                 ctypes.POINTER(_u64)]
    L.bwtc_hip_huffman_encode.argtypes = [_vp, _vp, _u32, _vp, _u32, _vp, _vp, _u64,
                                          ctypes.POINTER(_u64)]
    L.bwtc_hip_transform_and_encode.argtypes = [_vp, _vp, _u32, _u32, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_wavelet_section_stats.argtypes = [_vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32]
    L.bwtc_hip_transform_and_encode_wavelet.argtypes = [_vp, _vp, _u32, _u32, _u32, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_wavelet_encode.argtypes = [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_wavelet_encode_device.argtypes = [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_wavelet_encode_device_begin.argtypes = [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_wavelet_encode_end.argtypes = [_vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_wavelet_encode_device_prepare.argtypes = [_vp, _vp, _u32, _vp, _u32, _vp, _u32, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_wavelet_encode_queue.argtypes = [_vp, _u64, _u32, ctypes.POINTER(_u32)]
    L.bwtc_hip_wavelet_depth.restype = ctypes.c_uint32
    L.bwtc_hip_wavelet_depth.argtypes = [_vp]
    L.bwtc_hip_wavelet_set_depth.argtypes = [_vp, ctypes.c_uint32]
    L.bwtc_hip_wavelet_depth_needed.restype = ctypes.c_uint32
    L.bwtc_hip_wavelet_depth_needed.argtypes = [_vp]
    L.bwtc_hip_wavelet_reset.restype = None
    L.bwtc_hip_wavelet_reset.argtypes = [_vp]
    L.bwtc_hip_wavelet_start.argtypes = [_vp, ctypes.c_char]
    L.bwtc_hip_host_wavelet_sections.argtypes = [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, ctypes.c_char, _vp, _vp, _u64,
                                                 ctypes.POINTER(_u64)]
    L.bwtc_hip_host_wavelet_streams.argtypes = L.bwtc_hip_host_wavelet_sections.argtypes
    L.bwtc_hip_host_wavelet_streams_lanes.argtypes = L.bwtc_hip_host_wavelet_sections.argtypes
    # pair-replacing pre-stage (`--prepr`)
    L.bwtc_hip_grammar_create.restype = _vp
    L.bwtc_hip_grammar_create.argtypes = []
    L.bwtc_hip_grammar_destroy.restype = None
    L.bwtc_hip_grammar_destroy.argtypes = [_vp]
    L.bwtc_hip_grammar_rules.restype = _u32
    L.bwtc_hip_grammar_rules.argtypes = [_vp]
    L.bwtc_hip_grammar_special_symbols.restype = _u32
    L.bwtc_hip_grammar_special_symbols.argtypes = [_vp]
    L.bwtc_hip_grammar_is_special.argtypes = [_vp, ctypes.c_uint]
    L.bwtc_hip_grammar_write.argtypes = [_vp, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_grammar_read.argtypes = [_vp, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_pair_replace_device.argtypes = [_vp, _vp, _vp, _u64, _vp, ctypes.POINTER(_u64), ctypes.POINTER(_u32)]
    L.bwtc_hip_precompress.argtypes = [_vp, _vp, ctypes.c_char_p, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_host_precompress.argtypes = [_vp, ctypes.c_char_p, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_postprocess.argtypes = [_vp, _vp, _u64, _vp, _u64, ctypes.POINTER(_u64)]
    L.bwtc_hip_host_huffman_lengths.restype = None
    L.bwtc_hip_host_huffman_lengths.argtypes = [_vp, _vp]
    L.bwtc_hip_host_huffman_codes.restype = None
    L.bwtc_hip_host_huffman_codes.argtypes = [_vp, _vp]
    L.bwtc_hip_host_serialize_shape.restype = _u32
    L.bwtc_hip_host_serialize_shape.argtypes = [_vp, _vp, _u32]
    L.bwtc_hip_host_sections.restype = _u32
    L.bwtc_hip_host_sections.argtypes = [_vp, _vp]
    L.bwtc_hip_host_bwtblock_header.restype = _u32
    L.bwtc_hip_host_bwtblock_header.argtypes = [_vp, _u32, _vp, _u32]
    L.bwtc_hip_suffix_array.argtypes = [_vp, _vp, _u32, _vp]
    L.bwtc_hip_test_sort_u32.argtypes = [_vp, _vp, _vp, _u64, ctypes.c_int]
    L.bwtc_hip_test_sort_u64.argtypes = [_vp, _vp, _vp, _u64, ctypes.c_int]
    L.bwtc_hip_test_scan_u32.argtypes = [_vp, _vp, _u64]
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        raise BwtcHipError("%s failed with code %d" % (what, rc))


def _ptr(a):
    return a.ctypes.data_as(_vp)


class Grammar:
    """bwtc::Grammar of one precompressor block (bwtc_hip_grammar): what the `--prepr` rounds did to it."""

    def __init__(self):
        self.lib = load()
        self.h = self.lib.bwtc_hip_grammar_create()
        if not self.h:
            raise BwtcHipError("bwtc_hip_grammar_create failed")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.bwtc_hip_grammar_destroy(self.h)
            self.h = None

    @property
    def rules(self):
        return int(self.lib.bwtc_hip_grammar_rules(self.h))

    @property
    def special_symbols(self):
        return int(self.lib.bwtc_hip_grammar_special_symbols(self.h))

    def is_special(self, c):
        return bool(self.lib.bwtc_hip_grammar_is_special(self.h, int(c)))

    def write(self):
        out = np.zeros((1 << 17) + 8 * self.rules, np.uint8)      # a rule costs at most seven bytes, the freed symbols' table at most 64 KiB
        n = _u64(0)
        _check(self.lib.bwtc_hip_grammar_write(self.h, _ptr(out), out.size, ctypes.byref(n)), "bwtc_hip_grammar_write")
        return out[:n.value].copy()

    def read(self, raw):
        raw = np.ascontiguousarray(raw, np.uint8)
        n = _u64(0)
        _check(self.lib.bwtc_hip_grammar_read(self.h, _ptr(raw), raw.size, ctypes.byref(n)), "bwtc_hip_grammar_read")
        return int(n.value)

    def host_precompress(self, options, data):
        """Precompressor::precompress with both sweeps on the calling thread (no device)."""
        buf = np.ascontiguousarray(data, np.uint8).copy()
        n = _u64(0)
        _check(self.lib.bwtc_hip_host_precompress(self.h, options.encode(), _ptr(buf), buf.size, ctypes.byref(n)), "bwtc_hip_host_precompress")
        return buf[:n.value].copy()

    def postprocess(self, data, max_size):
        data = np.ascontiguousarray(data, np.uint8)
        out = np.zeros(max(1, max_size), np.uint8)
        n = _u64(0)
        _check(self.lib.bwtc_hip_postprocess(self.h, _ptr(data), data.size, _ptr(out), out.size, ctypes.byref(n)), "bwtc_hip_postprocess")
        return out[:n.value].copy()


class Context:
    """One GPU, one stream, one persistent workspace (bwtc_hip_ctx)."""

    def __init__(self, device=0, max_block_size=1 << 20):
        self.lib = load()
        if self.lib.bwtc_hip_device_count() <= 0:
            raise BwtcHipError("no HIP device visible")
        h = _vp()
        _check(self.lib.bwtc_hip_create(device, max_block_size, ctypes.byref(h)), "bwtc_hip_create")
        self.handle = h
        self.max_block_size = max_block_size

    def close(self):
        if getattr(self, "handle", None):
            self.lib.bwtc_hip_destroy(self.handle)
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def n_lf(self, size, starting_points):
        return int(self.lib.bwtc_hip_n_lf(size, starting_points))

    def precompress(self, grammar, options, data):
        """Precompressor::precompress on the GPU (bwtc_hip_precompress): the precompressed block."""
        buf = np.ascontiguousarray(data, np.uint8).copy()
        n = _u64(0)
        _check(self.lib.bwtc_hip_precompress(self.handle, grammar.h, options.encode(), _ptr(buf), buf.size, ctypes.byref(n)), "bwtc_hip_precompress")
        return buf[:n.value].copy()

    def stats(self):
        s = Stats()
        _check(self.lib.bwtc_hip_get_stats(self.handle, ctypes.byref(s)), "bwtc_hip_get_stats")
        return s

    def reset_kernel_timers(self):
        """Turns per-kernel event timing on and clears the accumulators."""
        _check(self.lib.bwtc_hip_set_profiling(self.handle, 1), "bwtc_hip_set_profiling")
        _check(self.lib.bwtc_hip_get_kernel_timers(self.handle, None, 1), "bwtc_hip_get_kernel_timers")

    def kernel_timers(self):
        k = KernelTimers()
        _check(self.lib.bwtc_hip_get_kernel_timers(self.handle, ctypes.byref(k), 0),
               "bwtc_hip_get_kernel_timers")
        return {"scatter_launches": int(k.scatter_launches), "scatter_bytes": int(k.scatter_bytes),
                "scatter_ms": float(k.scatter_ms)}

    def test_gpu_lanes(self, w, bounds, mode=0):
        """Chains of w-elements through the GPU lane engine (mode 0) or the host's scalar coder (mode 1): list of byte strings."""
        w = np.ascontiguousarray(w, np.uint16)
        b = np.ascontiguousarray(bounds, np.uint64)
        k = b.size - 1
        out = np.zeros(int(w.size) * 4 + 16 * k + 64, np.uint8)
        off = np.zeros(k + 1, np.uint64)
        _check(self.lib.bwtc_hip_test_gpu_lanes(self.handle, _ptr(w), w.size, _ptr(b), k, mode, _ptr(out), out.size, _ptr(off)),
               "bwtc_hip_test_gpu_lanes")
        return [out[int(off[j]):int(off[j + 1])].tobytes() for j in range(k)]

    def copy_probe(self, nbytes=1 << 30, reps=5):
        """GB/s (read + written) of the library's own 16-byte-per-lane copy kernel on this GPU (bwtc_hip_copy_probe)."""
        g = ctypes.c_double(0.0)
        _check(self.lib.bwtc_hip_copy_probe(self.handle, nbytes, reps, ctypes.byref(g)), "bwtc_hip_copy_probe")
        return float(g.value)

    # ---- device buffers without a HIP binding of one's own ------------------------------
    def dmalloc(self, nbytes):
        p = self.lib.bwtc_hip_malloc(self.handle, nbytes)
        if not p:
            raise BwtcHipError("bwtc_hip_malloc(%d) failed" % nbytes)
        return p

    def dfree(self, ptr):
        self.lib.bwtc_hip_free(self.handle, _vp(ptr))

    def to_device(self, d_ptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(self.lib.bwtc_hip_memcpy_to_device(self.handle, _vp(d_ptr), _ptr(arr), arr.nbytes),
               "bwtc_hip_memcpy_to_device")

    def host_alloc(self, nbytes):
        """Page-locked host bytes as a numpy array (freed with host_free, or with the process)."""
        p = self.lib.bwtc_hip_host_alloc(self.handle, nbytes)
        if not p:
            raise BwtcHipError("bwtc_hip_host_alloc(%d) failed" % nbytes)
        arr = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p
        return arr

    def host_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p:
            self.lib.bwtc_hip_host_free(self.handle, _vp(p))

    def to_device_async(self, d_ptr, arr):
        """Upload on the context's copy stream; returns at once (arr must stay alive and unchanged
        until copy_wait())."""
        _check(self.lib.bwtc_hip_memcpy_to_device_async(self.handle, _vp(d_ptr), _ptr(arr), arr.nbytes),
               "bwtc_hip_memcpy_to_device_async")

    def copy_wait(self):
        _check(self.lib.bwtc_hip_copy_wait(self.handle), "bwtc_hip_copy_wait")

    def wavelet_host_clock(self):
        """(seconds in the adaptive models, seconds in the range coders, blocks) of the 'B' coder's
        worker threads, summed over threads since the context was made."""
        m, c, b = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_uint64(0)
        _check(self.lib.bwtc_hip_wavelet_host_clock(self.handle, ctypes.byref(m), ctypes.byref(c), ctypes.byref(b)),
               "bwtc_hip_wavelet_host_clock")
        return m.value, c.value, b.value

    def wavelet_set_depth(self, depth):
        """Blocks that may be under way from now on; staging buffers beyond that go back to the system."""
        _check(self.lib.bwtc_hip_wavelet_set_depth(self.handle, int(depth)), "bwtc_hip_wavelet_set_depth")

    def wavelet_depth_needed(self):
        """Blocks to keep under way for the rate shown so far (0 until four blocks have finished)."""
        return int(self.lib.bwtc_hip_wavelet_depth_needed(self.handle))

    def wavelet_latency(self):
        """Mean seconds a block has been under way (device half started -> record finished)."""
        v = ctypes.c_double(0)
        _check(self.lib.bwtc_hip_wavelet_latency(self.handle, ctypes.byref(v)), "bwtc_hip_wavelet_latency")
        return v.value

    def numa_node(self):
        """NUMA node of the context's GPU, -1 when the system does not say."""
        return int(self.lib.bwtc_hip_numa_node(self.handle))

    def set_worker_cpus(self, cpus):
        """The context's worker threads may run on exactly these CPUs (empty: no restriction)."""
        a = np.ascontiguousarray(cpus, np.uint32)
        _check(self.lib.bwtc_hip_set_worker_cpus(self.handle, _ptr(a) if a.size else None, a.size), "bwtc_hip_set_worker_cpus")

    def wavelet_host_progress(self):
        """(blocks that joined the host half, blocks whose record the workers have finished)."""
        q, f = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self.lib.bwtc_hip_wavelet_host_progress(self.handle, ctypes.byref(q), ctypes.byref(f)),
               "bwtc_hip_wavelet_host_progress")
        return q.value, f.value

    def to_host(self, d_ptr, nbytes):
        out = np.empty(nbytes, np.uint8)
        _check(self.lib.bwtc_hip_memcpy_to_host(self.handle, _ptr(out), _vp(d_ptr), nbytes),
               "bwtc_hip_memcpy_to_host")
        return out

    def bwt_block(self, data, starting_points=8):
        """BWTManager::doTransform(block, freqs): returns (bwt bytes, LFpowers, freqs).

        The buffer handed to the library carries one extra byte after the block, as the
        reference's PrecompressorBlock does, and that byte is checked to be untouched."""
        data = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8)
                                    if not isinstance(data, np.ndarray) else data, dtype=np.uint8)
        size = data.size
        buf = np.empty(size + 1, np.uint8)
        buf[:size] = data
        buf[size] = 0xA5
        n_lf = self.n_lf(size, starting_points)
        lf = np.zeros(n_lf, np.uint32)
        freqs = np.zeros(256, np.uint32)
        _check(self.lib.bwtc_hip_bwt_block(self.handle, _ptr(buf), size, _ptr(lf), n_lf,
                                           _ptr(freqs)), "bwtc_hip_bwt_block")
        if buf[size] != 0xA5:
            raise BwtcHipError("byte after the block was modified")
        return buf[:size].copy(), lf, freqs

    def bwt_raw(self, T, n_lf=1, freqs=None):
        """BWTransform::doTransform(begin, length, LF, freqs) on T (sentinel included)."""
        T = np.array(T, dtype=np.uint8, copy=True)
        lf = np.zeros(n_lf, np.uint32)
        fr = np.zeros(256, np.uint32) if freqs is None else freqs
        _check(self.lib.bwtc_hip_bwt(self.handle, _ptr(T), T.size, _ptr(lf), n_lf, _ptr(fr)),
               "bwtc_hip_bwt")
        return T, lf, fr

    def bwt_block_device(self, d_in_ptr, d_out_ptr, size, starting_points=8):
        n_lf = self.n_lf(size, starting_points)
        lf = np.zeros(n_lf, np.uint32)
        freqs = np.zeros(256, np.uint32)
        _check(self.lib.bwtc_hip_bwt_block_device(self.handle, _vp(d_in_ptr), _vp(d_out_ptr), size,
                                                  _ptr(lf), n_lf, _ptr(freqs)),
               "bwtc_hip_bwt_block_device")
        return lf, freqs

    def inverse_bwt_block(self, bwt, lf):
        """InverseBWTransform::doTransform(BWTBlock&): returns the original block; raises
        BwtcHipError (code -4) when an LF power does not lie on the LF walk."""
        b = np.array(bwt, dtype=np.uint8, copy=True)
        lf = np.ascontiguousarray(lf, dtype=np.uint32)
        _check(self.lib.bwtc_hip_inverse_bwt_block(self.handle, _ptr(b), b.size, _ptr(lf), lf.size),
               "bwtc_hip_inverse_bwt_block")
        return b

    def inverse_bwt_block_device(self, d_bwt_ptr, d_out_ptr, size, lf):
        lf = np.ascontiguousarray(lf, dtype=np.uint32)
        _check(self.lib.bwtc_hip_inverse_bwt_block_device(self.handle, _vp(d_bwt_ptr), _vp(d_out_ptr),
                                                          size, _ptr(lf), lf.size),
               "bwtc_hip_inverse_bwt_block_device")

    def wavelet_section_stats(self, bwt, freqs):
        """utils::calculateRunsAndCharacters per section (WaveletTree ctor front-end): returns
        (section lengths, run_freqs[section][256], total runs, [dict(length -> count)])."""
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
        nsec = _u32(0)
        sec = np.zeros(256, np.uint32)
        rf = np.zeros(256 * 256, np.uint32)
        tot = np.zeros(256, np.uint64)
        off = np.zeros(257, np.uint32)
        cap = max(1 << 16, bwt.size // 8 + 4096)
        dl = np.zeros(cap, np.uint32)
        dc = np.zeros(cap, np.uint32)
        _check(self.lib.bwtc_hip_wavelet_section_stats(self.handle, _ptr(bwt), bwt.size, _ptr(freqs),
                                                       ctypes.byref(nsec), _ptr(sec), _ptr(rf), _ptr(tot),
                                                       _ptr(off), _ptr(dl), _ptr(dc), cap),
               "bwtc_hip_wavelet_section_stats")
        n = nsec.value
        dist = [dict(zip(dl[off[s]:off[s + 1]].tolist(), dc[off[s]:off[s + 1]].tolist())) for s in range(n)]
        return sec[:n].copy(), rf.reshape(256, 256)[:n].copy(), tot[:n].copy(), dist

    def wavelet_reset(self):
        """Start a new stream: the 'B' coder's carried model state goes back to its initial value."""
        self.lib.bwtc_hip_wavelet_reset(self.handle)

    def wavelet_encode(self, bwt, lf, freqs, threads=0):
        """WaveletEncoder ('B'): writeBlockHeader + encodeData + finishBlock on a transformed block."""
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        lf = np.ascontiguousarray(lf, dtype=np.uint32)
        freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
        cap = self.compress_bound(bwt.size)
        out = np.zeros(cap, np.uint8)
        n = _u64(0)
        _check(self.lib.bwtc_hip_wavelet_encode(self.handle, _ptr(bwt), bwt.size, _ptr(lf), lf.size,
                                                _ptr(freqs), threads, _ptr(out), cap, ctypes.byref(n)),
               "bwtc_hip_wavelet_encode")
        return out[:n.value].copy()

    def wavelet_encode_device(self, d_bwt_ptr, size, lf, freqs, out, threads=0):
        """Same on a device-resident block; `out` is a host uint8 array; returns bytes written."""
        lf = np.ascontiguousarray(lf, dtype=np.uint32)
        freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
        n = _u64(0)
        _check(self.lib.bwtc_hip_wavelet_encode_device(self.handle, _vp(d_bwt_ptr), size, _ptr(lf), lf.size,
                                                       _ptr(freqs), threads, _ptr(out), out.size,
                                                       ctypes.byref(n)),
               "bwtc_hip_wavelet_encode_device")
        return int(n.value)

    def wavelet_start(self, coder="B"):
        """New wavelet stream with the main model of coder letter `coder` ('B', 'b' or 'u')."""
        _check(self.lib.bwtc_hip_wavelet_start(self.handle, coder.encode()), "bwtc_hip_wavelet_start")

    def wavelet_encode_device_begin(self, d_bwt_ptr, size, lf, freqs, out, threads=0):
        """First half: device work now, models + range coder queued on the worker threads.
        `out` must stay alive until wavelet_encode_end(ticket); returns the ticket."""
        lf = np.ascontiguousarray(lf, dtype=np.uint32)
        freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
        t = _u64(0)
        _check(self.lib.bwtc_hip_wavelet_encode_device_begin(self.handle, _vp(d_bwt_ptr), size, _ptr(lf), lf.size,
                                                             _ptr(freqs), threads, _ptr(out), out.size,
                                                             ctypes.byref(t)),
               "bwtc_hip_wavelet_encode_device_begin")
        return int(t.value)

    def wavelet_encode_device_prepare(self, d_bwt_ptr, size, lf, freqs, out, threads=0):
        """The half of _begin that needs nothing from earlier blocks (all the device work)."""
        lf = np.ascontiguousarray(lf, dtype=np.uint32)
        freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
        t = _u64(0)
        _check(self.lib.bwtc_hip_wavelet_encode_device_prepare(self.handle, _vp(d_bwt_ptr), size, _ptr(lf), lf.size,
                                                               _ptr(freqs), threads, _ptr(out), out.size,
                                                               ctypes.byref(t)),
               "bwtc_hip_wavelet_encode_device_prepare")
        return int(t.value)

    def wavelet_encode_queue(self, ticket, state_in):
        """Gives a prepared block its place in a stream; returns the carried state after it."""
        st = _u32(0)
        _check(self.lib.bwtc_hip_wavelet_encode_queue(self.handle, ticket, state_in, ctypes.byref(st)),
               "bwtc_hip_wavelet_encode_queue")
        return int(st.value)

    def wavelet_encode_end(self, ticket):
        """Second half: waits for the block, returns the record's size."""
        n = _u64(0)
        _check(self.lib.bwtc_hip_wavelet_encode_end(self.handle, _u64(ticket), ctypes.byref(n)),
               "bwtc_hip_wavelet_encode_end")
        return int(n.value)

    def transform_and_encode_wavelet(self, data, starting_points=8, threads=0):
        """WaveletEncoder::transformAndEncode ('B'): returns (record, bwt bytes)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        buf = np.empty(data.size + 1, np.uint8)
        buf[:data.size] = data
        buf[data.size] = 0x5A
        cap = self.compress_bound(data.size)
        out = np.zeros(cap, np.uint8)
        n = _u64(0)
        _check(self.lib.bwtc_hip_transform_and_encode_wavelet(self.handle, _ptr(buf), data.size,
                                                              starting_points, threads, _ptr(out), cap,
                                                              ctypes.byref(n)),
               "bwtc_hip_transform_and_encode_wavelet")
        if buf[data.size] != 0x5A:
            raise BwtcHipError("byte after the block was modified")
        return out[:n.value].copy(), buf[:data.size].copy()

    def compress_bound(self, size):
        return int(self.lib.bwtc_hip_compress_bound(size))

    def huffman_encode_device(self, d_bwt_ptr, size, lf, freqs, d_out_ptr, out_cap=None):
        """HuffmanEncoder: writeBlockHeader + encodeData + finishBlock on a device-resident
        transformed block; returns the number of bytes written at d_out_ptr."""
        lf = np.ascontiguousarray(lf, dtype=np.uint32)
        freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
        n = _u64(0)
        cap = self.compress_bound(size) if out_cap is None else out_cap
        _check(self.lib.bwtc_hip_huffman_encode_device(self.handle, _vp(d_bwt_ptr), size, _ptr(lf),
                                                       lf.size, _ptr(freqs), _vp(d_out_ptr), cap,
                                                       ctypes.byref(n)),
               "bwtc_hip_huffman_encode_device")
        return int(n.value)

    def huffman_encode(self, bwt, lf, freqs):
        bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
        lf = np.ascontiguousarray(lf, dtype=np.uint32)
        freqs = np.ascontiguousarray(freqs, dtype=np.uint32)
        cap = self.compress_bound(bwt.size)
        out = np.zeros(cap, np.uint8)
        n = _u64(0)
        _check(self.lib.bwtc_hip_huffman_encode(self.handle, _ptr(bwt), bwt.size, _ptr(lf), lf.size,
                                                _ptr(freqs), _ptr(out), cap, ctypes.byref(n)),
               "bwtc_hip_huffman_encode")
        return out[:n.value].copy()

    def transform_and_encode(self, data, starting_points=8):
        """HuffmanEncoder::transformAndEncode: returns (encoded BWT-block record, bwt bytes)."""
        data = np.ascontiguousarray(data, dtype=np.uint8)
        buf = np.empty(data.size + 1, np.uint8)
        buf[:data.size] = data
        buf[data.size] = 0x5A
        cap = self.compress_bound(data.size)
        out = np.zeros(cap, np.uint8)
        n = _u64(0)
        _check(self.lib.bwtc_hip_transform_and_encode(self.handle, _ptr(buf), data.size,
                                                      starting_points, _ptr(out), cap,
                                                      ctypes.byref(n)),
               "bwtc_hip_transform_and_encode")
        if buf[data.size] != 0x5A:
            raise BwtcHipError("byte after the block was modified")
        return out[:n.value].copy(), buf[:data.size].copy()

    def suffix_array(self, T):
        T = np.ascontiguousarray(T, dtype=np.uint8)
        sa = np.zeros(T.size, np.uint32)
        _check(self.lib.bwtc_hip_suffix_array(self.handle, _ptr(T), T.size, _ptr(sa)),
               "bwtc_hip_suffix_array")
        return sa

    def test_sort(self, keys, vals, nbits):
        keys = np.array(keys, copy=True)
        vals = np.array(vals, dtype=np.uint32, copy=True)
        fn = self.lib.bwtc_hip_test_sort_u64 if keys.dtype == np.uint64 else self.lib.bwtc_hip_test_sort_u32
        _check(fn(self.handle, _ptr(keys), _ptr(vals), keys.size, nbits), "bwtc_hip_test_sort")
        return keys, vals

    def test_scan(self, data):
        data = np.array(data, dtype=np.uint32, copy=True)
        _check(self.lib.bwtc_hip_test_scan_u32(self.handle, _ptr(data), data.size), "bwtc_hip_test_scan")
        return data


def host_cpu_slice(numa_node, rank, ranks):
    """CPUs for the workers of context `rank` of `ranks` contexts that share NUMA node `numa_node`
    (-1: no node restriction): a contiguous slice of what this process may use.  Host only."""
    L = load()
    buf = np.zeros(4096, np.uint32)
    n = L.bwtc_hip_host_cpu_slice(int(numa_node), int(rank), int(ranks), _ptr(buf), buf.size)
    if n < 0:
        raise BwtcHipError("bwtc_hip_host_cpu_slice failed with code %d" % n)
    return [int(c) for c in buf[:n]]


def host_staging_bytes():
    """(now, peak) bytes of host staging memory this process holds for blocks under way."""
    L = load()
    a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
    L.bwtc_hip_host_staging_bytes(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def synth_into(kind, seed, out):
    """Fills the uint8 array `out` with the synthetic block of `kind` ('r', 'd', 't') and `seed`:
    the library's C++ generator, the same bytes as bwtc_amd.synth (tests/test_abi.py)."""
    _check(load().bwtc_hip_synth(kind.encode(), seed, out.size, _ptr(out)), "bwtc_hip_synth")
    return out
