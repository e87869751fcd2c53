"""Deterministic synthetic blocks for the BASELINE.json configs.

splitmix64, integer only, so the same bytes can be regenerated anywhere (tests, bench,
the C++ host tools).  C1 = random bytes, C2 = uniform ACGT, C3 = Zipf-ish token text.
"""
import numpy as np

_M64 = (1 << 64) - 1

TOKENS = [
    "the ", "of ", "and ", "in ", "to ", "a ", "is ", "that ", "for ", "it ", "as ", "was ",
    "with ", "be ", "by ", "on ", "not ", "he ", "this ", "are ", "or ", "his ", "from ", "at ",
    "which ", "but ", "have ", "an ", "had ", "they ", "you ", "were ", "their ", "one ", "all ",
    "we ", "can ", "her ", "has ", "there ", "been ", "if ", "more ", "when ", "will ", "would ",
    "who ", "so ", "no ", "[[", "]]", "&quot;", "<page>\n", "</page>\n", "==", "'''",
    "[[Category:", "<title>", "</title>\n", "<text>", "</text>\n", "http://", ".\n", ", ",
]
assert len(TOKENS) == 64


def splitmix64_stream(seed, count, start=0):
    """Outputs start+1 .. start+count of splitmix64 seeded with `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        k = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        x = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def gen_random_bytes(size, seed=1):
    """C1: little-endian bytes of successive splitmix64 outputs."""
    z = splitmix64_stream(seed, (size + 7) // 8)
    return z.view(np.uint8)[:size].copy()


def gen_dna(size, seed=2):
    """C2: symbol j of an output word is "ACGT"[(z >> 2j) & 3], j = 0..31."""
    z = splitmix64_stream(seed, (size + 31) // 32)
    sh = (np.arange(32, dtype=np.uint64) * np.uint64(2))[None, :]
    idx = ((z[:, None] >> sh) & np.uint64(3)).astype(np.uint8).reshape(-1)[:size]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[idx]


def gen_text(size, seed=3):
    """C3: Zipf-ish token text, four splitmix64 draws per token."""
    tok = [np.frombuffer(t.encode(), dtype=np.uint8) for t in TOKENS]
    tok_len = np.array([t.size for t in tok], dtype=np.int64)
    tok_cat = np.concatenate(tok)
    tok_off = np.concatenate([[0], np.cumsum(tok_len)[:-1]])
    out = np.empty(size + 16, np.uint8)
    pos = 0
    drawn = 0
    chunk = max(1024, min(1 << 20, size // 2 + 16))
    while pos < size:
        z = splitmix64_stream(seed, 4 * chunk, 4 * drawn).reshape(-1, 4)
        drawn += chunk
        k = ((z[:, 0] % np.uint64(64)) * (z[:, 1] % np.uint64(64)) // np.uint64(64)).astype(np.int64)
        extra = (z[:, 2] & np.uint64(31)) == 0
        letter = (np.uint64(ord("a")) + z[:, 3] % np.uint64(26)).astype(np.uint8)
        lens = tok_len[k] + extra
        ends = np.cumsum(lens)
        starts = ends - lens
        total = int(ends[-1])
        buf = np.empty(total, np.uint8)
        # token bytes
        rep_tok = np.repeat(np.arange(k.size), tok_len[k])
        within = np.arange(rep_tok.size) - np.repeat(np.cumsum(tok_len[k]) - tok_len[k], tok_len[k])
        buf[np.repeat(starts, tok_len[k]) + within] = tok_cat[tok_off[k][rep_tok] + within]
        buf[(ends - 1)[extra]] = letter[extra]
        take = min(total, size - pos)
        out[pos:pos + take] = buf[:take]
        pos += take
    return out[:size].copy()
