"""Integer-only synthetic byte streams.

Every stream is a pure function of (kind, seed, absolute byte offset), built on
the counter-based splitmix64 sequence, so any shard of a stream can be produced
independently on any rank / GPU and the concatenation equals the whole.

    uniform(seed, n)  -- little-endian splitmix64 words, truncated to n bytes
    zipf(seed, n)     -- 256 symbols, integer weights floor(2**24 / rank)
    text(seed, n)     -- 96 printable symbols, same weights, RANK order below

The stand-in for the reference's missing data/random_64m.dat
(/root/reference/.MISSING_LARGE_BLOBS:1, README.md:14) is uniform(42, 64 MiB).
"""
from __future__ import annotations

import numpy as np

_U = np.uint64
GOLDEN = 0x9E3779B97F4A7C15

RANK = (" etaoinshrdlcumwfgypbvkjxqz\n.,ETAOINSHRDLCUMWFGYPBVKJXQZ0123456789-'\"()/:;=_<>[]{}!?#$%&*+@\\^`|~")
assert len(RANK) == 96 and len(set(RANK)) == 96
ZIPF256 = [(r * 167 + 13) & 255 for r in range(256)]
KINDS = ("uniform", "zipf", "text")


def splitmix64(seed: int, first_word: int, nwords: int) -> np.ndarray:
    """Words first_word .. first_word+nwords-1 (1-based counter k) of the stream."""
    with np.errstate(over="ignore"):
        k = np.arange(first_word, first_word + nwords, dtype=_U)
        z = _U(seed & 0xFFFFFFFFFFFFFFFF) + k * _U(GOLDEN)
        z = (z ^ (z >> _U(30))) * _U(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> _U(27))) * _U(0x94D049BB133111EB)
        return z ^ (z >> _U(31))


def uniform(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """Bytes [offset, offset+n) of the uniform stream."""
    w0 = offset // 8
    w1 = (offset + n + 7) // 8
    raw = splitmix64(seed, w0 + 1, w1 - w0).astype("<u8").view(np.uint8)
    lo = offset - w0 * 8
    return raw[lo:lo + n].copy()


def _u32stream(seed: int, n: int, offset: int) -> np.ndarray:
    """32-bit draws [offset, offset+n): low half of each word first, then high."""
    w0 = offset // 2
    w1 = (offset + n + 1) // 2
    raw = splitmix64(seed, w0 + 1, w1 - w0).astype("<u8").view("<u4")
    lo = offset - w0 * 2
    return raw[lo:lo + n].astype(_U)


def zipf_table(K: int):
    cum = np.cumsum(np.array([(1 << 24) // r for r in range(1, K + 1)], dtype=_U))
    return cum, int(cum[-1])


def _zipf(seed: int, n: int, offset: int, K: int, sym_of_rank) -> np.ndarray:
    cum, W = zipf_table(K)
    t = (_u32stream(seed, n, offset) * _U(W)) >> _U(32)
    return np.asarray(sym_of_rank, dtype=np.uint8)[np.searchsorted(cum, t, side="right")]


def zipf(seed: int, n: int, offset: int = 0) -> np.ndarray:
    return _zipf(seed, n, offset, 256, ZIPF256)


def text(seed: int, n: int, offset: int = 0) -> np.ndarray:
    return _zipf(seed, n, offset, 96, [ord(c) for c in RANK])


def generate(kind: str, seed: int, n: int, offset: int = 0) -> np.ndarray:
    if kind == "uniform":
        return uniform(seed, n, offset)
    if kind == "zipf":
        return zipf(seed, n, offset)
    if kind == "text":
        return text(seed, n, offset)
    if kind == "zeros":
        return np.zeros(n, dtype=np.uint8)
    raise ValueError(f"unknown synthetic stream kind {kind!r}")
