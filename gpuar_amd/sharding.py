"""Packet-range sharding of one byte stream over the GPUs of a node.

Every 8192-byte packet is coded from a fresh model, so the stream shards into
contiguous packet ranges with no exchange step: rank r encodes its range into
its own segment and the host concatenates the segments in rank order behind
the 20-byte header.  Ranges are multiples of 64 packets (whole wavefronts)
except that the last non-empty rank takes the tail.
"""
from __future__ import annotations

from typing import List, Tuple

PACKET = 8192
WAVE_PACKETS = 64


def plan_shards(n_bytes: int, world: int) -> List[Tuple[int, int]]:
    """[(byte offset, byte length)] per rank; lengths sum to n_bytes; offsets are packet-aligned."""
    if world < 1:
        raise ValueError("world must be >= 1")
    n_packets = (n_bytes + PACKET - 1) // PACKET
    per = (n_packets + world - 1) // world
    per = (per + WAVE_PACKETS - 1) // WAVE_PACKETS * WAVE_PACKETS
    out = []
    for r in range(world):
        p0 = min(r * per, n_packets)
        p1 = min((r + 1) * per, n_packets)
        off = p0 * PACKET
        end = min(p1 * PACKET, n_bytes)
        out.append((off, max(0, end - off)))
    return out


def weak_shard(bytes_per_rank: int, rank: int) -> Tuple[int, int]:
    """Benchmark sharding with fixed work per GPU: rank r owns bytes [r*B, (r+1)*B), B packet-aligned."""
    b = bytes_per_rank // PACKET * PACKET
    return rank * b, b


def concat_segments(segments) -> bytes:
    """Host-side concatenation in rank order (what follows the 20-byte header in the .gip file)."""
    return b"".join(bytes(s) for s in segments)
