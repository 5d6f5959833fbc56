#!/usr/bin/env python3
"""Regenerates tests/golden/model_vectors.json: the reference's host-callable
codec driven with caller-owned model state.

Run in the build container only (needs /root/reference to build oracle/_ref):

    python tests/golden/make_golden_model.py

Each case is a chain of segments coded one after another WITHOUT re-initialising
the model in between -- the state arCompress / arDecompress leave in the
caller's AdaptiveProbabilityRange (/root/reference/src/gpuar.h:42-48, :75-76) is
the next call's starting state.  Expected packets, Fenwick arrays and totals all
come from the reference's own functions through oracle/_ref.  The file is data:
generator parameters, and the reference's outputs for them.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gpuar_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CHAINS = [
    ("uniform", 1, [3000, 4096, 1000]),
    ("zipf", 2, [1, 0, 17, 4000]),
    ("text", 3, [8192, 4096]),             # total ends at 256 + 12288, still below 2^14
    ("text", 4, [100, 100, 100, 100, 100]),
    ("uniform", 5, [0]),
    ("zipf", 6, [8191]),
]


def main():
    ref = O.ReferenceOracle()
    ranges0, total0 = ref.model_init()
    cases = []
    for kind, seed, sizes in CHAINS:
        ranges, total = ranges0.copy(), total0
        dranges, dtotal = ranges0.copy(), total0
        segs, off = [], 0
        for n in sizes:
            data = synth.generate(kind, seed, n, offset=off).tobytes()
            off += n
            pkt, ranges, total = ref.encode_packet_model(data, ranges, total)
            back, dranges, dtotal = ref.decode_packet_model(pkt, dranges, dtotal)
            assert back == data and dtotal == total and np.array_equal(dranges, ranges)
            seg = {"n": n, "offset": off - n, "clen": len(pkt), "packet_md5": hashlib.md5(pkt).hexdigest(),
                   "total_after": total, "ranges_after_hex": ranges.astype("<u2").tobytes().hex()}
            if len(pkt) <= 64:
                seg["packet_hex"] = pkt.hex()
            segs.append(seg)
        cases.append({"kind": kind, "seed": seed, "segments": segs})
    with open(os.path.join(HERE, "model_vectors.json"), "w") as f:
        json.dump({"_provenance": "expected outputs produced by oracle/_ref (the reference's unmodified "
                                  "initializeAdaptiveProbabilityRangeList / arCompress / arDecompress); "
                                  "regenerate with tests/golden/make_golden_model.py",
                   "initial_total": total0, "initial_ranges_hex": ranges0.astype("<u2").tobytes().hex(),
                   "cases": cases}, f, indent=0)
    print(f"wrote {len(cases)} chains")


if __name__ == "__main__":
    main()
