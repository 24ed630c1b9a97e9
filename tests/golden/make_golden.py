#!/usr/bin/env python3
"""Generate the committed fixtures of tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

c1_normal_data.npy  300 draws of Normal(0,1) -- the stand-in for `rand(Normal(0,1), 300)` under
                    Random.seed!(1234) of test/runtests.jl:20-23 (Julia's stream is not reproducible here).
traces.npz          same-seed traces of the three samplers from the CPU oracle in fp32: every HIP kernel must
                    reproduce them bit for bit, and the oracle itself is pinned against drift by them.
traces64.npz        the same cases from the fp64 build of the oracle (the reference's Float64 arithmetic).
The reference (pure Julia) holds no golden vectors of its own; see DESIGN.md section 2.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
import cases  # noqa: E402


def main():
    data = np.random.default_rng(1234).normal(0.0, 1.0, 300).astype(np.float32)
    np.save(os.path.join(HERE, "c1_normal_data.npy"), data)
    for dt, fname in (("f32", "traces.npz"), ("f64", "traces64.npz")):
        O.set_dtype(dt)
        out = {}
        for name, fn in cases.TRACE_CASES.items():
            res = fn(O)
            for k, v in res.items():
                if v is not None:
                    out["%s/%s" % (name, k)] = v
        np.savez_compressed(os.path.join(HERE, fname), **out)
        print("wrote", len(out), "arrays to", fname)


if __name__ == "__main__":
    main()
