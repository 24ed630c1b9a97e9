"""Where the waves of ONE scalar-factor half-step launch spend their time (MHX_EMCEE_STAMPS): s_memtime / s_memrealtime at the
phase boundaries of every wave.  Run on the GPU box:  MHX_EMCEE_SCALAR=8 MHX_EMCEE_SCAL_WPB=32 python tools/c3_stamps.py"""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
sys.path.insert(0, ROOT)
import numpy as np

os.environ["MHX_EMCEE_STAMPS"] = "1"
fn = os.path.join(tempfile.gettempdir(), "mhx_stamps.bin")
os.environ["MHX_EMCEE_STAMPS_FILE"] = fn
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mhx
import _opts
_opts.bridge(mhx)                  # MHX_* variables of the command line -> explicit engine options (tools only)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from c3_scalar_probe import sigma

d, W = 50, 16384
dt = sys.argv[1] if len(sys.argv) > 1 else "f64"
run = mhx.Run(mhx.DensityModel(mhx.CorrGaussian(sigma(d))), mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))),
              seed=3, ctx=mhx.Context(0, dt))
run.init(None)
names = ["entry", "phase1 done (rows, move, y->LDS)", "barrier 1 passed", "y in registers", "mat-vec done, q->LDS", "barrier 2 passed",
         "accept + state stores issued", "record stores issued (exit)"]
for rep in range(3):
    run.sample(200, 1, 1, 0, save=True)
    st = run.stats()
    h = np.fromfile(fn, dtype=np.uint64).reshape(-1, 8, 2)
    h = h[h[:, 0, 1] > 0]
    rt = h[:, :, 1].astype(np.int64)                     # s_memrealtime: 100 MHz, comparable across the chip
    mt = h[:, :, 0].astype(np.int64)                     # s_memtime: core clock of the wave's XCD
    t0 = rt[:, 0].min()
    print("launch %d: %.2f us per half-step by HIP events; %d waves stamped" % (rep, st["kernel_ms"] * 1e3 / st["launches"], len(h)))
    for k in range(8):
        r = (rt[:, k] - t0) * 0.01
        print("  %-36s first %.2f  median %.2f  last %.2f us after the first wave's entry" % (names[k], r.min(), np.median(r), r.max()))
    dm = np.diff(mt, axis=1)
    print("  per-wave core-clock cycles between stamps (median): " + " ".join("%d" % v for v in np.median(dm, axis=0)))
