"""C3 (dense-rotated) on the scalar-factor form of the cooperative stretch move: parity against the oracle on a small ensemble,
then the half-step time at full size for the lane-group form (MHX_EMCEE_SCALAR=0) and the scalar form with 4 / 8 / 16 waves.
Run on the GPU box:  python tools/c3_scalar_probe.py [f64|f32]"""
import os
import sys
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
sys.path.insert(0, ROOT)
import numpy as np


def sigma(d, rotated=True):
    i = np.arange(d)
    S = 0.9 ** np.abs(i[:, None] - i[None, :])
    if rotated:
        Q, _ = np.linalg.qr(np.random.default_rng(50).normal(size=(d, d)))
        S = Q @ S @ Q.T
    return S


def child(dt, mode, d, W):
    import mhx
    import _opts
    _opts.bridge(mhx)              # MHX_* variables of the command line -> explicit engine options (tools only)
    from oracle import oracle as O
    O.set_dtype(dt)
    Sig = sigma(d)
    model = mhx.DensityModel(mhx.CorrGaussian(Sig))
    if mode == "parity":
        run = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=3, ctx=mhx.Context(0, dt))
        run.init(None)
        run.sample(6, 2, 1, 0, save=True)
        st = run.stats()
        val, acc = run.samples()
        prior = O.Proposal(O.PROP_ISO, 1.0)
        ref = O.emcee(O.corr_gauss_from_cov(Sig, reduce_lanes=st["reduce_lanes"]), 2.0, 1, O.schedule(6, 2), 3, 0, W, None, prior=prior)
        bits = np.uint64 if dt == "f64" else np.uint32
        ok = np.array_equal(val.view(bits), ref["samples"].view(bits)) and np.array_equal(acc, ref["accepted"])
        print("parity d=%d W=%d %s: variant %d lanes %d -> %s (acc %.3f)" % (d, W, dt, st["kernel_variant"], st["reduce_lanes"],
                                                                        "BIT-EXACT" if ok else "MISMATCH", acc[1:].mean()), flush=True)
        return
    run = mhx.Run(model, mhx.Ensemble(W, mhx.StretchProposal(mhx.MvNormal(mhx.zeros(d), mhx.I))), seed=3, ctx=mhx.Context(0, dt))
    run.init(None)
    for _ in range(3):
        run.sample(500, 1, 1, 0, save=True)
    best = 1e9
    for _ in range(5):
        run.sample(500, 1, 1, 0, save=True)
        st = run.stats()
        best = min(best, st["kernel_ms"] * 1e3 / st["launches"])
    print("time d=%d W=%d %s scalar=%s: variant %d lanes %d: %.2f us per half-step, %.3e moves/s" % (
        d, W, dt, os.environ.get("MHX_EMCEE_SCALAR", "default"), st["kernel_variant"], st["reduce_lanes"], best, W / 2 / (best * 1e-6)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    dts = sys.argv[1:] or ["f64", "f32"]
    for dt in dts:
        for sc in ("4", "8", "16", None):
            for d, W in ((50, 200), (50, 130), (33, 64), (96 if dt == "f64" else 100, 256)):
                env = dict(os.environ)
                if sc is not None:
                    env["MHX_EMCEE_SCALAR"] = sc
                subprocess.call([sys.executable, __file__, dt, "parity", str(d), str(W)], env=env)
        for sc in ("0", "4", "8", "16"):
            env = dict(os.environ, MHX_EMCEE_SCALAR=sc)
            subprocess.call([sys.executable, __file__, dt, "time", "50", "16384"], env=env)
