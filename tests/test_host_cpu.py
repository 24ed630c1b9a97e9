"""CPU (no GPU): the C-ABI library loads and exports every symbol include/mhx.h declares, host-side
argument checking mirrors the reference's error behaviour, and the multi-GPU statistics path
(chains sharded by global id + one all-reduce) is exercised with world_size 2 on gloo."""
import ctypes
import json
import os
import re
import socket

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(mhx):
    hdr = open(os.path.join(ROOT, "include", "mhx.h")).read()
    declared = sorted(set(re.findall(r"\b(mhx_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = ctypes.CDLL(mhx.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), "libmhx.so does not export %s" % name
    assert sorted(mhx.EXPORTS) == declared
    assert lib.mhx_version() == 600


def test_no_gpu_fails_loudly_not_silently(mhx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(mhx.MhxError):                           # there is no CPU fallback
        mhx.sample(mhx.DensityModel(mhx.IsoGaussian(2)), mhx.RWMH(2), 3)


def test_host_side_argument_errors(mhx):
    assert mhx.RWMH(mhx.MvNormal(np.ones(3), mhx.I)).proposal.proposal.mean.sum() == 3   # a drifting walk is allowed ...
    with pytest.raises(mhx.ArgumentError):                      # ... but it is not symmetric (src/proposal.jl:195)
        mhx.SymmetricRandomWalkProposal(mhx.MvNormal(np.ones(3), mhx.I))
    with pytest.raises(mhx.ArgumentError):
        mhx.DensityModel(lambda x: 0.0)
    with pytest.raises(mhx.ArgumentError):
        mhx.MvNormal(mhx.zeros(3), np.eye(2))
    with pytest.raises(mhx.PosDefException):
        mhx.CorrGaussian(np.array([[1.0, 2.0], [2.0, 1.0]]))
    with pytest.raises(mhx.ArgumentError):
        mhx.Ensemble(10, mhx.MvNormal(mhx.zeros(2), mhx.I))     # only StretchProposal (as the reference)
    with pytest.raises(mhx.ArgumentError):
        mhx.MetropolisHastings("static")
    mv = mhx.RWMH(3).proposal.proposal                          # RWMH(d::Int) == MvNormal(zeros(d), I), mh-core.jl:51
    assert mv.dim == 3 and mv.scale == 1.0 and mv.kind == 0
    mv = mhx.RWMH([mhx.Normal(0, 2.0), mhx.Normal(0, 0.5)]).proposal.proposal   # vector of univariates, proposal.jl:26-28
    assert mv.kind == 1 and np.allclose(mv.vec, [2.0, 0.5])
    assert mhx.RobustAdaptiveMetropolis().α == 0.234 and mhx.RobustAdaptiveMetropolis().γ == 0.6   # RAM.jl:78-80
    assert mhx.StretchProposal(mhx.MvNormal(mhx.zeros(2), mhx.I)).stretch_length == 2.0          # emcee.jl:68


def test_shard_chains_partition():
    from mhx.dist import shard_chains
    for total, world in ((262144, 8), (10, 3), (7, 8), (65536, 1)):
        parts = [shard_chains(total, r, world) for r in range(world)]
        assert sum(n for _, n in parts) == total
        nxt = 0
        for first, n in parts:
            assert first == nxt
            nxt += n
    assert shard_chains(262144, 3, 8) == (3 * 32768, 32768)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, q):
    import sys
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "advancedmh.jl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from oracle import oracle as O
    from mhx.dist import allreduce_stats, shard_chains
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    total, d, N = 24, 3, 200
    first, n = shard_chains(total, rank, world)
    r = O.rwmh(O.iso_gauss(d), O.Proposal(O.PROP_ISO, 0.8), O.schedule(N, 50), 5, first, n)
    v = r["samples"].astype(np.float64)
    m, s2 = v.mean(axis=0), v.var(axis=0, ddof=1)
    diag = dict(sum_m=m.sum(axis=1), sum_m2=(m * m).sum(axis=1), sum_v=s2.sum(axis=1), n_chains=n, n_samples=N)
    out = allreduce_stats(diag, int(r["accept_counts"].sum()), n * (N - 1 + 50))
    q.put((rank, out["rhat"], out["ess_between"], out["acceptance_rate"], out["n_chains"]))
    dist.destroy_process_group()


def _gather_rank_main(rank, world, port, q):
    """Each rank changes its own slice of the moving half of a fake ensemble state (CPU tensors standing in for the
    device state) and runs ShardedEnsemble's exchange; afterwards every rank must hold all the slices."""
    import sys
    import torch
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "advancedmh.jl_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mhx.dist import ShardedEnsemble
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    W, P = 23, 8                                           # odd ensemble: halves of 11 and 12, ragged slices
    sh = ShardedEnsemble.__new__(ShardedEnsemble)
    sh.rank, sh.world, sh.group, sh.W, sh.pitch = rank, world, None, W, P
    sh._t = dict(xw=torch.zeros(W, P), lp=torch.zeros(W), acc=torch.zeros(W, dtype=torch.int32), last=torch.zeros(W, dtype=torch.uint8))
    for h in (0, 1):
        lo, cnt = (W // 2, W - W // 2) if h else (0, W // 2)
        sl = ShardedEnsemble.slices(cnt, world)
        b, c = sl[rank]
        for w in range(lo + b, lo + b + c):                  # "move" the walkers of this rank's slice
            sh._t["xw"][w] = float(100 * (rank + 1) + w)
            sh._t["lp"][w] = float(-w)
            sh._t["acc"][w] = w + 1
            sh._t["last"][w] = 1
        sh._all_gather(lo, cnt, sl)
    q.put((rank, sh._t["xw"][:, 0].tolist(), sh._t["lp"].tolist(), sh._t["acc"].tolist(), sh._t["last"].tolist()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_ensemble_exchange_gloo(world):
    """The all-gather of mhx.dist.ShardedEnsemble on gloo with 2 and 3 ranks, equal and ragged slices."""
    import torch.multiprocessing as mp
    from mhx.dist import ShardedEnsemble
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    W = 23
    owner = np.zeros(W, dtype=int)
    for h in (0, 1):
        lo, cnt = (W // 2, W - W // 2) if h else (0, W // 2)
        for r, (b, c) in enumerate(ShardedEnsemble.slices(cnt, world)):
            owner[lo + b:lo + b + c] = r
    want_x = [float(100 * (owner[w] + 1) + w) for w in range(W)]
    for rank, x0, lp, acc, last in res:
        assert x0 == want_x and lp == [float(-w) for w in range(W)]
        assert acc == [w + 1 for w in range(W)] and last == [1] * W


def test_two_rank_statistics_allreduce_gloo(oracle):
    """world_size 2 on gloo: per-shard sums all-reduced == the single-process result over all chains."""
    import torch.multiprocessing as mp
    from mhx.api import combine_diagnostics
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    total, d, N = 24, 3, 200
    r = oracle.rwmh(oracle.iso_gauss(d), oracle.Proposal(oracle.PROP_ISO, 0.8), oracle.schedule(N, 50), 5, 0, total)
    v = r["samples"].astype(np.float64)
    m, s2 = v.mean(axis=0), v.var(axis=0, ddof=1)
    want = combine_diagnostics(m.sum(axis=1), (m * m).sum(axis=1), s2.sum(axis=1), total, N)
    for rank, rhat, essb, accrate, nch in res:
        assert nch == total
        assert np.allclose(rhat, want["rhat"], rtol=1e-12)
        assert np.allclose(essb, want["ess_between"], rtol=1e-9)
        assert abs(accrate - r["accept_counts"].sum() / (total * (N - 1 + 50))) < 1e-12


def test_product_path_never_touches_the_oracle_or_a_cpu_fallback():
    """The oracle is test infrastructure: nothing under advancedmh.jl_amd/ (kernels, C ABI, host mirror) may
    import, include or link it; and the package ships no compatibility layer."""
    pkg = os.path.join(ROOT, "advancedmh.jl_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".h", ".hip", ".inc", ".jl", "Makefile")) or f == "mhx_jit_embed.inc":
                continue
            text = open(os.path.join(dirpath, f), errors="replace").read()
            for needle in ("mhx_oracle", "from oracle", "import oracle", "oracle/", "__HIP_PLATFORM", "hipify", "triton"):
                if needle in text:
                    bad.append((os.path.relpath(os.path.join(dirpath, f), ROOT), needle))
    assert not bad, bad
    # bench.py may use the oracle only in its cpu_baseline leg
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("from oracle import oracle") == 1 and "def cpu_baseline" in bench


def test_sharded_ensemble_slices_partition_a_half():
    """mhx.dist.ShardedEnsemble.slices (the rule of mhx_comm_slice): contiguous, covering the half, sizes within one of each other."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "advancedmh.jl_amd"))
    from mhx.dist import ShardedEnsemble
    for cnt in (1, 2, 7, 8, 50, 8192, 8193):
        for world in (1, 2, 3, 4, 8):
            sl = ShardedEnsemble.slices(cnt, world)
            assert len(sl) == world and sum(c for _, c in sl) == cnt
            pos = 0
            for b, c in sl:
                assert c >= 0 and b == pos
                pos += c
            assert max(c for _, c in sl) - min(c for _, c in sl) <= 1


def test_bench_byte_models_match_the_survey():
    """roofline.achieved = algorithmic bytes / kernel time: the per-unit figures of SURVEY.md 8(d) in both widths (fp64 doubles
    every real; the accept flag stays one byte)."""
    import argparse
    import bench
    ns = argparse.Namespace(dim=0, chains=0, inner=0, lanes=0, c2_literal=False, c3_rotated=False, c4_moving=False, c4_fixed=False,
                            c5_banana=False, normal_gen="auto")
    for dt, B in (("f32", 4), ("f64", 8)):
        c2 = bench.WORKLOADS["c2"](ns, dt)
        per_step = c2.bytes_per_launch() / (c2.C * c2.inner)
        assert abs(per_step - (B * 101 + 1 + 2 * (B * 100 + B + 4 + 1) / c2.inner)) < 1e-6        # 405 B/step as K -> inf (fp32); + x, lp, counter, flag per launch
        c3 = bench.WORKLOADS["c3"](ns, dt)
        assert c3.bytes_per_launch() / c3.units_per_step() == 3 * B * 50 + 2 * B + B * 51 + 1      # 813 B/move (fp32)
        c4 = bench.WORKLOADS["c4"](ns, dt)
        assert c4.bytes_per_launch() / c4.units_per_step() == B * 200 * 201 + 2 * B * 200 + 2 * B  # 162.4 KB/step (fp32)
    ns.c4_fixed = True
    c4f = bench.WORKLOADS["c4"](ns, "f32")
    assert c4f.bytes_per_launch() / c4f.units_per_step() == 4 * 200 * 201 // 2 + 8 * 200 + 8


def _run_bench(argv, env_extra=None, timeout=180):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no launcher around it IS two processes (the flag used to be parsed and ignored).
    Device-free rehearsal (--dry-run: launcher, gloo rendezvous, barrier, max-over-ranks time, summed totals)."""
    rc, line, err = _run_bench(["--gpus", "2", "--dry-run", "--steps", "4"])
    assert rc == 0, err
    assert line["n_ranks"] == 2 and line["distinct_processes"] == 2 and line["dry_run"] is True
    assert line["units_all_ranks"] == 2 * 4 * 1000 and line["value"] is None and line["n_gpus"] == 0
    assert "2 ranks" in line["config"]["collective"]


@pytest.mark.parametrize("fault", ["fail", "hang"])
def test_bench_transport_ladder_ends_in_a_line_when_rccl_init_fails_or_hangs(fault):
    """the first contact with an N-GPU node must not end with NO line.  The bench's collectives are a ladder --
    RCCL behind the C ABI, torch.distributed's nccl backend, gloo -- every rung under a deadline and agreed on by all ranks.  Here
    rung 1 is made to fail / to never answer (--fault-rccl-init), rung 2 has no device (dry run): the flow ends on gloo, the line
    is there, it says which transport carried it and why the others did not."""
    rc, line, err = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3", "--fault-rccl-init", fault, "--transport-timeout", "3"], timeout=240)
    assert rc == 0, err
    assert line is not None and line["n_ranks"] == 2 and line["distinct_processes"] == 2
    cfg = line["config"]
    assert cfg["collective"].startswith("gloo, 2 ranks") and cfg["ranks_reported_by_transport"] == 2
    assert set(cfg["rccl_error"]) == {"rank0", "rank1"}
    want = "injected failure" if fault == "fail" else "no answer within"
    assert all(want in e for e in cfg["rccl_error"].values()), cfg["rccl_error"]
    assert "torch.distributed nccl" in cfg["transport_errors"]


def test_bench_require_rccl_turns_the_fallback_into_an_error():
    rc, line, err = _run_bench(["--gpus", "2", "--dry-run", "--fault-rccl-init", "fail", "--require-rccl", "--transport-timeout", "3"], timeout=240)
    assert rc != 0 and line is None and "--require-rccl" in err


def test_bench_launcher_binds_each_rank_to_its_device_before_hip(tmp_path):
    """launch_ranks: rank r gets HIP_VISIBLE_DEVICES = the r-th entry of the launcher's own list (so that HIP never sees the other
    ranks' devices); --bind ordinal leaves the list alone.  Checked on the environment the children receive (no device needed)."""
    import bench
    import types
    seen = []

    class FakePopen:
        def __init__(self, argv, env=None, stdout=None):
            seen.append(env)

        def poll(self):
            return 0

        def terminate(self):
            pass
    real_popen, real_vis = bench.subprocess.Popen, bench.visible_devices
    bench.subprocess.Popen, bench.visible_devices = FakePopen, lambda: 4
    old = os.environ.get("HIP_VISIBLE_DEVICES")
    try:
        os.environ["HIP_VISIBLE_DEVICES"] = "4,5,6,7"
        args = types.SimpleNamespace(gpus=4, dry_run=False, allow_gloo=False, bind="visible")
        assert bench.launch_ranks(args) == 0
        assert [e["HIP_VISIBLE_DEVICES"] for e in seen] == ["4", "5", "6", "7"] and all(e["MHX_BENCH_BOUND"] == "1" for e in seen)
        assert [e["RANK"] for e in seen] == ["0", "1", "2", "3"] and len({e["MASTER_PORT"] for e in seen}) == 1
        del seen[:]
        del os.environ["HIP_VISIBLE_DEVICES"]
        assert bench.launch_ranks(args) == 0
        assert [e["HIP_VISIBLE_DEVICES"] for e in seen] == ["0", "1", "2", "3"]
        del seen[:]
        args.bind = "ordinal"
        assert bench.launch_ranks(args) == 0
        assert all("HIP_VISIBLE_DEVICES" not in e and "MHX_BENCH_BOUND" not in e for e in seen)
        del seen[:]
        args.gpus = 8                                            # more ranks than devices: refused unless it is the rehearsal
        assert bench.launch_ranks(args) == 2 and not seen
    finally:
        bench.subprocess.Popen, bench.visible_devices = real_popen, real_vis
        if old is None:
            os.environ.pop("HIP_VISIBLE_DEVICES", None)
        else:
            os.environ["HIP_VISIBLE_DEVICES"] = old
    assert bench.pci_number("0000:05:00.0") != bench.pci_number("0000:15:00.0") and bench.pci_number("0000:05:00.0") == (5 << 8)


def test_bench_under_a_launcher_is_one_rank_of_it():
    """the driver's form: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_ranks"] == 2 and line["distinct_processes"] == 2


def test_bench_refuses_to_claim_ranks_or_gpus_it_does_not_have():
    import torch
    # --gpus disagrees with the launcher's WORLD_SIZE: refused before anything runs
    rc, line, err = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and line is None and "WORLD_SIZE=3" in err
    rc, line, err = _run_bench(["--gpus", "1", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc != 0 and line is None
    if not torch.cuda.is_available():
        # fewer devices than ranks: no line (the engine has no CPU path and a 2-GPU line needs 2 GPUs)
        rc, line, err = _run_bench(["--gpus", "2"])
        assert rc != 0 and line is None and "refusing" in err
        rc, line, err = _run_bench([])
        assert rc != 0 and line is None


def test_bench_line_fits_the_drivers_tail():
    """The driver keeps an 8 KB tail of stdout: the whole line has to fit.  Checked on the last full line a GPU box printed
    (profiles/r0[456]*_bench_full.json, committed) and on the worst case the compact blocks can reach."""
    import glob
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[456]*_bench_full.json")))
    for f in files:
        text = open(f).read().strip().splitlines()[-1]
        assert len(text) < 8000, (f, len(text))
        line = json.loads(text)
        for k in ("c1", "c2_literal", "c3", "c3_rotated", "c4", "c4_moving", "c4_fixed", "c5", "c5_banana"):
            assert k in line["configs"], (f, k)
            assert "error" not in line["configs"][k], (f, k, line["configs"][k])
        if os.path.basename(f).startswith("r06"):                # round 6: the boundary's own figure and the many-ensemble config ride in the line
            sa = line["e2e_host"]["save_all"]
            assert sa["compact"] == 1 and sa["value"] >= 1.0e8 and "host_expand_GBps" in sa and "link_GBps" in sa, sa
            assert line["configs"]["c3_small"]["ensembles"] == 256 and line["configs"]["c3"]["bound"] == "latency"
    # a compact block with every optional key and full-width numbers stays small: 9 of them + the top level < 8000
    blk = {"value": 1.2345e9, "ms_per_step": 123.45, "acc": 0.234, "bound": "valu", "frac": 0.4321, "traffic_ratio": 1.2345,
           "launch_us": 12345.0, "kernel": "sequential-ensemble-sweep", "lanes": 64, "hbm_frac": 0.0171, "band": 1,
           "cpu": {"value": 1.2345e6, "cores": 256, "single_thread": 12340.0, "parallel_efficiency": 0.432, "why": "smt-or-memory"}}
    assert len(json.dumps(blk, separators=(",", ":"))) < 400
    assert bench.sig(1234567.891, 5) == 1234600.0 and bench.sig(None) is None and bench.sig(0.000123456, 3) == 0.000123


def test_cpu_baseline_driver_matches_the_oracle_and_reports_threads(oracle):
    """oracle/mhx_oracle_mt.c (bench.py's CPU baseline: POSIX threads, a chain per task) runs the oracle's own sampler: the
    thread count changes the wall time, never a result; busy seconds come back per thread."""
    oracle.set_dtype("f64")
    t, p, s = oracle.iso_gauss(10), oracle.Proposal(oracle.PROP_ISO, 0.5), oracle.schedule(40)
    w, busy = oracle.mt_rwmh(t, p, s, 7, 0, 24, 3, save=True)
    assert w > 0 and busy.shape == (3,) and busy.sum() > 0
    w1, busy1 = oracle.mt_rwmh(t, p, s, 7, 0, 24, 1, save=False)
    assert busy1.shape == (1,)
    tg = oracle.corr_gauss_from_cov(np.eye(6) + 0.3)
    w, busy = oracle.mt_ram(tg, oracle.schedule(1, 30, 1, 30), 4, 0, 8, 2, init1=np.zeros(6))
    assert w > 0 and busy.shape == (2,)
    oracle.set_dtype("f32")
