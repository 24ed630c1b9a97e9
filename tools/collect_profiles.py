#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of tools/profile_round.sh from gpurun_out/ into profiles/ and refresh profiles/traffic.json
(what bench.py reads into roofline.traffic / roofline.valu).  Usage: tools/collect_profiles.py <tag> [<tag> ...]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = {"c2": ("rwmh_coop", "mhx_jit_rwmh_reg"), "c5": ("rwmh_coop",), "c3": ("mhx_jit_emcee_sweep", "mhx_jit_emcee_mfma_sweep", "mhx_jit_emcee_persist", "emcee_half"), "c4": ("k_ram<", "k_ram_defer<"), "c1": ("k_rwmh_wave",)}


def dominant(cfg, name):
    return any(d in name for d in DOMINANT[cfg]) and "init" not in name


def main(tags):
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for tag in tags:
        for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_*"))):
            if not os.path.isdir(d):
                continue
            name = os.path.basename(d)                       # <tag>_<cfg>_<dtype>
            _, cfg, dt = name.rsplit("_", 2)
            if cfg not in DOMINANT:                          # (another tool's directory under the same tag, e.g. r05_isa_c2)
                continue
            sj = os.path.join(d, "summary.json")
            if not os.path.exists(sj):
                continue
            rep = json.load(open(sj))
            bench = json.loads(open(os.path.join(d, "bench.json")).read().strip() or "{}")
            shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", name + "_summary.txt"))
            shutil.copy(os.path.join(d, "bench.json"), os.path.join(ROOT, "profiles", name + "_bench.json"))
            ks = glob.glob(os.path.join(d, "ktrace", "**", "*kernel_stats.csv"), recursive=True)
            if ks:
                shutil.copy(ks[0], os.path.join(ROOT, "profiles", name + "_kernel_stats.csv"))
            # the dominant kernel's counters, per mhx_run_sample call (= bench step): per-dispatch mean x dispatches per step
            key = next((k for k in rep.get("counters", {}) if dominant(cfg, k)), None)
            if key is None or not bench:
                continue
            c = rep["counters"][key]
            per_step = bench["config"].get("launches_per_step", 1)
            stats = next((r for r in rep.get("kernel_stats", []) if r["Name"] == key or dominant(cfg, r["Name"])), None)
            entry = {
                "units_per_launch": bench["config"]["units_per_step_per_gpu"],
                "dispatches_per_step": per_step,
                "FETCH_SIZE_KiB_per_dispatch": c["FETCH_SIZE"]["mean"], "WRITE_SIZE_KiB_per_dispatch": c["WRITE_SIZE"]["mean"],
                "hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0 * per_step,
                "valu_insts_per_launch": c["SQ_INSTS_VALU"]["mean"] * per_step,
                "salu_insts_per_launch": c["SQ_INSTS_SALU"]["mean"] * per_step,
                "sq_wave_cycles": c["SQ_WAVE_CYCLES"]["mean"], "sq_wait_inst_any": c["SQ_WAIT_INST_ANY"]["mean"],
                "sq_wait_any": c["SQ_WAIT_ANY"]["mean"], "sq_active_inst_valu": c["SQ_ACTIVE_INST_VALU"]["mean"],
                "kernel": key, "trace_avg_ns_per_dispatch": float(stats["AverageNs"]) if stats else None,
                "bench_hip_event_ms_per_dispatch": bench["roofline"]["avg_launch_ms"],
                "correction": "FETCH_SIZE x2 (gfx950 half-count, MI355X_MICROARCH.md HBM section), WRITE_SIZE x1 (calibrated on the "
                              "known byte counts of the C2 and C4 kernels)",
                "source": "profiles/%s_summary.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_*, separate passes)" % name,
            }
            # variants of a config profiled under their own tag: r04rot_c3_f64 -> c3_rotated_f64, r04ban_c5_f64 -> c5_banana_f64, ...
            variant = {"rot": "_rotated", "ban": "_banana", "mov": "_moving", "fix": "_fixed", "lit": "_literal", "usr": "_user", "def": "_deferred", "sml": "_small"}.get(tag[-3:], "")
            traffic["%s%s_%s" % (cfg, variant, dt)] = entry
            print(name, "-> traffic[%s%s_%s]: hbm %.4g B, valu %.4g per step; trace %.4g ns vs HIP events %.4g ns per dispatch" % (
                cfg, variant, dt, entry["hbm_bytes_per_launch"], entry["valu_insts_per_launch"], entry["trace_avg_ns_per_dispatch"] or 0,
                entry["bench_hip_event_ms_per_dispatch"] * 1e6))
    json.dump(traffic, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
