#!/usr/bin/env python3
"""isa_mix.py -- a class-weighted VALU issue bound for one kernel (`frac_4cycle_class = 2 x frac` assumed every
instruction to be in the 4-cycle class; this replaces the assumption by a count).

Two sources, combined:
  * DYNAMIC class counts from the PMC class counters of gfx950 (rocprofv3 --pmc, separate passes, tools/isa_mix_run.sh):
    SQ_INSTS_VALU and its breakdown SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_{F32,F64}, _INT32, _INT64, _CVT -- what the waves really issued;
  * the STATIC mnemonic histogram of the kernel's code object (llvm-objdump of the gfx950 code inside the hipcc object), used only to
    split a PMC class into issue-cost groups (INT32 = v_bitop3 / v_mul_lo,hi / v_mad_u64_u32 / simple logic and adds; "other" =
    SQ_INSTS_VALU minus the classes = moves, selects, DPP, readlane, compares) in the static proportions of that class's mnemonics;
  * issue cost per group = what tools/ubench/valu_rates.hip measured on this chip at 8 waves per SIMD (profiles/r02a_valu_rates.log),
    as a ratio to v_fma_f32 times the architectural 2 cycles of a wave64 instruction on a SIMD-32.

    valu_weighted_frac = sum_g count_g x cycles_g / (launch seconds x clock x SIMDs)          (1024 SIMDs, 2.4 GHz)

It is an ISSUE bound: 1.0 would mean the VALU pipes never idle.  `valu_frac` (every instruction priced at 2 cycles) is the same sum
with cycles_g = 2.

    tools/isa_mix.py --object advancedmh.jl_amd/csrc/mhx_api_f64.o --kernel 'k_rwmh_coopILi2ELi13ELi0ELi0ELb0ELi1E' \\
                     --pmc gpurun_out/r05_isa_c2/summary.json --launch-ms 2.93 --out profiles/r05_isa_mix_c2.json
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

LLVM = "/opt/rocm/lib/llvm/bin"
CLOCK_HZ, SIMDS = 2.4e9, 1024
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rates(path=os.path.join(ROOT, "profiles", "r02a_valu_rates.log")):
    """ns per instruction per wave at 8 waves per SIMD (the saturated issue rate) by micro-benchmark name"""
    out, on = {}, False
    for ln in open(path):
        if ln.startswith("---"):
            on = "8 per SIMD" in ln
            continue
        m = re.match(r"(\S[\S ]*?)\s+blocks=\s*\d+\s+memtime ticks/instr=\s*[\d.]+\s+ns/instr=\s*([\d.]+)", ln)
        if on and m:
            out[m.group(1).strip()] = float(m.group(2))
    return out


def group_cycles():
    """issue cycles per wave64 instruction of each cost group: measured ratio to v_fma_f32 x 2"""
    r = rates()
    base = r["v_fma_f32"]
    rel = lambda *names: sum(r[n] for n in names) / len(names) / base
    return {
        "simple32": 2.0,                                                    # v_fma/mul/add_f32, logic, adds, shifts, moves: the 2-cycle class (by definition)
        "pk32": 2.0 * rel("v_pk_fma_f32"),
        "int_mul_bitop": 2.0 * rel("v_bitop3_b32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24"),
        "mad_u64": 2.0 * (r["mad_u64+xor"] - r["v_xor_b32"]) / base,        # (measured as a dependent mad + xor pair)
        "f64_arith": 2.0 * rel("v_fma_f64", "v_mul_f64", "v_add_f64"),
        "f64_trans": 2.0 * rel("v_rcp_f64", "v_sqrt_f64", "v_rsq_f64"),
        "f32_trans": 2.0 * rel("v_sqrt_f32"),
        "cvt": 2.0 * rel("v_cvt_f32_u32"),
        "acc": 2.0 * r["acc wr+rd"] / 2.0 / base,                            # one v_accvgpr_read / _write (measured as a pair)
    }


def cost_group(mn):
    """issue-cost group of a VALU mnemonic"""
    m = mn.split("_e32")[0].split("_e64")[0].split("_dpp")[0].split("_sdwa")[0]
    if m.startswith("v_accvgpr"):
        return "acc"
    if m.startswith("v_pk_"):
        return "pk32"
    if re.search(r"_f64$|f64_", m) and re.match(r"v_(rcp|rsq|sqrt)_", m):
        return "f64_trans"
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|f16|legacy|iflag)", m):
        return "f32_trans"
    if re.search(r"_f64$", m) or m in ("v_fma_f64", "v_ldexp_f64", "v_div_scale_f64", "v_div_fmas_f64", "v_div_fixup_f64", "v_frexp_mant_f64", "v_rndne_f64", "v_trunc_f64", "v_floor_f64", "v_fract_f64"):
        return "f64_arith"
    if m.startswith("v_cvt_"):
        return "cvt"
    if m.startswith("v_mad_u64_u32") or m.startswith("v_mad_i64_i32"):
        return "mad_u64"
    if m.startswith("v_bitop3") or re.match(r"v_mul_(lo|hi)_[ui]32", m) or re.match(r"v_mul_[ui]32_[ui]24", m) or m.startswith("v_mad_u32_u24") or m.startswith("v_mad_i32_i24"):
        return "int_mul_bitop"
    return "simple32"


def pmc_class(mn):
    """the PMC breakdown counter a mnemonic is counted by (None: only in SQ_INSTS_VALU -- moves, selects, compares, DPP moves)"""
    m = mn.split("_e32")[0].split("_e64")[0].split("_dpp")[0].split("_sdwa")[0]
    for w in ("F64", "F32"):
        s = "_f" + w[1:]
        if m.endswith(s) or (s + "_") in m:
            if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_", m):
                return "TRANS_" + w
            if re.match(r"v_(fma|fmac|fmaak|fmamk|mad|mac|div_fmas|pk_fma)_", m):
                return "FMA_" + w
            if re.match(r"v_(mul|pk_mul|ldexp)_", m):
                return "MUL_" + w
            if re.match(r"v_(add|sub|subrev|pk_add)_", m):
                return "ADD_" + w
            if m.startswith("v_cvt_"):
                return "CVT"
            return None                                                     # min / max / cmp / cndmask / div_scale ...: not in a class counter
    if m.startswith("v_cvt_"):
        return "CVT"
    if re.match(r"v_(mad_u64_u32|mad_i64_i32|lshlrev_b64|lshrrev_b64|ashrrev_i64|add_co_u32|addc_co_u32|sub_co_u32|subb_co_u32)", m) and "64" in m:
        return "INT64"
    # (v_bitop3_b32 is NOT counted by SQ_INSTS_VALU_INT32 on gfx950: C2 issues ~2.5e8 of them per launch against an INT32 count of
    # 5e7 -- profiles/r05_isa_mix_c2.json; it stays in the unclassified remainder, split by static shares like the moves)
    if re.match(r"v_(add|sub|subrev|mul|mad|and|or|xor|not|bfe|bfi|lshl|lshr|ashr|alignbit|alignbyte|min|max|perm|sad|xnor|and_or|or3|add3|lshl_add|lshl_or|xad|mbcnt|bcnt|ffbh|ffbl)", m) and re.search(r"(_[uib]32|_u24|_i24|_b32|_u32_u24)", m):
        return "INT32"
    return None


def disassemble(obj, kernel_re):
    tmp = tempfile.mkdtemp()
    fb, co = os.path.join(tmp, "fb"), os.path.join(tmp, "dev.co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fb, obj, os.path.join(tmp, "discard.o")])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fb,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", co], text=True)
    cur, hist, found = None, Counter(), None
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = m.group(1)
            if re.search(kernel_re, cur):
                assert found in (None, cur), "kernel pattern matches %s and %s" % (found, cur)
                found = cur
            continue
        if cur is not None and cur == found:
            t = ln.strip().split()
            if t and re.match(r"^[vsdbgt]_|^buffer_|^global_|^flat_|^ds_|^scratch_", t[0]):
                hist[t[0]] += 1
    assert found, "no kernel matches %r" % kernel_re
    return found, hist


def disassemble_hsaco(path, kernel_re):
    """the same histogram from a loadable code object (a run-time kernel kept by the JIT cache)"""
    txt = subprocess.check_output([LLVM + "/llvm-objdump", "-d", path], text=True)
    cur, hist, found = None, Counter(), None
    for ln in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = m.group(1)
            if re.search(kernel_re, cur) and found is None:
                found = cur
            continue
        if cur is not None and cur == found:
            t = ln.strip().split()
            if t and re.match(r"^[vsdbgt]_|^buffer_|^global_|^flat_|^ds_|^scratch_", t[0]):
                hist[t[0]] += 1
    assert found, "no kernel matches %r in %s" % (kernel_re, path)
    return found, hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--object")
    ap.add_argument("--hsaco", help="a code object of the JIT cache instead of --object (run-time kernels)")
    ap.add_argument("--kernel", required=True, help="regex on the (mangled) kernel symbol")
    ap.add_argument("--pmc", help="summary.json of tools/summarize_profile.py over the pmc_valu* passes (tools/isa_mix_run.sh)")
    ap.add_argument("--pmc-kernel", help="regex on the kernel name in the PMC summary (default: derived from --kernel)")
    ap.add_argument("--launch-ms", type=float, help="average launch duration (HIP events / rocprofv3 trace)")
    ap.add_argument("--out")
    a = ap.parse_args()
    sym, hist = disassemble_hsaco(a.hsaco, a.kernel) if a.hsaco else disassemble(a.object, a.kernel)
    valu = Counter({k: v for k, v in hist.items() if k.startswith("v_")})
    cyc = group_cycles()
    static_groups = Counter()
    for mn, n in valu.items():
        static_groups[cost_group(mn)] += n
    rep = {"kernel": sym, "static": {"valu_instructions": sum(valu.values()), "salu": sum(v for k, v in hist.items() if k.startswith("s_")),
                                      "vmem_lds": sum(v for k, v in hist.items() if not k.startswith(("v_", "s_"))),
                                      "by_cost_group": dict(static_groups), "top": valu.most_common(25)},
           "cycles_per_instruction_by_group": {k: round(v, 3) for k, v in cyc.items()},
           "source_of_cycles": "profiles/r02a_valu_rates.log at 8 waves per SIMD, ratio to v_fma_f32 x 2 cycles"}
    static_w = sum(static_groups[g] * cyc[g] for g in static_groups) / max(1, sum(static_groups.values()))
    rep["static"]["mean_cycles_per_valu_instruction"] = round(static_w, 3)
    if a.pmc:
        summ = json.load(open(a.pmc))["counters"]
        kre = a.pmc_kernel or "k_rwmh_coop"
        rows = [v for k, v in summ.items() if re.search(kre, k)]
        assert rows, "no kernel in %s matches %r: %s" % (a.pmc, kre, list(summ)[:5])
        # the counters of one kernel may sit under several spellings of its name; the largest SQ_INSTS_VALU is the bench kernel
        cnt = max(rows, key=lambda r: r.get("SQ_INSTS_VALU", {}).get("mean", 0.0))
        get = lambda name: cnt.get("SQ_INSTS_VALU_" + name, {}).get("mean")
        total = cnt["SQ_INSTS_VALU"]["mean"]
        classes = ["ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "INT32", "INT64", "CVT"]
        dyn = {c: get(c) for c in classes if get(c) is not None}
        other = total - sum(dyn.values())
        # split each PMC class into cost groups by the static shares of that class's mnemonics
        by_group = Counter()
        static_by_class = {}
        for mn, n in valu.items():
            static_by_class.setdefault(pmc_class(mn), Counter())[cost_group(mn)] += n
        for c, n in list(dyn.items()) + [(None, other)]:
            shares = static_by_class.get(c)
            if not shares:
                by_group["simple32"] += n
                continue
            tot = float(sum(shares.values()))
            for g, k in shares.items():
                by_group[g] += n * k / tot
        weighted = sum(by_group[g] * cyc[g] for g in by_group)
        rep["dynamic"] = {"SQ_INSTS_VALU": total, "by_pmc_class": dyn, "other_moves_selects_compares_dpp": other,
                          "by_cost_group": {g: round(v) for g, v in by_group.items()},
                          "mean_cycles_per_valu_instruction": round(weighted / total, 3)}
        if a.launch_ms:
            sec = a.launch_ms * 1e-3
            rep["launch_ms"] = a.launch_ms
            rep["valu_frac_2cycle"] = round(total * 2.0 / (sec * CLOCK_HZ * SIMDS), 4)
            rep["valu_weighted_frac"] = round(weighted / (sec * CLOCK_HZ * SIMDS), 4)
    text = json.dumps(rep, indent=1)
    if a.out:
        open(a.out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
