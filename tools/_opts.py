"""Tools only: the A/B scripts of earlier rounds steered the library through MHX_* environment variables; the library no longer reads
them (include/mhx.h: mhx_ctx_set_option).  `bridge(mhx)` turns the variables a tool was started with into explicit options on the
default contexts -- binding the TOOLS build first when one of them is a probe (whose runs are then marked tainted)."""
import os

PROBES = ("ZIG_PROBE", "EMCEE_PROBE", "EMCEE_STAMPS", "EMCEE_STAMPS_FILE", "ZIG_FORCE_FAIL", "FAULT_SLAB", "JIT_DEFS", "JIT_FLAGS")
NOT_OPTIONS = ("LIB", "DTYPE", "CACHE_DIR", "NO_JIT_CACHE", "RCCL_LIB", "FUZZ_SEED", "BENCH_LAUNCHED", "BENCH_BOUND", "BENCH_FORCE_DIST", "STORE_PORT", "DEVICE", "GEN")


def bridge(mhx):
    env = {k[4:]: v for k, v in os.environ.items() if k.startswith("MHX_") and k[4:] not in NOT_OPTIONS}
    if any(k in PROBES for k in env):
        mhx.use_library(mhx.TOOLS_LIB_PATH)
    for k, v in env.items():
        try:
            mhx.set_option(k, v)
        except Exception as e:                       # not an option of this build: say so, do not pretend
            print("tools/_opts.py: MHX_%s ignored (%s)" % (k, str(e)[:80]))
    return env
