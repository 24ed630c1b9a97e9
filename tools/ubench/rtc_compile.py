import ctypes, sys, os
rtc = ctypes.CDLL("/opt/rocm/lib/libhiprtc.so")
def compile(src, opts, out):
    prog = ctypes.c_void_p()
    r = rtc.hiprtcCreateProgram(ctypes.byref(prog), src.encode(), b"jit.hip", 0, None, None)
    assert r == 0, r
    arr = (ctypes.c_char_p * len(opts))(*[o.encode() for o in opts])
    r = rtc.hiprtcCompileProgram(prog, len(opts), arr)
    n = ctypes.c_size_t()
    rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
    if n.value > 1:
        log = ctypes.create_string_buffer(n.value); rtc.hiprtcGetProgramLog(prog, log); print(log.value.decode()[:2000])
    assert r == 0, r
    rtc.hiprtcGetCodeSize(prog, ctypes.byref(n))
    code = ctypes.create_string_buffer(n.value); rtc.hiprtcGetCode(prog, code)
    open(out, "wb").write(code.raw)
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "advancedmh.jl_amd", "csrc")
src = '#include "mhx_rwmh_kernels.h"\n'
base = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-DMHX_REAL64=1", "-I" + csrc,
        "-DMHX_JIT_RWMH_COOP=1", "-DMHX_JIT_L=2", "-DMHX_JIT_NBL=13", "-DMHX_JIT_TK=0", "-DMHX_JIT_PK=0", "-DMHX_JIT_MOM=0", "-DMHX_JIT_GEN=1"]
compile(src, base + sys.argv[2:], sys.argv[1])
