// mhx_jit_ext.h -- the run-time kernels compiled by the ROCm installation's own clang++ instead of the hiprtc library.
//
// hiprtc is whatever `libhiprtc.so.7` / `libamd_comgr.so.3` the PROCESS loaded first: inside a Python process that imported a PyTorch
// wheel it is the wheel's bundled copy (ROCm 7.0's compiler under a ROCm 7.2 installation), and the cooperative RWMH kernel it builds
// runs 7 % behind the same source built by hipcc (179 against 133 spilled scalar registers; profiles/r06_jit_compiler_ab.txt).
// The compiler binary of the installation is not subject to that: <rocm>/lib/llvm/bin/clang++ -x hip --offload-device-only is what
// built the pre-built kernels of this library.  Pure host code (no HIP), shared by both instantiations of the engine.
#pragma once
#include <string>
#include <vector>

// identity of the offline compiler for the on-disk cache key: "clang++:<path>:<size>:<mtime>", "" when none is found.
// Looked for once per process: $MHX_JIT_CLANG (a path; "0" or "" = none), $ROCM_PATH/lib/llvm/bin/clang++, /opt/rocm/lib/llvm/bin/clang++
const std::string& mhx_jit_ext_identity();
// compile `source` (+ headers by name) to a gfx950 code object with the options hiprtc would get; false (and `log`) when there is no
// compiler or it failed -- the caller then asks hiprtc, whose log is the one a user sees
bool mhx_jit_ext_compile(const std::string& source, const char* const* hdr_src, const char* const* hdr_name, int nhdr,
                         const std::vector<std::string>& opts, std::vector<char>* code, std::string* log);
