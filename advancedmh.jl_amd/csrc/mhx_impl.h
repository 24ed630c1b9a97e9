// mhx_impl.h -- the two instantiations of the engine behind the C ABI.
//
// libmhx.so holds the engine twice: namespace mhx_f32 (mhx_real = float) and namespace mhx_f64 (mhx_real = double; the
// reference computes in Float64), both compiled from mhx_api.hip + the kernel headers.  The public entry points of
// include/mhx.h live in mhx_abi.cpp and forward to the instantiation the handle belongs to (the first word of every
// handle is its mhx_dtype).  This header declares the api_* functions of both, so that definitions (mhx_api.hip) and
// calls (mhx_abi.cpp) are checked against ONE set of prototypes by the C++ linker (mangled names carry the types).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/mhx.h"

// thread-local error message shared by everything in the library (mhx_last_error); returns `code`
int mhx_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

struct mhx_handle_hdr { int32_t dtype; };

// hiprtc is entered by one thread at a time (the member threads of a group may each need a specialisation): mhx_abi.cpp
void mhx_jit_lock();
void mhx_jit_unlock();

#define MHX_IMPL_DECLARE(NS, REAL)                                                                                     \
    namespace NS {                                                                                                     \
    struct mhx_ctx;                                                                                                    \
    struct mhx_target;                                                                                                 \
    struct mhx_run;                                                                                                    \
    int api_ctx_create(int device, mhx_ctx** out);                                                                     \
    int api_ctx_destroy(mhx_ctx* ctx);                                                                                 \
    int api_ctx_device(const mhx_ctx* ctx);                                                                            \
    int api_ctx_set_option(mhx_ctx* ctx, const char* name, const char* value);                                         \
    int api_ctx_get_option(const mhx_ctx* ctx, const char* name, char* buf, size_t len);                               \
    int api_ctx_pci_bus_id(const mhx_ctx* ctx, char* buf, size_t len);                                                 \
    int api_run_shape(const mhx_run* r, int32_t* dim, int32_t* nchains);                                               \
    int api_ctx_jit_counts(const mhx_ctx* ctx, long* compiles, long* cache_hits);                                      \
    int api_ctx_jit_compiler(const mhx_ctx* ctx, char* compiler, size_t len, long* ext_compiles);                      \
    int api_ctx_host_pin_counts(const mhx_ctx* ctx, long* registered, long* released);                                  \
    int api_target_builtin(mhx_ctx* ctx, int kind, int dim, const REAL* params, size_t nparams, mhx_target** out);     \
    int api_target_from_hip_source(mhx_ctx* ctx, const char* src, int dim, const REAL* data, size_t ndata,             \
                                   mhx_target** out);                                                                  \
    int api_target_destroy(mhx_target* t);                                                                             \
    int api_target_eval(mhx_ctx* ctx, const mhx_target* t, const REAL* x, int n, REAL* lp);                            \
    int api_rwmh_create(mhx_ctx* ctx, const mhx_target* t, const mhx_rwmh_cfg* cfg, mhx_run** out);                    \
    int api_emcee_create(mhx_ctx* ctx, const mhx_target* t, const mhx_emcee_cfg* cfg, mhx_run** out);                  \
    int api_ram_create(mhx_ctx* ctx, const mhx_target* t, const mhx_ram_cfg* cfg, mhx_run** out);                      \
    int api_mala_create(mhx_ctx* ctx, const mhx_target* t, const mhx_mala_cfg* cfg, mhx_run** out);                    \
    int api_ram_set_factor(mhx_run* r, const REAL* S);                                                                 \
    int api_ram_set_factor_all(mhx_run* r, const REAL* S);                                                             \
    int api_ram_get_factor(mhx_run* r, REAL* S, uint8_t* status);                                                      \
    int api_ram_get_diag_range(mhx_run* r, REAL* diag_min, REAL* diag_max);                                            \
    int api_ram_get_adapt_state(mhx_run* r, REAL* log_alpha, double* eta, uint8_t* isaccept, uint64_t* iteration);     \
    int api_ram_get_step_stats(mhx_run* r, REAL* log_alpha, double* eta, long capacity, long* n_recorded);             \
    int api_ram_watch_factors(mhx_run* r, const int32_t* chains, int n);                                               \
    int api_ram_get_watched_factors(mhx_run* r, REAL* S, long capacity, long* n_recorded, int* n_watched);             \
    int api_run_init(mhx_run* r, const REAL* initial_params);                                                          \
    int api_run_sample(mhx_run* r, const mhx_schedule* s, int save_samples);                                           \
    int api_run_get_samples(mhx_run* r, REAL* samples, uint8_t* accepted);                                             \
    int api_run_sample_to_host(mhx_run* r, const mhx_schedule* s, REAL* samples, uint8_t* accepted, int slab_samples);  \
    int api_run_device_samples(mhx_run* r, void** samples, void** accepted, int64_t* n_samples);                       \
    int api_run_get_state(mhx_run* r, REAL* x, REAL* lp, uint32_t* accept_counts);                                     \
    int api_run_set_state(mhx_run* r, const REAL* x);                                                                  \
    int api_run_state_size(mhx_run* r, size_t* bytes);                                                                 \
    int api_run_save_state(mhx_run* r, void* blob, size_t bytes);                                                      \
    int api_run_load_state(mhx_run* r, const void* blob, size_t bytes);                                                \
    int api_run_stats(mhx_run* r, mhx_stats* out);                                                                     \
    int api_run_host_stats(mhx_run* r, mhx_host_stats* out);                                                           \
    int api_run_destroy(mhx_run* r);                                                                                   \
    int api_run_diagnostics(mhx_run* r, const mhx_diag_cfg* cfg, double* sum_m, double* sum_m2, double* sum_v,         \
                            double* ess);                                                                              \
    int api_run_ess_bulk_tail(mhx_run* r, const mhx_diag_cfg* cfg, const int32_t* params, int32_t nparams,             \
                              double* ess_bulk, double* ess_tail);                                                     \
    int api_emcee_half_step(mhx_run* r, int half, int begin, int count);                                               \
    int api_emcee_end_sweep(mhx_run* r);                                                                               \
    int api_emcee_device_state(mhx_run* r, REAL** xw, int32_t* pitch, REAL** lp, uint32_t** acc_count,                 \
                               uint8_t** last_acc);                                                                    \
    int api_emcee_exchange_plan(mhx_run* r, int half, int world, size_t* stride, void** stream);                       \
    int api_emcee_exchange_pack(mhx_run* r, int half, int rank, int world, void* part);                                \
    int api_emcee_exchange_unpack(mhx_run* r, int half, int rank, int world, const void* stage, size_t stride);        \
    }

MHX_IMPL_DECLARE(mhx_f32, float)
MHX_IMPL_DECLARE(mhx_f64, double)
