// mhx_diag_kernels.h -- chain diagnostics on the device sample tensor [N][dim+1][C] (chain fastest).
//
// What MCMCChains prints for the reference (README.md:59-63: mean, ESS, R-hat) needs, per parameter:
//   m_c, s2_c          per-chain mean and (unbiased) variance over the N draws
//   sum_c m_c, sum_c m_c^2, sum_c s2_c   -> between/within variances, R-hat, between-chain ESS
//   chain-averaged autocovariances up to max_lag over a subset of chains -> Geyer ESS
// All reads are coalesced over the chain index; accumulation is fp64.
#pragma once
#include "mhx_device_math.h"

MHX_NS_BEGIN

// grid (ceil(C/256), dim+1): one thread per (chain, parameter)
MHX_DEV void mhx_diag_moments_body(const mhx_real* __restrict__ samples, const long N, const int d1, const long C,
                                   double* __restrict__ mean, double* __restrict__ sums /* [3][d1] */, double* red)
{
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    double m = 0.0, v = 0.0;
    if (c < C) {
        const mhx_real* s = samples + (long)p * C + c;
        const long stride = (long)d1 * C;
        double sum = 0.0;
        for (long t = 0; t < N; ++t) sum += (double)s[t * stride];
        m = sum / (double)N;
        double ss = 0.0;
        for (long t = 0; t < N; ++t) { const double e = (double)s[t * stride] - m; ss += e * e; }
        v = N > 1 ? ss / (double)(N - 1) : 0.0;
        mean[(long)p * C + c] = m;
    }
    // block reduction of (m, m^2, v) then one fp64 atomic each
    double vals[3] = {c < C ? m : 0.0, c < C ? m * m : 0.0, c < C ? v : 0.0};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double x = vals[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0) red[q * 4 + (threadIdx.x >> 6)] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double x = red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1] + red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3];
        atomicAdd(&sums[(long)threadIdx.x * d1 + p], x);
    }
}

// grid (ceil(nc/64), dim+1, max_lag+1), block 64: acov[k][p] += sum_{c<nc} sum_t (x_t-m_c)(x_{t+k}-m_c)
MHX_DEV void mhx_diag_autocov_body(const mhx_real* __restrict__ samples, const long N, const int d1, const long C,
                                   const long nc, const double* __restrict__ mean, double* __restrict__ acov)
{
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    const long k = blockIdx.z;
    double acc = 0.0;
    if (c < nc && k < N) {
        const mhx_real* s = samples + (long)p * C + c;
        const long stride = (long)d1 * C;
        const mhx_real m = (mhx_real)mean[(long)p * C + c];
        mhx_real a0 = MHX_R(0.0), a1 = MHX_R(0.0);
        long t = 0;
        for (; t + 1 < N - k; t += 2) {
            a0 = mhx_fma(s[t * stride] - m, s[(t + k) * stride] - m, a0);
            a1 = mhx_fma(s[(t + 1) * stride] - m, s[(t + 1 + k) * stride] - m, a1);
            if ((t & 1023) == 1022) { acc += (double)a0 + (double)a1; a0 = a1 = MHX_R(0.0); }
        }
        for (; t < N - k; ++t) a0 = mhx_fma(s[t * stride] - m, s[(t + k) * stride] - m, a0);
        acc += (double)a0 + (double)a1;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (threadIdx.x == 0) atomicAdd(&acov[k * d1 + p], acc);
}

// the same three sums from running moments (runs that kept no sample tensor): m_c = mean, s2_c = M2/(n-1)
MHX_DEV void mhx_diag_from_moments_body(const mhx_real* __restrict__ mom_mean, const mhx_real* __restrict__ mom_m2,
                                        const long nsamp, const int d1, const long C, double* __restrict__ sums,
                                        double* red)
{
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int p = blockIdx.y;
    double m = 0.0, v = 0.0;
    if (c < C) {
        m = (double)mom_mean[(long)p * C + c];
        v = nsamp > 1 ? (double)mom_m2[(long)p * C + c] / (double)(nsamp - 1) : 0.0;
    }
    double vals[3] = {c < C ? m : 0.0, c < C ? m * m : 0.0, c < C ? v : 0.0};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double x = vals[q];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0) red[q * 4 + (threadIdx.x >> 6)] = x;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const double x = red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1] + red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3];
        atomicAdd(&sums[(long)threadIdx.x * d1 + p], x);
    }
}
MHX_NS_END
