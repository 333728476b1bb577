// Moment-function evaluation and the fused level-difference accumulation kernel (gfx950).
//
// Hot loop replaced (reference, one chunk of one level; quantity_estimate.py:43-65):
//   moments_fn.eval_all(x)           -> [M, n, 2, R] temporaries (legvander: R ufunc passes)
//   transpose/reshape                -> copy
//   mask_nan_samples                 -> isnan pass over the expanded array + gather copy
//   chunk[:, :, 0] - chunk[:, :, 1]  -> temporary
//   np.sum(diff), np.sum(diff ** 2)  -> two more passes
// Here: one pass over the raw fine/coarse arrays (16 B per sample pair, coalesced 8-B loads), the R-term
// recurrences for fine and coarse kept in registers, per-lane fp64 accumulators for sum(d) and sum(d^2),
// wave shuffles + LDS for the block partial, and a fixed-order second kernel for the grid reduction
// (bitwise reproducible run to run).
#include <cstdlib>
#include <cstring>

#include "device_basis.hpp"

namespace mlmc {

// ------------------------------------------------------------------------------------------
// eval_all: out[i][r] (Moments.eval_all, moments.py:90-93) -- not a hot kernel; used by the PDF solver's
// callers, plots and tests.  One thread per value.
// ------------------------------------------------------------------------------------------
template <int KIND>
__global__ void k_eval(BasisParams bp, const double *__restrict__ scale_c,
                       const double *__restrict__ x, int64_t n, int size, double *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool keep;
    double t = transform_value(bp, x[i], keep);
    TermGen<KIND> g;
    g.init(keep ? t : 0.0, 1.0, bp);
    const double nan = __builtin_nan("");
    for (int r = 0; r < size; ++r) {
        double q = g.next(r);
        if (KIND == MLMC_LEGENDRE && scale_c) q *= scale_c[r];     // scale_c == nullptr: the scaled q_i the accumulators sum
        out[i * size + r] = keep ? q : nan;
    }
}

// out[i][j] = sum_r phi[i][r] * T[j][r]   (TransformedMoments._eval_all, moments.py:256-259)
__global__ void k_apply_matrix(const double *__restrict__ phi, const double *__restrict__ T, int64_t n, int R0, int size,
                               double *__restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * size) return;
    int64_t i = idx / size;
    int j = (int)(idx % size);
    double acc = 0.0;
    for (int r = 0; r < R0; ++r) acc = __builtin_fma(phi[i * R0 + r], T[(int64_t)j * R0 + r], acc);
    out[idx] = acc;
}

// The same product on the matrix cores (n >= 64): out = phi T^T as a GEMM with M = n, N = size, K = R0.  A workgroup takes 64
// value rows, wave w the rows [16 w, 16 w + 16); columns in groups of 64 (four 16 x 16 tiles per wave), k in steps of 4:
// operand lane (l & 15, l >> 4) = (row | column, k) for both A = phi and B = T, result lane holds rows (l >> 4) + 4 r, column
// l & 15.  T (at most a few MB) and the 16 rows of phi a wave walks are served by the caches.  As a scalar loop per output
// element (above) the product of 1.2e7 x 64 values with a 49 x 64 matrix took 29 ms per call -- 90 % of a covariance estimate
// of transformed moments; this form takes ~1 ms.
__global__ __launch_bounds__(256) void k_apply_matrix_mfma(const double *__restrict__ phi, const double *__restrict__ T, int64_t n,
                                                            int R0, int size, double *__restrict__ out) {
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * 64 + 16 * wave;
    if (i0 >= n) return;
    const int64_t ia = i0 + (lane & 15) < n ? i0 + (lane & 15) : n - 1;         // clamped row of the A operand
    const int kq = lane >> 4;
    const double *__restrict__ arow = phi + ia * (int64_t)R0;
    for (int j0 = 0; j0 < size; j0 += 64) {
        v4f64 acc[4];
        const double *brow[4];
        bool bok[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            acc[t] = (v4f64){0.0, 0.0, 0.0, 0.0};
            const int j = j0 + 16 * t + (lane & 15);
            bok[t] = j < size;
            brow[t] = T + (int64_t)(bok[t] ? j : 0) * R0;
        }
        for (int k0 = 0; k0 < R0; k0 += 4) {
            const int k = k0 + kq;
            const bool kin = k < R0;
            const double a = kin ? arow[k] : 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double b = (kin && bok[t]) ? brow[t][k] : 0.0;
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int j = j0 + 16 * t + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = i0 + (lane >> 4) + 4 * r;
                if (i < n && j < size) out[i * (int64_t)size + j] = acc[t][r];
            }
        }
    }
}

// Values of the UNDERLYING family in the accumulators' scaling (Legendre: q_i = P_i / c_i), [n][b->p.size]: the input of the
// difference Gram matrix of TransformedMoments over more than 128 underlying moments (launch_cov_from_values, gram_mode 1).
int launch_eval_scaled_base(const mlmc_basis *b, const double *d_x, int64_t n, double *d_out) {
    if (n == 0) return 0;
    hipStream_t st = rt().stream;
    const int threads = 256;
    const int64_t blocks = (n + threads - 1) / threads;
    const BasisParams &bp = b->p;
    const double *no_scale = nullptr;
    switch (bp.kind) {
        case MLMC_LEGENDRE: hipLaunchKernelGGL(k_eval<MLMC_LEGENDRE>, dim3(blocks), dim3(threads), 0, st, bp, no_scale, d_x, n, bp.size, d_out); break;
        case MLMC_MONOMIAL: hipLaunchKernelGGL(k_eval<MLMC_MONOMIAL>, dim3(blocks), dim3(threads), 0, st, bp, no_scale, d_x, n, bp.size, d_out); break;
        case MLMC_FOURIER: hipLaunchKernelGGL(k_eval<MLMC_FOURIER>, dim3(blocks), dim3(threads), 0, st, bp, no_scale, d_x, n, bp.size, d_out); break;
        case MLMC_SPLINE: hipLaunchKernelGGL(k_eval<MLMC_SPLINE>, dim3(blocks), dim3(threads), 0, st, bp, no_scale, d_x, n, bp.size, d_out); break;
        default: return fail("unknown basis kind");
    }
    MLMC_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_eval(const mlmc_basis *b, const double *d_x, int64_t n, int size, double *d_out, double *scratch) {
    if (n == 0) return 0;
    hipStream_t st = rt().stream;
    int threads = 256;
    int64_t blocks = (n + threads - 1) / threads;
    const BasisParams &bp = b->p;
    double *target = d_out;
    double *d_tmp = nullptr;
    int esize = size;
    if (b->out_size > 0) {   // evaluate all R0 underlying terms, then apply the matrix
        esize = bp.size;
        if (scratch) d_tmp = scratch;   // caller's workspace of n * (underlying size) doubles: no allocation, no sync
        else MLMC_HIP_CHECK(hipMalloc(&d_tmp, sizeof(double) * (size_t)n * esize));
        target = d_tmp;
    }
    switch (bp.kind) {
        case MLMC_LEGENDRE: hipLaunchKernelGGL(k_eval<MLMC_LEGENDRE>, dim3(blocks), dim3(threads), 0, st, bp, b->d_scale, d_x, n, esize, target); break;
        case MLMC_MONOMIAL: hipLaunchKernelGGL(k_eval<MLMC_MONOMIAL>, dim3(blocks), dim3(threads), 0, st, bp, b->d_scale, d_x, n, esize, target); break;
        case MLMC_FOURIER: hipLaunchKernelGGL(k_eval<MLMC_FOURIER>, dim3(blocks), dim3(threads), 0, st, bp, b->d_scale, d_x, n, esize, target); break;
        case MLMC_IDENTITY: hipLaunchKernelGGL(k_eval<MLMC_IDENTITY>, dim3(blocks), dim3(threads), 0, st, bp, b->d_scale, d_x, n, esize, target); break;
        case MLMC_SPLINE: hipLaunchKernelGGL(k_eval<MLMC_SPLINE>, dim3(blocks), dim3(threads), 0, st, bp, b->d_scale, d_x, n, esize, target); break;
        default: return fail("unknown basis kind");
    }
    MLMC_HIP_CHECK(hipGetLastError());
    if (b->out_size > 0) {
        int64_t tot = n * size;
        if (n >= 64)
            hipLaunchKernelGGL(k_apply_matrix_mfma, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, d_tmp, b->d_matrix, n, bp.size, size, d_out);
        else
            hipLaunchKernelGGL(k_apply_matrix, dim3((tot + 255) / 256), dim3(256), 0, st, d_tmp, b->d_matrix, n, bp.size, size, d_out);
        MLMC_HIP_CHECK(hipGetLastError());
        if (!scratch) {
            MLMC_HIP_CHECK(wait_stream(st));
            MLMC_HIP_CHECK(hipFree(d_tmp));
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// mask kernel for quantities with M > 1 components: a sample is dropped when ANY component of fine or
// coarse is masked (mask_nan_samples reduces over axis 0, quantity_estimate.py:12).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mask(BasisParams bp, const double *__restrict__ f, const double *__restrict__ c, int64_t n,
                                              int n_comp, uint8_t *__restrict__ mask, int64_t *__restrict__ counts) {
    __shared__ int red[4][2];
    int kept = 0, removed = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        bool keep = true;
        for (int m = 0; m < n_comp; ++m) {
            bool k1, k2 = true;
            transform_value(bp, f[(int64_t)m * n + i], k1);
            if (c) transform_value(bp, c[(int64_t)m * n + i], k2);
            keep = keep && k1 && k2;
        }
        mask[i] = keep ? 1 : 0;
        kept += keep;
        removed += !keep;
    }
    kept = wave_sum_i(kept);
    removed = wave_sum_i(removed);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = kept; red[threadIdx.x >> 6][1] = removed; }
    __syncthreads();
    if (threadIdx.x < 2) {   // one pair of integer atomics per block (exact, order independent); same-address atomics
        // serialise in L2 at ~12 ns each, so they are kept to a few thousand per launch
        const int v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd((unsigned long long *)&counts[threadIdx.x], (unsigned long long)v);
    }
}

int launch_mask(const mlmc_accum *a, const double *d_f, const double *d_c, int64_t n, uint8_t *d_mask, int64_t *d_counts_level) {
    if (n == 0) return 0;
    const int64_t want = (n + 255) / 256;
    hipLaunchKernelGGL(k_mask, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, rt().stream, a->basis->p, d_f, d_c, n,
                       a->n_comp, d_mask, d_counts_level);
    MLMC_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// The accumulation kernel.
//   RT    : accumulators per lane for sum(d) and sum(d^2): terms [t0, t0 + RT) of this pass (terms >= R, if any,
//           are computed and discarded, so the loop body is one branch-free basic block)
//   T0C   : first term of the pass when known at compile time (0: first pass, 64: second pass of 64 < R <= 128 -- every
//           term index and recurrence coefficient is then a constant); -1: t0 is a run-time argument (R > 128)
// One launch covers up to MAX_SEG segments (= pushed chunks of different levels): every block belongs to one
// segment and the segments get blocks in proportion to their work, so a whole multi-level estimate is ONE grid
// with one ramp-up and one tail.  Within a segment each lane walks the samples with a stride, two samples per trip
// (four independent recurrences for a fine/coarse pair) so that the fp64 pipe always has independent work; the next
// trip's loads are issued before the current trip's arithmetic.  Per pair and term: mul + fma (fine), mul + fma
// (coarse), sub, add, fma = 7 fp64 instructions (level 0: 4).
// ------------------------------------------------------------------------------------------
#ifdef MLMC_PROF
__device__ unsigned long long *g_prof;   // tools/prof_moments.hip
#endif
constexpr int MAX_SEG = 16;
constexpr int PRIO_SLICE_BITS = 16;   // 65536 cycles = 27 us at 2.4 GHz (A/B of 13..17 on one box: 16 and 17 best by ~1 %)
struct Seg {
    const double *fine, *coarse;   // coarse == nullptr: level 0
    const uint8_t *mask;           // optional keep flags (quantities with M > 1 components)
    int64_t n;
    int block0, nblocks;           // blocks [block0, block0 + nblocks) of the grid work on this segment
};
struct SegTable {
    int nseg;
    int prio_mod;                  // waves per SIMD (= resident blocks per CU) that share the issue priority in turn
    Seg seg[MAX_SEG];
};
struct ReduceTarget {
    double *totals;                // [2][int_R] of this segment's (level, component)
    int64_t *counts;               // (kept, removed) of the level; nullptr: do not count
    // optional finished outputs in host-mapped pinned memory (plain bases, one component): the thread that updates a
    // total also writes the caller-visible value, so finalize needs neither a finalize kernel nor a D2H copy
    double *h_s, *h_sp;            // [R] rows of this level in the pinned mirror; nullptr: not used
    int64_t *h_n;                  // &n[level]; n_rm[level] is h_n[h_n_stride]
    double *p_nd;                  // optional second copy for the packed all-reduce buffer: counts as doubles,
    double *p_s, *p_sp;            //   rows of this level (device memory of the caller); nullptr: not used
    const double *scale;           // P_i = scale[i] * Q_i
    int64_t h_n_stride;
};
struct ReduceTable {
    ReduceTarget t[MAX_SEG];
};

template <int KIND, int RT, bool PAIR, int T0C, bool PLAIN>
__device__ __forceinline__ void accum_samples(const BasisParams &bp, 
                                              const double *__restrict__ fine, const double *__restrict__ coarse,
                                              const uint8_t *__restrict__ mask, int64_t n, int t0, int bid, int nb,
                                              unsigned prio_mod, double (&s)[RT], double (&sp)[RT], int &n_keep, int &n_rm) {
    const int64_t T = (int64_t)nb * ACC_THREADS;
    const int64_t gtid = (int64_t)bid * ACC_THREADS + threadIdx.x;

    int64_t i0 = gtid, i1 = gtid + T;
    double f0 = 0, f1 = 0, c0 = 0, c1 = 0;
    uint8_t m0 = 1, m1 = 1;
    // PLAIN: the common case (linear transform, clipping to the domain, no keep flags from a mask kernel) without the
    // run-time switches of the general path -- at small R the scalar branches of those switches cost as much as the terms
    if (i0 < n) { f0 = fine[i0]; if (PAIR) c0 = coarse[i0]; if (!PLAIN && mask) m0 = mask[i0]; }
    if (i1 < n) { f1 = fine[i1]; if (PAIR) c1 = coarse[i1]; if (!PLAIN && mask) m1 = mask[i1]; }

    // Issue-priority time slicing.  The kernel runs two or three waves per SIMD (VGPR bound) and the SIMD arbiter serves
    // the older wave first: with equal priorities the first-dispatched wave of each SIMD runs at its single-wave rate,
    // finishes after ~60 % of the kernel and leaves its neighbours to run alone -- and a lone wave cannot fill the
    // fp64 pipe (measured 6.1 vs 4.4 cycles per instruction with two).  So the prio_mod waves of a SIMD (wave slot
    // modulo prio_mod = resident blocks per CU) take turns at the higher priority in slices of 2^PRIO_SLICE_BITS
    // shader cycles read from the shared clock: all progress at the same average rate and end together
    // (-4 % kernel time at R = 32).
    const unsigned prio_phase = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 15u) % prio_mod;   // HW_ID.wave_id
    unsigned long long prio_clock = __builtin_amdgcn_s_memtime();
    while (i0 < n) {
        if (((unsigned)(prio_clock >> PRIO_SLICE_BITS)) % prio_mod == prio_phase) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);
        prio_clock = __builtin_amdgcn_s_memtime();   // read now, used at the next trip: the latency is hidden
        const bool v1 = i1 < n;
        const double xf0 = f0, xf1 = f1, xc0 = c0, xc1 = c1;
        const uint8_t mm0 = m0, mm1 = m1;
        // prefetch the next trip
        const int64_t j0 = i0 + 2 * T, j1 = i1 + 2 * T;
        if (j0 < n) { f0 = fine[j0]; if (PAIR) c0 = coarse[j0]; if (!PLAIN && mask) m0 = mask[j0]; }
        if (j1 < n) { f1 = fine[j1]; if (PAIR) c1 = coarse[j1]; if (!PLAIN && mask) m1 = mask[j1]; }

        bool kf0, kf1, kc0 = true, kc1 = true;
        const double tf0 = PLAIN ? transform_plain(bp, xf0, kf0) : transform_value(bp, xf0, kf0);
        const double tf1 = PLAIN ? transform_plain(bp, xf1, kf1) : transform_value(bp, xf1, kf1);
        double tc0 = 0, tc1 = 0;
        if (PAIR) {
            tc0 = PLAIN ? transform_plain(bp, xc0, kc0) : transform_value(bp, xc0, kc0);
            tc1 = PLAIN ? transform_plain(bp, xc1, kc1) : transform_value(bp, xc1, kc1);
        }
        const bool k0 = kf0 && kc0 && (PLAIN || mm0 != 0);
        const bool k1 = v1 && kf1 && kc1 && (PLAIN || mm1 != 0);
        n_keep += (int)k0 + (int)k1;
        n_rm += (int)(!k0) + (int)(v1 && !k1);
        const double w0 = k0 ? 1.0 : 0.0, w1 = k1 ? 1.0 : 0.0;

        TermGen<KIND> gf0, gf1, gc0, gc1;
        gf0.init(k0 ? tf0 : 0.0, w0, bp);
        gf1.init(k1 ? tf1 : 0.0, w1, bp);
        if (PAIR) { gc0.init(k0 ? tc0 : 0.0, w0, bp); gc1.init(k1 ? tc1 : 0.0, w1, bp); }

        if constexpr (T0C > 0) {   // second pass of 64 < R <= 128: advance the recurrences without accumulating, unrolled
#pragma unroll
            for (int i = 0; i < T0C; ++i) {
                gf0.next(i); gf1.next(i);
                if (PAIR) { gc0.next(i); gc1.next(i); }
            }
        } else if constexpr (T0C < 0) {   // R > 128: run-time term window, t0 = 128, 192, ...: the first 128 steps with compile-time indices,
                                // the rest in blocks of 64 whose coefficients are fetched ahead of the dependent steps (one
                                // load per step inside the recurrence chain cost 20 x the step itself)
#pragma unroll
            for (int i = 0; i < 128; ++i) {
                gf0.next(i); gf1.next(i);
                if (PAIR) { gc0.next(i); gc1.next(i); }
            }
            for (int b = 128; b < t0; b += 64) {
                double gb[64];
#pragma unroll
                for (int j = 0; j < 64; ++j) gb[j] = KIND == MLMC_LEGENDRE ? kLegendreG.v[b + j] : 0.0;
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    gf0.skip(j, gb[j]); gf1.skip(j, gb[j]);
                    if (PAIR) { gc0.skip(j, gb[j]); gc1.skip(j, gb[j]); }
                }
            }
        }
        if constexpr (T0C < 0) {
            // run-time window: the coefficients of 16 terms are fetched (uniform address) ahead of their dependent steps
            static_assert(RT % 16 == 0, "run-time term windows use 64-term tiles");
#pragma unroll
            for (int i0 = 0; i0 < RT; i0 += 16) {
                double ga[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) ga[j] = KIND == MLMC_LEGENDRE ? kLegendreG.v[t0 + i0 + j] : 0.0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int i = i0 + j;
                    double d0 = gf0.skip(i, ga[j]);
                    double d1 = gf1.skip(i, ga[j]);
                    if (PAIR) { d0 -= gc0.skip(i, ga[j]); d1 -= gc1.skip(i, ga[j]); }
                    s[i] += d0;
                    sp[i] = __builtin_fma(d0, d0, sp[i]);
                    s[i] += d1;
                    sp[i] = __builtin_fma(d1, d1, sp[i]);
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            double d0 = gf0.next(t0 + i);
            double d1 = gf1.next(t0 + i);
            if (PAIR) { d0 -= gc0.next(t0 + i); d1 -= gc1.next(t0 + i); }
            s[i] += d0;
            sp[i] = __builtin_fma(d0, d0, sp[i]);
            s[i] += d1;
            sp[i] = __builtin_fma(d1, d1, sp[i]);
        }
        }
        i0 = j0;
        i1 = j1;
    }
}

// Small tiles (RT <= 16, the reference's everyday n_moments = 5..10): the work per sample is a few dozen instructions and
// the kernel sits between the HBM and the issue roof, so what matters is memory-level parallelism -- four samples per
// lane per trip (eight 8-byte loads in flight, issued a whole trip ahead) instead of two.  Common configuration only.
template <int KIND, int RT, bool PAIR>
__device__ __forceinline__ void accum_samples_wide(const BasisParams &bp, const double *__restrict__ fine,
                                                   const double *__restrict__ coarse, int64_t n, int bid, int nb,
                                                   double (&s)[RT], double (&sp)[RT], int &n_keep, int &n_rm) {
    constexpr int S = 4;
    const int64_t T = (int64_t)nb * ACC_THREADS;
    int64_t i = (int64_t)bid * ACC_THREADS + threadIdx.x;
    double f[S], c[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        const int64_t ik = i + k * T;
        f[k] = 0.0;
        c[k] = 0.0;
        if (ik < n) { f[k] = fine[ik]; if (PAIR) c[k] = coarse[ik]; }
    }
    while (i < n) {
        double xf[S], xc[S];
        bool valid[S];
#pragma unroll
        for (int k = 0; k < S; ++k) { xf[k] = f[k]; xc[k] = c[k]; valid[k] = i + k * T < n; }
        const int64_t j = i + S * T;
#pragma unroll
        for (int k = 0; k < S; ++k) {   // next trip's loads
            const int64_t jk = j + k * T;
            if (jk < n) { f[k] = fine[jk]; if (PAIR) c[k] = coarse[jk]; }
        }
        TermGen<KIND> gf[S], gc[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            bool kf, kc = true;
            const double tf = transform_plain(bp, xf[k], kf);
            const double tc = PAIR ? transform_plain(bp, xc[k], kc) : 0.0;
            const bool keep = valid[k] && kf && kc;
            n_keep += (int)keep;
            n_rm += (int)(valid[k] && !keep);
            const double w = keep ? 1.0 : 0.0;
            gf[k].init(keep ? tf : 0.0, w, bp);
            if (PAIR) gc[k].init(keep ? tc : 0.0, w, bp);
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
#pragma unroll
            for (int k = 0; k < S; ++k) {
                double d = gf[k].next(t);
                if (PAIR) d -= gc[k].next(t);
                s[t] += d;
                sp[t] = __builtin_fma(d, d, sp[t]);
            }
        }
        i = j;
    }
}

// PLAIN kernels hold only the switch-free loops (fewer registers: a third wave per SIMD up to RT = 28); the general
// kernels only the loops with the run-time switches.  The host launches PLAIN when the basis is in the common
// configuration and no segment carries a mask.
template <int KIND, int RT, int T0C, bool PLAIN>
__global__ __launch_bounds__(ACC_THREADS) void k_moments_accum(BasisParams bp, SegTable tab,
                                                              int t0_arg, double *__restrict__ partials,
                                                              int64_t *__restrict__ pcounts) {
    const int t0 = T0C >= 0 ? T0C : t0_arg;
    // segment of this block: static indices only, so the table stays in scalar registers
    Seg sg = tab.seg[0];
#pragma unroll
    for (int k = 1; k < MAX_SEG; ++k)
        if (k < tab.nseg && (int)blockIdx.x >= tab.seg[k].block0) sg = tab.seg[k];
    const int bid = (int)blockIdx.x - sg.block0;
#ifdef MLMC_PROF
    const unsigned long long prof_r0 = __builtin_amdgcn_s_memrealtime(), prof_c0 = __builtin_amdgcn_s_memtime();
#endif

    double s[RT], sp[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) { s[i] = 0.0; sp[i] = 0.0; }
    int n_keep = 0, n_rm = 0;
    const unsigned prio_mod = tab.prio_mod > 0 ? (unsigned)tab.prio_mod : 1u;
    constexpr bool WIDE = RT <= 16 && T0C == 0;
    if (PLAIN && WIDE) {
        if (sg.coarse) accum_samples_wide<KIND, RT, true>(bp, sg.fine, sg.coarse, sg.n, bid, sg.nblocks, s, sp, n_keep, n_rm);
        else accum_samples_wide<KIND, RT, false>(bp, sg.fine, sg.coarse, sg.n, bid, sg.nblocks, s, sp, n_keep, n_rm);
    } else if (sg.coarse) {
        accum_samples<KIND, RT, true, T0C, PLAIN>(bp, sg.fine, sg.coarse, sg.mask, sg.n, t0, bid, sg.nblocks, prio_mod, s, sp, n_keep, n_rm);
    } else {
        accum_samples<KIND, RT, false, T0C, PLAIN>(bp, sg.fine, sg.coarse, sg.mask, sg.n, t0, bid, sg.nblocks, prio_mod, s, sp, n_keep, n_rm);
    }
#ifdef MLMC_PROF
    const unsigned long long prof_r1 = __builtin_amdgcn_s_memrealtime(), prof_c1 = __builtin_amdgcn_s_memtime();
#endif
    // ---- block partial: the four waves add their lanes' accumulators into one LDS image [value][lane] in a
    // fixed order, then one thread per value sums the 64 lanes (row stride 65: conflict-free both ways) ----
    __shared__ double red[2 * RT][WAVE + 1];
    __shared__ int ldc[4][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
            if (w == 0) {
#pragma unroll
                for (int i = 0; i < RT; ++i) { red[i][lane] = s[i]; red[RT + i][lane] = sp[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < RT; ++i) { red[i][lane] += s[i]; red[RT + i][lane] += sp[i]; }
            }
        }
        __syncthreads();
    }
    n_keep = wave_sum_i(n_keep);
    n_rm = wave_sum_i(n_rm);
    if (lane == 0) { ldc[wave][0] = n_keep; ldc[wave][1] = n_rm; }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * RT; j += ACC_THREADS) {
        double v = 0.0;
#pragma unroll 8
        for (int l = 0; l < WAVE; ++l) v += red[j][l];
        partials[(int64_t)blockIdx.x * (2 * RT) + j] = v;
    }
    if (threadIdx.x < 2) {
        int v = ldc[0][threadIdx.x] + ldc[1][threadIdx.x] + ldc[2][threadIdx.x] + ldc[3][threadIdx.x];
        pcounts[(int64_t)blockIdx.x * 2 + threadIdx.x] = v;
    }
#ifdef MLMC_PROF
    if ((threadIdx.x & 63) == 0) {   // per wave: start, loop end, block end (100 MHz ticks); shader cycles of the loop; XCC/CU id
        unsigned long long *p = g_prof + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 5;
        p[0] = prof_r0; p[1] = prof_r1; p[2] = __builtin_amdgcn_s_memrealtime(); p[3] = prof_c1 - prof_c0;
        p[4] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |   // HW_ID
               ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);   // XCC_ID
    }
#endif
}

// ------------------------------------------------------------------------------------------
// 48 < R <= 64 in ONE pass over the samples at two waves per SIMD (Legendre, monomials).
// A 64-term register tile needs 256 accumulator VGPRs (one wave per SIMD, and a lone wave cannot fill the fp64 pipe); two
// 32-term passes re-run the first 32 recurrence steps in the second pass (+29 % instructions).  Here the four waves of a
// workgroup split the TERMS of the same samples: waves 0-1 ("head") own terms [0, SPLIT_HEAD) -- loads, transform, keep
// flags, counts, the first recurrence steps -- and hand the state of every recurrence (x and the last two terms; masked
// samples: zeros) to waves 2-3 ("tail") through a double-buffered LDS image; the tail continues the same recurrences through
// terms [SPLIT_HEAD, 64) of the previous trip's samples while the head works on the next trip.  One workgroup barrier per trip (two
// samples per head lane).  Every recurrence step runs exactly once: the instruction count is that of a single
// 64-term pass, the occupancy that of the 32-term tile.  Partial rows [2][64] as a 64-term tile would write them.
// ------------------------------------------------------------------------------------------
constexpr int SPLIT_LANES = 128;                 // sample lanes per workgroup (head lanes = tail lanes)
// terms of the head / the tail: the head also loads, transforms and counts (about 50 instructions per trip), the tail only
// reads the hand-over, so the tail takes four terms more -- 30 / 34 leaves both sides the same instruction count per trip
// (with 32 / 32 the tail waited at the barrier: SQ_WAIT_ANY 19 % of the wave cycles)
#ifndef MLMC_SPLIT_HEAD
#define MLMC_SPLIT_HEAD 30
#endif
constexpr int SPLIT_HEAD = MLMC_SPLIT_HEAD;      // of 64 terms (48 < R <= 64)
// (Measured and not adopted: the same split at 32 terms, 14 / 18, 106 VGPRs = four waves per SIMD, for 24 < R <= 32 -- the
// hand-over and the barrier per trip cost more than the extra occupancy brings: 0.224 ms against 0.206 ms for the one-pass
// kernel on BASELINE configs[1], same-box A/B.)
__host__ __device__ constexpr int split_max(int ht, int tt) { return ht > tt ? ht : tt; }
// the two workgroups of a CU take turns at the higher issue priority, like the waves of k_moments_accum (see accum_samples)
#ifndef MLMC_SPLIT_WPS
#define MLMC_SPLIT_WPS 2      // workgroups per CU (= waves per SIMD) the split kernel's registers are held to (3: +4 % time)
#endif
#ifdef MLMC_SPLIT_NO_PRIO
#define MLMC_SPLIT_PRIO_INIT
#define MLMC_SPLIT_PRIO_TRIP
#else
#ifndef MLMC_SPLIT_PRIO_BITS
#define MLMC_SPLIT_PRIO_BITS 16
#endif
#define MLMC_SPLIT_PRIO_INIT                                                                                              \
    const unsigned prio_phase = (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 15u) % MLMC_SPLIT_WPS; /* HW_ID.wave_id */ \
    unsigned long long prio_clock = __builtin_amdgcn_s_memtime();
#define MLMC_SPLIT_PRIO_TRIP                                                                                              \
    if ((((unsigned)(prio_clock >> MLMC_SPLIT_PRIO_BITS)) % MLMC_SPLIT_WPS) == prio_phase) __builtin_amdgcn_s_setprio(2); \
    else __builtin_amdgcn_s_setprio(0);                                                                                   \
    prio_clock = __builtin_amdgcn_s_memtime();
#endif

template <int KIND>
__device__ __forceinline__ void split_export(const TermGen<KIND> &g, double *__restrict__ slot) {
    slot[0 * SPLIT_LANES] = g.x;
    slot[1 * SPLIT_LANES] = g.p1;
    if (KIND == MLMC_LEGENDRE) slot[2 * SPLIT_LANES] = g.p2;
}
template <int KIND>
__device__ __forceinline__ void split_import(TermGen<KIND> &g, const double *__restrict__ slot) {
    g.x = slot[0 * SPLIT_LANES];
    g.p1 = slot[1 * SPLIT_LANES];
    g.p2 = KIND == MLMC_LEGENDRE ? slot[2 * SPLIT_LANES] : 0.0;
    g.c1 = g.s1 = 0.0;
}

// NS = samples per head lane and trip (one workgroup barrier per trip).  Hand-over image per buffer: [NS * (PAIR ? 2 : 1)]
// recurrences x 3 words x SPLIT_LANES lanes; the kernel sizes it for pairs.
#ifndef MLMC_SPLIT_NS
#define MLMC_SPLIT_NS 2
#endif
template <int KIND, bool PAIR, bool PLAIN, int HT, int TT, bool SQ, int NS, int T0 = 0>
__device__ __forceinline__ void split_head(const BasisParams &bp, const double *__restrict__ fine, const double *__restrict__ coarse,
                                           const uint8_t *__restrict__ mask, int64_t n, int bid, int nb, int n_trips,
                                           double *__restrict__ hand, double (&s)[split_max(HT, TT)], double (&sp)[split_max(HT, TT)],
                                           int &n_keep, int &n_rm) {
    const int64_t T = (int64_t)nb * SPLIT_LANES;
    const int l128 = threadIdx.x & (SPLIT_LANES - 1);
    int64_t idx[NS];
    double f[NS], c[NS];
    uint8_t m[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        idx[q] = (int64_t)bid * SPLIT_LANES + l128 + q * T;
        // loads are unconditional on a clamped index (n > 0 here; `valid` decides what counts): a load under a lane
        // mask is merged into the old value after an s_waitcnt right behind it, and the prefetch would hide nothing
        const int64_t j = idx[q] < n ? idx[q] : n - 1;
        f[q] = fine[j];
        c[q] = PAIR ? coarse[j] : 0.0;
        m[q] = (!PLAIN && mask) ? mask[j] : (uint8_t)1;
    }
    MLMC_SPLIT_PRIO_INIT
    for (int k = 0; k < n_trips; ++k) {
        MLMC_SPLIT_PRIO_TRIP
        bool valid[NS];
        double xf[NS], xc[NS];
        uint8_t mm[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            valid[q] = idx[q] < n;
            xf[q] = f[q]; xc[q] = c[q]; mm[q] = m[q];
            idx[q] += (int64_t)NS * T;                            // next trip's loads first (clamped, unconditional: see above)
            const int64_t j = idx[q] < n ? idx[q] : n - 1;
            f[q] = fine[j];
            if (PAIR) c[q] = coarse[j];
            if (!PLAIN && mask) m[q] = mask[j];
        }
        TermGen<KIND> gf[NS], gc[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            bool kf, kc = true;
            const double tf = PLAIN ? transform_plain(bp, xf[q], kf) : transform_value(bp, xf[q], kf);
            double tc = 0;
            if (PAIR) tc = PLAIN ? transform_plain(bp, xc[q], kc) : transform_value(bp, xc[q], kc);
            const bool keep = valid[q] && kf && kc && (PLAIN || mm[q] != 0);
            n_keep += (int)keep;
            n_rm += (int)(valid[q] && !keep);
            const double w = keep ? 1.0 : 0.0;
            gf[q].init(keep ? tf : 0.0, w, bp);
            if (PAIR) gc[q].init(keep ? tc : 0.0, w, bp);
        }
        if (T0 > 0) {      // second term window (terms [T0, T0 + HT + TT)): advance the recurrences without accumulating
#pragma unroll
            for (int i = 0; i < T0; ++i) {
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    gf[q].next(i);
                    if (PAIR) gc[q].next(i);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < HT; ++i) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                double d = gf[q].next(T0 + i);
                if (PAIR) d -= gc[q].next(T0 + i);
                s[i] += d;
                if (SQ) sp[i] = __builtin_fma(d, d, sp[i]);
            }
        }
        double *__restrict__ slot = hand + (size_t)(k & 1) * (6 * NS * SPLIT_LANES) + l128;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            split_export<KIND>(gf[q], slot + (3 * q) * SPLIT_LANES);
            if (PAIR) split_export<KIND>(gc[q], slot + (3 * (NS + q)) * SPLIT_LANES);
        }
        __syncthreads();
    }
    __syncthreads();      // the tail's last trip
}

template <int KIND, bool PAIR, int HT, int TT, bool SQ, int NS, int T0 = 0>
__device__ __forceinline__ void split_tail(const BasisParams &bp, int n_trips, const double *__restrict__ hand,
                                           double (&s)[split_max(HT, TT)], double (&sp)[split_max(HT, TT)]) {
    const int l128 = threadIdx.x & (SPLIT_LANES - 1);
    __syncthreads();      // the head's first trip
    MLMC_SPLIT_PRIO_INIT
    for (int k = 0; k < n_trips; ++k) {
        MLMC_SPLIT_PRIO_TRIP
        const double *__restrict__ slot = hand + (size_t)(k & 1) * (6 * NS * SPLIT_LANES) + l128;
        TermGen<KIND> gf[NS], gc[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            split_import<KIND>(gf[q], slot + (3 * q) * SPLIT_LANES);
            if (PAIR) split_import<KIND>(gc[q], slot + (3 * (NS + q)) * SPLIT_LANES);
        }
#pragma unroll
        for (int i = 0; i < TT; ++i) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                double d = gf[q].next(T0 + HT + i);
                if (PAIR) d -= gc[q].next(T0 + HT + i);
                s[i] += d;
                if (SQ) sp[i] = __builtin_fma(d, d, sp[i]);
            }
        }
        __syncthreads();
    }
}

// HT + TT terms; WPS = waves per SIMD the register allocation is held to.
// SQ = false: the MEAN-ONLY form -- the sums of squares are not accumulated (their partial rows are NaN, and so is
// everything derived from them).  Without the second accumulator array a wave holds twice the terms in the same
// registers, so ONE pass covers up to 128 terms (62 + 66) at 6 instead of 7 instructions per term and pair: the pass behind
// `estimate_mean(covariance(q, fn), variance=False)` for Legendre / monomial moments, whose R x R level means follow
// from the level sums of 2 R - 1 moments (mlmc_amd/linearize.py; Estimate.construct_density, estimator.py:304-331).
// T0 > 0: the window of terms [T0, T0 + HT + TT) -- the second pass of a mean-only estimate of 129..256 moments (the head
// first walks T0 recurrence steps without accumulating, so it gets the smaller share of the accumulating terms).
template <int KIND, bool PLAIN, int HT, int TT, int WPS, bool SQ = true, int NS = MLMC_SPLIT_NS, int T0 = 0>
__global__ __launch_bounds__(ACC_THREADS, WPS) void k_moments_accum_split(BasisParams bp, SegTable tab,
                                                                        double *__restrict__ partials,
                                                                        int64_t *__restrict__ pcounts) {
    __shared__ double hand[2 * 6 * NS * SPLIT_LANES];  // [buffer][recurrence (NS fine, NS coarse) x (x, Q_k-1, Q_k-2)][lane]
    constexpr int MAXT = split_max(HT, TT), NTOT = HT + TT;
    __shared__ double wsum[4][2 * MAXT];
    __shared__ int ldc[2][2];
    Seg sg = tab.seg[0];
#pragma unroll
    for (int k = 1; k < MAX_SEG; ++k)
        if (k < tab.nseg && (int)blockIdx.x >= tab.seg[k].block0) sg = tab.seg[k];
    const int bid = (int)blockIdx.x - sg.block0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // trips of this workgroup (uniform): lane 0 of trip k starts at sample bid * 128 + k * 2 T
    const int64_t T2 = NS * (int64_t)sg.nblocks * SPLIT_LANES;
    const int64_t first = (int64_t)bid * SPLIT_LANES;
    const int n_trips = first < sg.n ? (int)((sg.n - first + T2 - 1) / T2) : 0;

    double s[MAXT], sp[MAXT];
#pragma unroll
    for (int i = 0; i < MAXT; ++i) { s[i] = 0.0; sp[i] = 0.0; }
    int n_keep = 0, n_rm = 0;
    if (wave < 2) {
        if (sg.coarse) split_head<KIND, true, PLAIN, HT, TT, SQ, NS, T0>(bp, sg.fine, sg.coarse, sg.mask, sg.n, bid, sg.nblocks, n_trips, hand, s, sp, n_keep, n_rm);
        else split_head<KIND, false, PLAIN, HT, TT, SQ, NS, T0>(bp, sg.fine, sg.coarse, sg.mask, sg.n, bid, sg.nblocks, n_trips, hand, s, sp, n_keep, n_rm);
    } else {
        if (sg.coarse) split_tail<KIND, true, HT, TT, SQ, NS, T0>(bp, n_trips, hand, s, sp);
        else split_tail<KIND, false, HT, TT, SQ, NS, T0>(bp, n_trips, hand, s, sp);
    }
    // ---- block partial: butterfly sums inside every wave (fixed order), then head pair / tail pair added in fixed order ----
#pragma unroll
    for (int i = 0; i < MAXT; ++i) {
        const double a = wave_sum(s[i]), b = SQ ? wave_sum(sp[i]) : __builtin_nan("");
        if (lane == 0) { wsum[wave][i] = a; wsum[wave][MAXT + i] = b; }
    }
    n_keep = wave_sum_i(n_keep);
    n_rm = wave_sum_i(n_rm);
    if (lane == 0 && wave < 2) { ldc[wave][0] = n_keep; ldc[wave][1] = n_rm; }
    __syncthreads();
    if (threadIdx.x < 2 * NTOT) {                 // partial row [which][term], term < NTOT
        const int which = threadIdx.x / NTOT, term = threadIdx.x % NTOT;
        const int half = term < HT ? 0 : 1, i = half ? term - HT : term;
        partials[(int64_t)blockIdx.x * (2 * NTOT) + threadIdx.x] = wsum[2 * half][which * MAXT + i] + wsum[2 * half + 1][which * MAXT + i];
    }
    if (threadIdx.x < 2) pcounts[(int64_t)blockIdx.x * 2 + threadIdx.x] = ldc[0][threadIdx.x] + ldc[1][threadIdx.x];
}

// Grid reduction of one accumulation launch: block k (1024 threads) sums the partial rows of segment k in a fixed
// order (bitwise reproducible): totals[which][t0 + i] += sum_b partials[b][which * RT + i], counts += sum_b pcounts[b].
// 16 row groups x 64 columns, every thread keeps its <= 32 row loads in flight.  (Measured alternative: letting the
// last-arriving block of the accumulation kernel do this behind an agent-scope release / acquire costs ~13 us of
// kernel tail per launch -- every block's L2 write-back -- against ~4 us for this launch.)
__global__ __launch_bounds__(1024) void k_reduce_partials(const double *__restrict__ partials,
                                                         const int64_t *__restrict__ pcounts, SegTable tab, ReduceTable rtab,
                                                         int width, int RT, int t0, int int_R) {
    __shared__ double lds[16][64];
    Seg sg = tab.seg[0];
    ReduceTarget tg = rtab.t[0];
#pragma unroll
    for (int k = 1; k < MAX_SEG; ++k)
        if ((int)blockIdx.x == k) { sg = tab.seg[k]; tg = rtab.t[k]; }
    const int row0 = sg.block0, nblocks = sg.nblocks;
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int c0 = 0; c0 < width; c0 += 64) {
        const int col = c0 + c;
        double acc = 0.0;
        if (col < width) {
            const double *__restrict__ pc = partials + (int64_t)row0 * width + col;
#pragma unroll 8
            for (int b = g; b < nblocks; b += 16) acc += pc[(int64_t)b * width];
        }
        __syncthreads();
        lds[g][c] = acc;
        __syncthreads();
        if (g == 0 && col < width) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) v += lds[k][c];
            const int which = col / RT, term = t0 + col % RT;
            if (term < int_R) {
                const double t = tg.totals[(int64_t)which * int_R + term] + v;
                tg.totals[(int64_t)which * int_R + term] = t;
                if (tg.h_s) {   // same expressions as k_moments_finalize
                    const double c = tg.scale[term];
                    if (which == 0) tg.h_s[term] = c * t;
                    else tg.h_sp[term] = (c * c) * t;
                    if (tg.p_s) {
                        if (which == 0) tg.p_s[term] = c * t;
                        else tg.p_sp[term] = (c * c) * t;
                    }
                }
            }
        }
    }
    if (tg.counts && g == 15) {   // exact integer sums on the last wave
        int64_t a = 0, b = 0;
        for (int i = c; i < nblocks; i += 64) { a += pcounts[2 * (int64_t)(row0 + i)]; b += pcounts[2 * (int64_t)(row0 + i) + 1]; }
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
        if (c == 0) {
            a += tg.counts[0];
            b += tg.counts[1];
            tg.counts[0] = a;
            tg.counts[1] = b;
            if (tg.h_n) { tg.h_n[0] = a; tg.h_n[tg.h_n_stride] = b; }
            if (tg.p_nd) { tg.p_nd[0] = (double)a; tg.p_nd[tg.h_n_stride] = (double)b; }   // exact below 2^53
        }
    }
}

// ------------------------------------------------------------------------------------------
// Sparse accumulation for SPLINE moments: at most four B-splines are non-zero per value, so a sample touches at most
// eight of the R sums.  Every wave keeps its own copy of the 2 R sums in LDS and updates it with ds_add_f64 (one wave's
// atomics hit only its own copy; the four copies are added in a fixed order at the end).  Same segment table, partial
// rows [block][2 R] and grid reduction as the dense kernel.  phi_0 = 1 is handled through the sample count.
// ------------------------------------------------------------------------------------------
constexpr int SPLINE_MAX_R = 512;

// need_sq = 0: the sums of squares are not wanted (mean-only estimate of TransformedMoments over a spline basis, e.g. the
// orthogonal-moments pass of Estimate.construct_density): half of the LDS atomics -- the kernel's bound, ~3 lane-atomics per
// cycle and CU for any operand type (tools/ubench_lds_atomics.hip) -- are skipped; the squares' partial rows are zeros.
template <bool NEED_SQ>
__global__ __launch_bounds__(ACC_THREADS) void k_spline_accum(BasisParams bp, SegTable tab, int R, double *__restrict__ partials,
                                                             int64_t *__restrict__ pcounts) {
    extern __shared__ double sacc[];              // [4 waves][2][RP], RP = R + 8 (slack for the local 8-window)
    __shared__ int ldc[4][2];
    const int RP = R + 8;
    Seg sg = tab.seg[0];
#pragma unroll
    for (int k = 1; k < MAX_SEG; ++k)
        if (k < tab.nseg && (int)blockIdx.x >= tab.seg[k].block0) sg = tab.seg[k];
    const int bid = (int)blockIdx.x - sg.block0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 2 * RP; i += ACC_THREADS) sacc[i] = 0.0;
    __syncthreads();
    double *__restrict__ ws = sacc + (size_t)wave * 2 * RP;   // sums of d
    double *__restrict__ wsp = ws + RP;                        // sums of d^2
    const bool pair = sg.coarse != nullptr;
    int n_keep = 0, n_rm = 0;
    const int64_t T = (int64_t)sg.nblocks * ACC_THREADS;
    for (int64_t i = (int64_t)bid * ACC_THREADS + threadIdx.x; i < sg.n; i += T) {
        bool kf, kc = true;
        const double tf = transform_value(bp, sg.fine[i], kf);
        double tc = 0.0;
        if (pair) tc = transform_value(bp, sg.coarse[i], kc);
        const bool keep = kf && kc && (!sg.mask || sg.mask[i] != 0);
        n_keep += (int)keep;
        n_rm += (int)!keep;
        if (!keep) continue;
        TermGen<MLMC_SPLINE> gf, gc;
        gf.init(tf, 1.0, bp);
        const double nf[4] = {gf.n0, gf.n1, gf.n2, gf.n3};
        if (!pair) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = gf.k + j;
                if (r > 0) { atomicAdd(&ws[r], nf[j]); if (NEED_SQ) atomicAdd(&wsp[r], nf[j] * nf[j]); }
            }
            continue;
        }
        gc.init(tc, 1.0, bp);
        const double nc[4] = {gc.n0, gc.n1, gc.n2, gc.n3};
        const int delta = gc.k - gf.k;
        if (delta == 0) {                                     // the common case: fine and coarse in the same knot span
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = gf.k + j;
                const double d = nf[j] - nc[j];
                if (r > 0) { atomicAdd(&ws[r], d); if (NEED_SQ) atomicAdd(&wsp[r], d * d); }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                     // indices touched by the fine value
                const int r = gf.k + j, jc = j - delta;       // coarse slot holding the same B-spline, if any
                const double c = jc == 0 ? nc[0] : (jc == 1 ? nc[1] : (jc == 2 ? nc[2] : (jc == 3 ? nc[3] : 0.0)));
                const double d = nf[j] - c;
                if (r > 0) { atomicAdd(&ws[r], d); if (NEED_SQ) atomicAdd(&wsp[r], d * d); }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                     // indices touched by the coarse value only
                const int r = gc.k + j, jf = j + delta;
                if ((jf < 0 || jf > 3) && r > 0) { atomicAdd(&ws[r], -nc[j]); if (NEED_SQ) atomicAdd(&wsp[r], nc[j] * nc[j]); }
            }
        }
    }
    n_keep = wave_sum_i(n_keep);
    n_rm = wave_sum_i(n_rm);
    if (lane == 0) { ldc[wave][0] = n_keep; ldc[wave][1] = n_rm; }
    __syncthreads();
    const int keep_block = ldc[0][0] + ldc[1][0] + ldc[2][0] + ldc[3][0];
    for (int j = threadIdx.x; j < 2 * R; j += ACC_THREADS) {
        const int which = j / R, r = j % R;
        const int o = which * RP + r;
        double v = ((sacc[o] + sacc[2 * RP + o]) + sacc[4 * RP + o]) + sacc[6 * RP + o];
        if (r == 0) v = pair ? 0.0 : (double)keep_block;     // phi_0 = 1: differences vanish; level 0 counts the samples
        partials[(int64_t)blockIdx.x * (2 * R) + j] = v;
    }
    if (threadIdx.x < 2) {
        int v = ldc[0][threadIdx.x] + ldc[1][threadIdx.x] + ldc[2][threadIdx.x] + ldc[3][threadIdx.x];
        pcounts[(int64_t)blockIdx.x * 2 + threadIdx.x] = v;
    }
}

template <int KIND, int RT, int T0C, bool PLAIN>
static int launch_accum_rt(const BasisParams &bp, const SegTable &tab, int total_blocks, int t0,
                           double *partials, int64_t *pcounts) {
    hipLaunchKernelGGL((k_moments_accum<KIND, RT, T0C, PLAIN>), dim3(total_blocks), dim3(ACC_THREADS), 0, rt().stream, bp, tab, t0, partials,
                       pcounts);
    MLMC_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int KIND, int RT, int T0C, bool PLAIN>
static int occupancy_rt(int *per_cu) {
    MLMC_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, (const void *)k_moments_accum<KIND, RT, T0C, PLAIN>, ACC_THREADS, 0));
    return 0;
}

// Register tile of a pass over n_terms terms starting at t0.  First pass: the smallest tile that fits (exact sizes for
// Legendre).  Later passes with t0 = 32 or 64 (48 < R <= 128): compile-time term window, tiles of 16 / 32 / 48 / 64.
// Anything later (R > 128): the run-time kernel.
static int pick_rt(int kind, int n_terms, int t0) {
    if (kind == MLMC_IDENTITY) return 4;
    if (t0 == 32) return n_terms <= 16 ? 16 : 32;
    if (t0 == 64) return n_terms <= 16 ? 16 : (n_terms <= 32 ? 32 : (n_terms <= 48 ? 48 : 64));
    if (t0 > 0) return 64;
    if (kind == MLMC_LEGENDRE) {
        const int opts[] = {4, 6, 8, 10, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64};   // 5, 10: the reference's usual n_moments
        for (int o : opts)
            if (n_terms <= o) return o;
        return 64;
    }
    const int opts[] = {8, 16, 24, 32, 48, 64};
    for (int o : opts)
        if (n_terms <= o) return o;
    return 64;
}

// op 0: *out = resident blocks per CU of the instantiation; op 1: launch
static int accum_dispatch(int op, bool plain, const BasisParams &bp, int rt_sel, const SegTable *tab, int total_blocks,
                          int t0, double *partials, int64_t *pcounts, int *out) {
#define MLMC_RT_GO1(KIND, N, T0C, P) \
    (op == 0 ? occupancy_rt<KIND, N, T0C, P>(out) : launch_accum_rt<KIND, N, T0C, P>(bp, *tab, total_blocks, t0, partials, pcounts))
#define MLMC_RT_GO(KIND, N, T0C) (plain ? MLMC_RT_GO1(KIND, N, T0C, true) : MLMC_RT_GO1(KIND, N, T0C, false))
#define MLMC_RT_CASE(KIND, N) case N: return MLMC_RT_GO(KIND, N, 0)
#define MLMC_RT_LATER(KIND)                                                                                  \
    if (t0 == 32) return rt_sel == 16 ? MLMC_RT_GO(KIND, 16, 32) : MLMC_RT_GO(KIND, 32, 32);                 \
    if (t0 == 64) {                                                                                          \
        switch (rt_sel) {                                                                                    \
            case 16: return MLMC_RT_GO(KIND, 16, 64);                                                        \
            case 32: return MLMC_RT_GO(KIND, 32, 64);                                                        \
            case 48: return MLMC_RT_GO(KIND, 48, 64);                                                        \
            default: return MLMC_RT_GO(KIND, 64, 64);                                                        \
        }                                                                                                    \
    }                                                                                                        \
    return MLMC_RT_GO(KIND, 64, -1)
    if (t0 > 0) {
        switch (bp.kind) {
            case MLMC_LEGENDRE: MLMC_RT_LATER(MLMC_LEGENDRE);
            case MLMC_MONOMIAL: MLMC_RT_LATER(MLMC_MONOMIAL);
            case MLMC_FOURIER: MLMC_RT_LATER(MLMC_FOURIER);
            default: return fail("moments: a later pass needs a polynomial / Fourier basis");
        }
    }
    switch (bp.kind) {
        case MLMC_LEGENDRE:
            switch (rt_sel) {
                MLMC_RT_CASE(MLMC_LEGENDRE, 4); MLMC_RT_CASE(MLMC_LEGENDRE, 6); MLMC_RT_CASE(MLMC_LEGENDRE, 8);
                MLMC_RT_CASE(MLMC_LEGENDRE, 10); MLMC_RT_CASE(MLMC_LEGENDRE, 12);
                MLMC_RT_CASE(MLMC_LEGENDRE, 16); MLMC_RT_CASE(MLMC_LEGENDRE, 20); MLMC_RT_CASE(MLMC_LEGENDRE, 24);
                MLMC_RT_CASE(MLMC_LEGENDRE, 28); MLMC_RT_CASE(MLMC_LEGENDRE, 32); MLMC_RT_CASE(MLMC_LEGENDRE, 40);
                MLMC_RT_CASE(MLMC_LEGENDRE, 48); MLMC_RT_CASE(MLMC_LEGENDRE, 56);
                default: return MLMC_RT_GO(MLMC_LEGENDRE, 64, 0);
            }
        case MLMC_MONOMIAL:
            switch (rt_sel) {
                MLMC_RT_CASE(MLMC_MONOMIAL, 8); MLMC_RT_CASE(MLMC_MONOMIAL, 16); MLMC_RT_CASE(MLMC_MONOMIAL, 24);
                MLMC_RT_CASE(MLMC_MONOMIAL, 32); MLMC_RT_CASE(MLMC_MONOMIAL, 48);
                default: return MLMC_RT_GO(MLMC_MONOMIAL, 64, 0);
            }
        case MLMC_FOURIER:
            switch (rt_sel) {
                MLMC_RT_CASE(MLMC_FOURIER, 8); MLMC_RT_CASE(MLMC_FOURIER, 16); MLMC_RT_CASE(MLMC_FOURIER, 24);
                MLMC_RT_CASE(MLMC_FOURIER, 32); MLMC_RT_CASE(MLMC_FOURIER, 48);
                default: return MLMC_RT_GO(MLMC_FOURIER, 64, 0);
            }
        case MLMC_IDENTITY: return MLMC_RT_GO1(MLMC_IDENTITY, 4, 0, false);
        default: return fail("unknown basis kind");
    }
#undef MLMC_RT_LATER
#undef MLMC_RT_CASE
#undef MLMC_RT_GO
#undef MLMC_RT_GO1
}

// term-split kernel (48 < R <= 64; mean-only: 64 < R <= 128): op 0: *out = resident blocks per CU; op 1: launch
template <int KIND, bool PLAIN, int HT, int TT, int WPS, bool SQ, int T0 = 0>
static int split_go(int op, const BasisParams &bp, const SegTable *tab, int total_blocks, double *partials, int64_t *pcounts, int *out) {
    if (op == 0) {
        MLMC_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(out, (const void *)k_moments_accum_split<KIND, PLAIN, HT, TT, WPS, SQ, MLMC_SPLIT_NS, T0>, ACC_THREADS, 0));
        return 0;
    }
    hipLaunchKernelGGL((k_moments_accum_split<KIND, PLAIN, HT, TT, WPS, SQ, MLMC_SPLIT_NS, T0>), dim3(total_blocks), dim3(ACC_THREADS), 0, rt().stream, bp, *tab, partials, pcounts);
    MLMC_HIP_CHECK(hipGetLastError());
    return 0;
}
// terms of the head of the mean-only variants: the head also loads, transforms and counts (~50 instructions per trip)
constexpr int SPLIT_MEAN_HEAD_96 = 46, SPLIT_MEAN_HEAD_128 = 62;
// second window (terms 128..255): the head walks 128 steps (4 instructions per pair and step) before its accumulating terms
constexpr int SPLIT_MEAN_HEAD_W2 = 40;
// n_split: terms of the instantiation -- 64 (mean + variance), 96 or 128 (mean only); t0 = 128: the second window (mean only)
static int split_dispatch(int op, bool plain, const BasisParams &bp, int n_split, int t0, const SegTable *tab, int total_blocks, double *partials,
                          int64_t *pcounts, int *out) {
#define MLMC_SPLIT_GO(KIND, P)                                                                                                          \
    (t0 == 128 ? split_go<KIND, P, SPLIT_MEAN_HEAD_W2, 128 - SPLIT_MEAN_HEAD_W2, 2, false, 128>(op, bp, tab, total_blocks, partials, pcounts, out) \
     : n_split == 64 ? split_go<KIND, P, SPLIT_HEAD, 64 - SPLIT_HEAD, MLMC_SPLIT_WPS, true>(op, bp, tab, total_blocks, partials, pcounts, out)   \
     : n_split == 96 ? split_go<KIND, P, SPLIT_MEAN_HEAD_96, 96 - SPLIT_MEAN_HEAD_96, 2, false>(op, bp, tab, total_blocks, partials, pcounts, out) \
                     : split_go<KIND, P, SPLIT_MEAN_HEAD_128, 128 - SPLIT_MEAN_HEAD_128, 2, false>(op, bp, tab, total_blocks, partials, pcounts, out))
    if (bp.kind == MLMC_LEGENDRE) return plain ? MLMC_SPLIT_GO(MLMC_LEGENDRE, true) : MLMC_SPLIT_GO(MLMC_LEGENDRE, false);
    return plain ? MLMC_SPLIT_GO(MLMC_MONOMIAL, true) : MLMC_SPLIT_GO(MLMC_MONOMIAL, false);
#undef MLMC_SPLIT_GO
}

// Launch the pending segments of `a` (all passes over the terms), then the grid reduction.
int flush_moments(mlmc_accum *a) {
    const int nseg = (int)a->pending.size();
    if (nseg == 0) return 0;
    hipStream_t st = rt().stream;
    const int R = a->R;
    const BasisParams &bp = a->basis->p;
    const bool sparse_spline = bp.kind == MLMC_SPLINE;
    if (sparse_spline && R > SPLINE_MAX_R) return fail("spline moments: at most 512 basis functions");
    // Terms per pass.  A 64-term tile needs 256 accumulator VGPRs = one wave per SIMD, and a lone wave issues at 6.1
    // cycles per instruction against 4.4-4.7 for two.  Polynomial bases with 48 < R <= 64 therefore run the term-split
    // kernel (k_moments_accum_split: one pass, two waves per SIMD, every recurrence step once); MLMC_HIP_NO_SPLIT=1 keeps
    // the earlier form -- two 32-term passes, the second re-running 32 recurrence steps without accumulating (+29 %
    // instructions) -- for A/B runs.  Fourier, whose second pass would repeat the sincos, keeps one 64-term pass.
    // R <= 48 is one pass at two waves per SIMD; R > 64 uses 64-term passes.
    static const bool no_split = std::getenv("MLMC_HIP_NO_SPLIT") != nullptr;
    const bool poly = bp.kind == MLMC_LEGENDRE || bp.kind == MLMC_MONOMIAL;
    const bool poly64 = poly && R > 48 && R <= 64;
    // mean-only estimate of plain polynomial moments with 64 < R <= 128: ONE pass of the term-split kernel without the sums
    // of squares (k_moments_accum_split<..., SQ = false>)
    const bool split_mean = poly && a->mean_only_plain && R > 64 && R <= 256 && !no_split;
    const bool split = (poly64 && !no_split) || split_mean;
    const int pass_terms = split_mean ? 128 : ((poly64 && !split) ? 32 : MAX_TERMS_PER_PASS);
    for (int t0 = 0; t0 < (sparse_spline ? 1 : R); t0 += pass_terms) {
        const int n_terms = (R - t0 < pass_terms) ? R - t0 : pass_terms;
        const int n_split = split_mean ? ((t0 == 0 && R <= 96) ? 96 : 128) : 64;
        const int rt_sel = sparse_spline ? R : (split ? n_split : pick_rt(bp.kind, n_terms, t0));
        const int width = 2 * rt_sel;
        int per_cu = 4;
        bool plain = bp.kind != MLMC_IDENTITY && !bp.is_log && bp.is_clip;
        for (const PendingSeg &p : a->pending) plain = plain && p.mask == nullptr;
        if (split) {
            static int occ_split[2][2][4];   // [Legendre | monomial][plain][64 | 96 | 128 terms | second window]; 0 = not asked yet
            int &cached = occ_split[bp.kind == MLMC_LEGENDRE ? 0 : 1][plain ? 1 : 0][t0 ? 3 : (n_split == 64 ? 0 : (n_split == 96 ? 1 : 2))];
            if (cached == 0)
                if (int rc = split_dispatch(0, plain, bp, n_split, t0, nullptr, 0, nullptr, nullptr, &cached)) return rc;
            per_cu = cached;
        } else if (!sparse_spline) {
            static int occ_cache[8][4][2][65];   // resident blocks per CU of (kind, pass class, plain, RT); 0 = not asked yet
            int &cached = occ_cache[bp.kind & 7][t0 == 0 ? 0 : (t0 == 64 ? 1 : (t0 == 32 ? 2 : 3))][plain ? 1 : 0][rt_sel];
            if (cached == 0)
                if (int rc = accum_dispatch(0, plain, bp, rt_sel, nullptr, 0, t0, nullptr, nullptr, &cached)) return rc;
            per_cu = cached;
        }
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        const int resident = rt().n_cu * per_cu;
        // blocks per segment in proportion to its work: 7 fp64 instructions per pair and term, 4 at level 0 plus the
        // per-sample overhead (measured time ratio pair : level-0 = 1.5 at R = 32)
        constexpr double W_PAIR = 7.0, W_SINGLE = 4.3;
        double wsum = 0.0;
        for (const PendingSeg &p : a->pending) wsum += (double)p.n * (p.coarse ? W_PAIR : W_SINGLE);
        SegTable tab;
        ReduceTable rtab;
        std::memset(&tab, 0, sizeof(tab));
        std::memset(&rtab, 0, sizeof(rtab));
        tab.nseg = nseg;
        tab.prio_mod = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
        int total = 0;
        int64_t bytes = 0;
        for (int k = 0; k < nseg; ++k) {
            const PendingSeg &p = a->pending[k];
            // floor: the whole grid must be resident at once (one block over the limit would run as a second round)
            int nb = (int)((double)resident * ((double)p.n * (p.coarse ? W_PAIR : W_SINGLE)) / wsum);
            const int64_t per_trip = split ? (int64_t)MLMC_SPLIT_NS * SPLIT_LANES : 2 * ACC_THREADS;      // samples a workgroup takes per trip
            const int64_t want = (p.n + per_trip - 1) / per_trip;
            if (nb > want) nb = (int)want;
            if (nb < 1) nb = 1;
            tab.seg[k].fine = p.fine;
            tab.seg[k].coarse = p.coarse;
            tab.seg[k].mask = p.mask;
            tab.seg[k].n = p.n;
            tab.seg[k].block0 = total;
            tab.seg[k].nblocks = nb;
            total += nb;
            rtab.t[k].totals = a->d_totals + ((int64_t)p.level * a->n_comp + p.comp) * a->int_width;
            rtab.t[k].counts = (p.count && t0 == 0) ? a->d_counts + (int64_t)p.level * 2 : nullptr;
            if (a->host_outputs) {   // plain basis, one component: the reduction writes the finished rows (see ReduceTarget)
                rtab.t[k].h_s = a->h_out_s + (int64_t)p.level * R;
                rtab.t[k].h_sp = a->h_out_sp + (int64_t)p.level * R;
                rtab.t[k].h_n = a->h_out_n + p.level;
                rtab.t[k].h_n_stride = a->n_levels;
                if (a->packed_target) {   // mlmc_accum_finalize_packed(MLMC_DEVICE): n | n_rm | s | sp as doubles
                    double *pk = a->packed_target;
                    rtab.t[k].p_nd = pk + p.level;
                    rtab.t[k].p_s = pk + 2 * (int64_t)a->n_levels + (int64_t)p.level * R;
                    rtab.t[k].p_sp = rtab.t[k].p_s + (int64_t)a->n_levels * R;
                }
                rtab.t[k].scale = a->basis->d_scale;
                a->level_flushed[p.level] = 1;
            }
            bytes += p.n * (p.coarse ? 16 : 8);
        }
        if (int rc = ensure((void **)&a->d_partials, &a->partials_cap, sizeof(double) * (size_t)total * width)) return rc;
        if (int rc = ensure((void **)&a->d_pcounts, &a->pcounts_cap, sizeof(int64_t) * (size_t)total * 2)) return rc;
        if (int rc = timing_begin(a)) return rc;
        if (sparse_spline) {
            const size_t lds = sizeof(double) * 4 * 2 * (size_t)(R + 8);
            if (a->mean_only)   // MOMENTS of TransformedMoments, variances not wanted: the squares feed nothing
                hipLaunchKernelGGL(k_spline_accum<false>, dim3(total), dim3(ACC_THREADS), lds, st, bp, tab, R, a->d_partials, a->d_pcounts);
            else
                hipLaunchKernelGGL(k_spline_accum<true>, dim3(total), dim3(ACC_THREADS), lds, st, bp, tab, R, a->d_partials, a->d_pcounts);
            MLMC_HIP_CHECK(hipGetLastError());
        } else if (split) {
            if (int rc = split_dispatch(1, plain, bp, n_split, t0, &tab, total, a->d_partials, a->d_pcounts, nullptr)) return rc;
        } else if (int rc = accum_dispatch(1, plain, bp, rt_sel, &tab, total, t0, a->d_partials, a->d_pcounts, nullptr)) {
            return rc;
        }
        if (int rc = timing_end(a)) return rc;
        a->launches += 1;
        a->alg_bytes += bytes;
        hipLaunchKernelGGL(k_reduce_partials, dim3(nseg), dim3(1024), 0, st, a->d_partials, a->d_pcounts, tab, rtab, width, rt_sel, t0, R);
        MLMC_HIP_CHECK(hipGetLastError());
    }
    a->pending.clear();
    return 0;
}

// Queue one chunk (device pointers).  Chunks of different levels are gathered into one launch; a second chunk of a
// level that is already queued (or a full table) flushes first.  `defer` false: launch immediately.
int launch_moments_accum(mlmc_accum *a, int level, int comp, const double *d_f, const double *d_c, const uint8_t *d_mask,
                         int64_t n, bool count, bool defer) {
    if (n == 0) return 0;
    for (const PendingSeg &p : a->pending)
        if (p.level == level && p.comp == comp) {
            if (int rc = flush_moments(a)) return rc;
            break;
        }
    if ((int)a->pending.size() >= MAX_SEG)
        if (int rc = flush_moments(a)) return rc;
    a->pending.push_back(PendingSeg{d_f, d_c, d_mask, n, level, comp, count});
    if (!defer) return flush_moments(a);
    return 0;
}

// ------------------------------------------------------------------------------------------
// finalize (MOMENTS): internal sums of the scaled basis -> caller rows.
//   no transform : s_i = c_i S_i,  sp_i = c_i^2 SP_i
//   transform T  : s_j = sum_i T_ji c_i S_i,  sp_j = sum_ik T_ji T_jk c_i c_k G_ik   (G = sum d d^T)
// (sum_n (T d_n)_j^2 = (T G T^T)_jj: the per-sample matrix product of TransformedMoments._eval_all,
//  moments.py:256-259, is applied once to the accumulated second moments instead of to every sample.)
// ------------------------------------------------------------------------------------------
__global__ void k_moments_finalize(const double *__restrict__ totals, const double *__restrict__ scale_c,
                                   const double *__restrict__ T, int R, int RP, int Rout, int has_T, int64_t int_width,
                                   int n_lc, double *__restrict__ out_s, double *__restrict__ out_sp,
                                   const int64_t *__restrict__ counts, int n_levels, int64_t *__restrict__ out_n,
                               double *__restrict__ out_nd, int mean_only) {
    const int lc = blockIdx.x;   // (level, comp)
    if (lc >= n_lc) return;
    if (lc == 0)
        for (int l = threadIdx.x; l < n_levels; l += blockDim.x) {   // (kept, removed) pairs -> n[L], n_rm[L]
            out_n[l] = counts[2 * l];
            out_n[n_levels + l] = counts[2 * l + 1];
            out_nd[l] = (double)counts[2 * l];               // exact below 2^53: lets one fp64 all-reduce carry the counts
            out_nd[n_levels + l] = (double)counts[2 * l + 1];
        }
    const double *tot = totals + (int64_t)lc * int_width;
    for (int j = threadIdx.x; j < Rout; j += blockDim.x) {
        double s, sp;
        if (!has_T) {
            const double c = scale_c[j];
            s = c * tot[j];
            sp = (c * c) * tot[R + j];
        } else {
            const double *G = tot + 2 * R;   // [RP][RP]
            s = 0.0;
            sp = 0.0;
            for (int i = 0; i < R; ++i) {
                const double ti = T[(int64_t)j * R + i] * scale_c[i];
                s = __builtin_fma(ti, tot[i], s);
                if (mean_only) continue;
                double row = 0.0;
                for (int k = 0; k < R; ++k) row = __builtin_fma(T[(int64_t)j * R + k] * scale_c[k], G[(int64_t)i * RP + k], row);
                sp = __builtin_fma(ti, row, sp);
            }
            if (mean_only) sp = __builtin_nan("");   // MLMC_MODE_MEAN_ONLY: the diff Gram matrix was not accumulated
        }
        out_s[(int64_t)lc * Rout + j] = s;
        out_sp[(int64_t)lc * Rout + j] = sp;
    }
}

int launch_moments_finalize(mlmc_accum *a) {
    const int n_lc = a->n_levels * a->n_comp;
    hipLaunchKernelGGL(k_moments_finalize, dim3(n_lc), dim3(128), 0, rt().stream, a->d_totals, a->basis->d_scale,
                       a->basis->d_matrix, a->R, a->RP, a->Rout, a->basis->out_size > 0 ? 1 : 0, a->int_width, n_lc,
                       a->d_out_s, a->d_out_sp, a->d_counts, a->n_levels, a->d_out_n, a->d_out_nd, a->mean_only ? 1 : 0);
    MLMC_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mlmc
