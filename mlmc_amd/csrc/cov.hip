// Moment-covariance accumulation on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), gfx950.
//
// Reference (quantity_estimate.py:131-147 + :59-65): per sample the outer products f f^T and c c^T are
// materialised ([2, M, n, R, R] doubles), subtracted, and summed together with their squares.
// Here, with d = f - c and s = f + c (f, c the moment vectors of the fine / coarse value of one sample):
//     f f^T - c c^T              = 1/2 (d s^T + s d^T)
//     (f_i f_j - c_i c_j)^2      = 1/4 (d_i^2 s_j^2 + 2 d_i s_i d_j s_j + s_i^2 d_j^2)
// so the level sums are three GEMM-shaped contractions over the sample index
//     G0 = D^T S,   G1 = (D.D)^T (S.S),   G2 = (D.S)^T (D.S)
//     sum (ff^T - cc^T) = 1/2 (G0 + G0^T),   sum (..)^2 = 1/4 (G1 + G1^T + 2 G2)
// with no cancellation between large Gram matrices.  Level 0 (no coarse): d = s = f.
//
// Work split: a 256-thread workgroup owns a (16 T x 16 T) block of the output and walks batches of 64
// samples.  Phase 1: two waves run the term recurrences (lane = (sample, fine|coarse)) and write the
// values term-major into LDS (conflict-free: row stride 66 doubles).  Phase 2: every wave reads MFMA
// fragments (A[i][k]: lane -> term i = lane & 15, sample k = lane >> 4) and issues 3 MFMAs per
// (row tile, column tile, 4 samples).  Two workgroups per CU so one's phase 1 runs under the other's MFMAs.
#include "device_basis.hpp"

namespace mlmc {

typedef double v4f64 __attribute__((ext_vector_type(4)));
#ifdef MLMC_PROF_COV   // diagnostic build only (tools/prof_cov.hip): per-wave shader cycles of the phases and barrier waits
__device__ unsigned long long *g_prof_cov;
#define MLMC_COV_STAMP(slot)                                      \
    {                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        prof_acc[slot] += now_ - prof_t;                          \
        prof_t = now_;                                            \
    }
#else
#define MLMC_COV_STAMP(slot)
#endif
// Issue priorities of the two phases.  The workgroups of a CU take turns: while some run their matrix phase another one
// evaluates, and below the matrix instructions its dependent recurrence chain would get one issue slot per matrix instruction
// (the fp64 pipe is one resource).  At the higher priority the evaluation phase takes a third of that time and the pipe never
// waits for it (-2.1 % at R = 64, -1.8 % / -5.6 % at R = 32 with / without variances, -3.9 % at R = 16, same-box A/B; round 1
// had it the other way round).
// Fairness between the workgroups of a CU (64-term kernel): at equal priority the SIMD arbiter serves the older wave first,
// the workgroup that was dispatched first runs ahead, finishes early (two per CU: 3.0 against 3.6 ms at a pair level; four
// per CU: 2.5 / 2.9 / 3.2 / 3.5 ms) and leaves the others to run with fewer waves per SIMD.  So the matrix phases change their
// priority in slices of 2^MLMC_COV_SLICE_BITS shader cycles by the workgroup's rank on its CU:
//   with variances at a pair level (RANKS): the ranks rotate through the priorities 0 .. 3, evaluation at 3;
//   otherwise: one workgroup at 1, the others at 0, evaluation at 2 (those kernels end together anyway; rotating ranks cost
//   them 1-2 %).
// Two per CU: -4.8 % (slices of 2^15 / 2^17 / 2^19 / 2^21: 15.88 / 15.70 / 15.60 / 15.66 ms per configs[2] estimate); four per
// CU with rotating ranks: a further -1.7 %.  The smaller tiles end together without slices and lose 1-4 % with them.
#ifndef MLMC_COV_PRIO_MFMA
#define MLMC_COV_PRIO_MFMA 0
#define MLMC_COV_PRIO_EVAL 2
#endif
#ifndef MLMC_COV_SLICE_BITS
#define MLMC_COV_SLICE_BITS 17
#endif
#ifndef MLMC_COV_EXTRA
// The youngest workgroup of a CU still ends ~6 % after the others (what the priorities do not reach -- LDS and scalar
// arbitration -- stays oldest-first): after every MLMC_COV_EXTRA rounds of turns it gets one turn on top (0: never).
// Extra turn after 1 / 2 / 4 rounds: 15.12 / 15.04 / 15.14 ms per configs[2] estimate against 15.19 without.
#define MLMC_COV_EXTRA 2
#endif
#define MLMC_COV_MFMA_PRIO(ranks, slot, nslots, clock)                                                \
    do {                                                                                              \
        const unsigned t_ = (unsigned)((clock) >> MLMC_COV_SLICE_BITS);                               \
        unsigned top_ = t_ % (nslots);                                                                \
        if (MLMC_COV_EXTRA > 0) {                                                                     \
            const unsigned u_ = t_ % ((nslots) * MLMC_COV_EXTRA + 1u);                                \
            top_ = u_ == (nslots) * MLMC_COV_EXTRA ? (nslots) - 1u : u_ % (nslots);                   \
        }                                                                                             \
        const unsigned rank_ = ((slot) + (nslots) - top_ + (nslots) - 1u) % (nslots);  /* top -> nslots - 1 */ \
        if (ranks) {                                                                                  \
            if (rank_ == 0) __builtin_amdgcn_s_setprio(0);                                            \
            else if (rank_ == 1) __builtin_amdgcn_s_setprio(1);                                       \
            else if (rank_ == 2) __builtin_amdgcn_s_setprio(2);                                       \
            else __builtin_amdgcn_s_setprio(3);                                                       \
        } else {                                                                                      \
            if (rank_ == (nslots) - 1u) __builtin_amdgcn_s_setprio(1);                                \
            else __builtin_amdgcn_s_setprio(0);                                                       \
        }                                                                                             \
    } while (0)
#define MLMC_COV_EVAL_PRIO(ranks)                                                                     \
    do {                                                                                              \
        if (ranks) __builtin_amdgcn_s_setprio(3);                                                     \
        else __builtin_amdgcn_s_setprio(MLMC_COV_PRIO_EVAL);                                          \
    } while (0)
constexpr int COV_BATCH = 64;
#ifndef MLMC_COV_BLK22
#define MLMC_COV_BLK22 1
#endif
#ifndef MLMC_COV_T4_ONE_LIST
#define MLMC_COV_T4_ONE_LIST 1
#endif
// 64-term kernel: batches of 32 pairs (35 KB of LDS) and FOUR workgroups per CU -- against 64 pairs and two workgroups: -1 %
// with variances, -8 % mean-only, -3.5 % at level 0 (same-box A/B): four waves per SIMD cover each other's barriers and
// evaluation phases better than two (the kernel fits the 128 registers per lane of that occupancy).
#ifndef MLMC_COV_T4_BATCH
#define MLMC_COV_T4_BATCH 32
#define MLMC_COV_T4_WGS 4
#endif
constexpr int COV_T4_BATCH = MLMC_COV_T4_BATCH;   // pairs per batch of the 64-term kernel (level 0: twice as many samples)
constexpr int COV_T4_WGS = MLMC_COV_T4_WGS;       // its workgroups per CU
// samples per batch: small tiles evaluated from raw samples take 128 sample pairs, or 256 samples at level 0 (one LDS array
// instead of two, and all four waves evaluate at both kinds of level); everything else 64
// 64-term kernel: the samples of a batch sit in LDS in the order the fragment reads want them -- the two samples a lane needs
// in k-steps 2j and 2j + 1 next to each other -- so one ds_read_b128 fetches a lane's operand values for TWO k-steps (half
// the LDS instructions of phase 2).  0: one ds_read_b64 per k-step (round 2).
#ifndef MLMC_COV_B128
#define MLMC_COV_B128 1
#endif
#ifndef MLMC_COV_B128_PAD
#define MLMC_COV_B128_PAD (MLMC_COV_B128 ? 4 : 2)
#endif
// position of sample s of a batch in its LDS row: s = 8 j + 4 h + g (k-step 2 j + h, lane group g) -> 8 j + 2 g + h
__device__ __forceinline__ int cov_t4_pos(int s) { return MLMC_COV_B128 ? ((s & ~7) | ((s & 3) << 1) | ((s >> 2) & 1)) : s; }
#ifndef MLMC_COV_WIDE_BATCH
#define MLMC_COV_WIDE_BATCH 32      // pairs per batch of the two-window (off-diagonal) blocks: 70 KB of LDS, two workgroups per
                                    // CU (64 pairs, 135 KB, one per CU: +3 % with variances, +5 % mean-only at R = 128)
#endif
// Gram matrices a kernel MODE leaves in its partial rows
__host__ __device__ constexpr int cov_ng(int mode) { return mode == 0 ? 3 : (mode == 3 ? 2 : 1); }
#ifndef MLMC_COV_VAR_RANKS
#define MLMC_COV_VAR_RANKS 1        // rotating matrix-phase priorities in the variance-only 64-term kernel (as with all three Grams)
#endif
__host__ __device__ constexpr int cov_batch(int T, bool wide, bool vals, bool pair = true) {
    return (T <= 2 && !wide && !vals) ? (pair ? 128 : 256) : ((wide && pair && !vals) ? MLMC_COV_WIDE_BATCH : 64);
}

// Spline moments in phase 1: at most four B-splines (and phi_0 = 1) are non-zero per value, so instead of selecting the
// value of every term in registers (four compares and selects per term: the evaluation phase of a 128-moment spline
// covariance took four times that of Legendre) the lane clears its column of the window with plain LDS stores -- which
// cost the shared fp64 pipe nothing -- and drops the five values in place.  Window = terms [TA, TA + NT).
template <int NT, int STRIDE, int TA>
__device__ __forceinline__ void cov_spline_store(const TermGen<MLMC_SPLINE> &g, double *__restrict__ dst, int samp) {
#pragma unroll
    for (int i = 0; i < NT; ++i) dst[i * STRIDE + samp] = 0.0;
    if (TA == 0) dst[samp] = g.w;               // phi_0 = 1 (0 for a masked sample)
    const double nj[4] = {g.n0, g.n1, g.n2, g.n3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int term = g.k + j;               // B_term, term >= 1 (B_0 is replaced by phi_0)
        const int r = term - TA;
        if (term >= 1 && r >= 0 && r < NT) dst[r * STRIDE + samp] = nj[j];
    }
}

// MODE 0: G0, G1, G2 (covariance mean + variance);  MODE 1: G = D^T D only (variance of transformed moments);
// MODE 2: G0 = D^T S only (covariance mean without its variance, e.g. Estimate.construct_density)
// MODE 3: G1, G2 without G0 -- the variance of the covariance, its mean coming from the level sums of the 2 R - 1 moments of
// the product linearisation (api.hip: lin accumulator)
// BI, BJ (T = 4 only): the 64 x 64 output block (rows = terms [64 BI, 64 BI + 64), columns = terms [64 BJ, ...)) of
// a covariance with more than 64 moments; off-diagonal blocks keep two term windows in LDS (one workgroup per CU).
// VALS: `fine` / `coarse` hold already evaluated moment values [n][R] (row-major, NaN rows = masked samples) instead
// of raw samples: the path for moment functions that are not evaluated in registers (TransformedMoments).
template <int KIND, int T, bool PAIR, int MODE, int BI = 0, int BJ = 0, bool VALS = false>
__global__ __launch_bounds__(256, (BI != BJ && (!PAIR || MLMC_COV_WIDE_BATCH == 64 || VALS)) ? 1 : 2) void k_cov_accum(BasisParams bp, 
                                                      const double *__restrict__ fine,
                                                      const double *__restrict__ coarse,
                                                      const uint8_t *__restrict__ mask, int64_t n, int R,
                                                      double *__restrict__ partials, int64_t *__restrict__ pcounts,
                                                      int vals_ta, int vals_tb) {
    // vals_ta / vals_tb (VALS only): first column of the row / column window inside the materialised value rows [n][R] -- run-time
    // values, so ONE pair of instantiations (diagonal, off-diagonal) serves every 64 x 64 block of a covariance of any size
    constexpr int NT = 16 * T;                 // terms held in LDS
    // samples per batch: 128 for the small tiles evaluated from raw samples (all four waves run recurrences, the
    // barriers and the phase-1 latency are shared by twice the MFMA work), 64 otherwise
    constexpr int BATCH = cov_batch(T, BI != BJ, VALS, PAIR);
    constexpr int STRIDE = BATCH + 2;          // doubles per term row: == 2 (mod 32) -> ds_read_b64 fragments hit 32 distinct bank pairs
    constexpr int NSL = 4 / T;                 // k-slices (waves sharing a row tile split the samples)
    // MODE 3 in THIS kernel: MODE 0 without the G0 instructions -- same accumulator sets and partial layout, the G0 slot stays
    // zero (the 64-term kernel has its own two-matrix layout)
    constexpr bool M0 = MODE == 0 || MODE == 3;
    constexpr bool DO_G0 = MODE == 0;
    constexpr int NG = M0 ? (PAIR ? 3 : 2) : 1;
    constexpr bool WIDE = BI != BJ;            // two different term windows
    constexpr int TA = 64 * BI, TB = 64 * BJ;  // first term of the row / column window
    constexpr int N_EVAL = (BI > BJ ? TA : TB) + NT;   // terms the recurrence has to run through
    static_assert(T == 4 || (BI == 0 && BJ == 0), "term windows need T = 4");
    __shared__ double lds_f[NT * STRIDE];
    __shared__ double lds_c[PAIR ? NT * STRIDE : 1];
    __shared__ double lds_fb[WIDE ? NT * STRIDE : 1];              // column window (off-diagonal blocks)
    __shared__ double lds_cb[(WIDE && PAIR) ? NT * STRIDE : 1];
    __shared__ int ldc[4][2];
    __shared__ unsigned char keep_s[VALS ? BATCH : 1];

    // blockIdx.y = component of a vector quantity ([M][n] arrays, one mask for all): its own samples and partial rows
    fine += (int64_t)blockIdx.y * n;
    if (PAIR) coarse += (int64_t)blockIdx.y * n;
    partials += (int64_t)blockIdx.y * gridDim.x * ((T <= 2 && BI == BJ) ? 4 : 4 / T) * ((M0 ? 3 : 1) * (16 * T) * (16 * T));
    if (blockIdx.y) pcounts = nullptr;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int I = wave % T, kslice = wave / T;

    // Small tiles (T <= 2, one term window): a wave owns a k-slice (a quarter of the batch's samples) and ALL T x T tiles.
    // The A fragment of tile row I and the B fragment of tile column I are the same LDS words (lane -> (term 16 I +
    // lane % 16, sample 4 ks + lane / 16) in both operand layouts), so one read of the fine / coarse values of the T
    // row blocks feeds T x T MFMAs per Gram matrix -- a third of the LDS traffic of the row-tile mapping below, which at
    // these tile sizes was as busy as the matrix pipe itself (1.5 KB of ds_read per MFMA against 2 KB per 64 cycles).
    constexpr bool SLICED = (T <= 2) && !WIDE;
    constexpr int TI = SLICED ? T : 1;
    v4f64 acc[NG][TI][T];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < T; ++j) acc[g][i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

    // phase-1 role of this lane
    const int n_eval_waves = PAIR ? BATCH / 32 : BATCH / 64;
    const bool evaluator = wave < n_eval_waves;
    const int samp = PAIR ? (wave * 32 + (lane & 31)) : (wave * 64 + lane);   // sample slot in the batch
    const bool is_coarse = PAIR && (lane >> 5);
    const double *__restrict__ src = is_coarse ? coarse : fine;
    double *__restrict__ dst = is_coarse ? lds_c : lds_f;
    double *__restrict__ dst_b = WIDE ? (is_coarse ? lds_cb : lds_fb) : dst;
    int n_keep = 0, n_rm = 0;

    const int64_t n_batches = (n + BATCH - 1) / BATCH;
    int64_t batch = blockIdx.x;
    double xv = 0.0;
    uint8_t mv = 1;
    if (!VALS && evaluator && batch < n_batches) {
        int64_t idx = batch * BATCH + samp;
        if (idx < n) { xv = src[idx]; if (mask) mv = mask[idx]; }
    }
    // VALS: the NT x BATCH tile(s) of values are loaded by the whole workgroup -- consecutive threads take consecutive terms of one
    // sample row (coalesced segments of the row-major [n][R] values), all loads of a thread independent, one batch ahead.  (One
    // evaluator lane per sample walking its row serially, as in the sample-evaluating form, left this phase latency bound: 18
    // TFLOP/s where the same matrix work from registers runs at 52.)
    constexpr int VSPP = 256 / NT;                       // samples per pass
    constexpr int VNP = VALS ? BATCH / VSPP : 1;         // passes = values per thread, operand and window
    const int vtt = threadIdx.x % NT, vss = threadIdx.x / NT;
    const bool v_in_a = VALS && vals_ta + vtt < R, v_in_b = VALS && WIDE && vals_tb + vtt < R;
    double pf[VNP][4];
    auto vals_load = [&](int64_t b) {
#pragma unroll
        for (int p = 0; p < VNP; ++p) {
            int64_t idx = b * BATCH + p * VSPP + vss;
            if (idx >= n) idx = n - 1;                   // clamped, unconditional: the keep flags decide what counts
            const int64_t ra = idx * (int64_t)R + (v_in_a ? vals_ta + vtt : 0), rb = idx * (int64_t)R + (v_in_b ? vals_tb + vtt : 0);
            pf[p][0] = fine[ra];
            pf[p][1] = PAIR ? coarse[ra] : 0.0;
            pf[p][2] = WIDE ? fine[rb] : 0.0;
            pf[p][3] = (WIDE && PAIR) ? coarse[rb] : 0.0;
        }
    };
    if (VALS && batch < n_batches) vals_load(batch);
#ifdef MLMC_PROF_COV
    unsigned long long prof_acc[4] = {0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
    const unsigned long long prof_t0 = prof_t, prof_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_s_setprio(MLMC_COV_PRIO_EVAL);
    for (; batch < n_batches; batch += gridDim.x) {
        // ---------------- phase 1: moment values of this batch -> LDS ----------------
        if (VALS) {
            // (1) keep flags, one thread per sample (a masked value is NaN in every column: column 0 tells)
            if (threadIdx.x < BATCH) {
                const int64_t idx = batch * BATCH + threadIdx.x;
                const bool valid = idx < n;
                bool keep = valid && (!mask || mask[idx] != 0);
                if (keep) { const double v0 = fine[idx * (int64_t)R]; keep = !(v0 != v0); }
                if (PAIR && keep) { const double c0 = coarse[idx * (int64_t)R]; keep = !(c0 != c0); }
                n_keep += (int)keep;
                n_rm += (int)(valid && !keep);
                keep_s[threadIdx.x] = keep ? 1 : 0;
            }
            __syncthreads();
            // The tile of THIS batch was requested a batch ago (registers pf): it is dropped into LDS under the keep flags, then the
            // next batch's loads are issued and fly during the matrix phase.
#pragma unroll
            for (int p = 0; p < VNP; ++p) {
                const int sl = p * VSPP + vss;
                const bool k = keep_s[sl] != 0;
                lds_f[vtt * STRIDE + sl] = (k && v_in_a) ? pf[p][0] : 0.0;
                if (PAIR) lds_c[vtt * STRIDE + sl] = (k && v_in_a) ? pf[p][1] : 0.0;
                if (WIDE) {
                    lds_fb[vtt * STRIDE + sl] = (k && v_in_b) ? pf[p][2] : 0.0;
                    if (PAIR) lds_cb[vtt * STRIDE + sl] = (k && v_in_b) ? pf[p][3] : 0.0;
                }
            }
            if (batch + gridDim.x < n_batches) vals_load(batch + gridDim.x);
        } else if (evaluator) {
            const int64_t idx = batch * BATCH + samp;
            const bool valid = idx < n;
            bool keep;
            double t = transform_value(bp, xv, keep);
            keep = keep && valid && (mv != 0);
            if (PAIR) {   // the shuffle must run in every lane (no short-circuit): fine lane l pairs with coarse lane l + 32
                const int other = __shfl_xor((int)keep, 32, 64);
                keep = keep && (other != 0);
            }
            if (!is_coarse) { n_keep += (int)keep; n_rm += (int)(valid && !keep); }
            // prefetch the next batch's value
            const int64_t nidx = (batch + gridDim.x) * BATCH + samp;
            if (nidx < n) { xv = src[nidx]; if (mask) mv = mask[nidx]; }
            TermGen<KIND> g;
            g.init(keep ? t : 0.0, keep ? 1.0 : 0.0, bp);
            if constexpr (KIND == MLMC_SPLINE) {
                cov_spline_store<NT, STRIDE, TA>(g, dst, samp);
                if (WIDE) cov_spline_store<NT, STRIDE, TB>(g, dst_b, samp);
            } else {
                // all NT terms, fully unrolled (compile-time indices, no branches); rows >= R of the Gram matrices are
                // never read back
#pragma unroll
                for (int i = 0; i < N_EVAL; ++i) {
                    const double q = g.next(i);
                    if (i >= TA && i < TA + NT) dst[(i - TA) * STRIDE + samp] = q;
                    if (WIDE && i >= TB && i < TB + NT) dst_b[(i - TB) * STRIDE + samp] = q;
                }
            }
        }
        MLMC_COV_STAMP(0)
        __syncthreads();
        MLMC_COV_STAMP(1)
        // ---------------- phase 2: MFMA over the batch ----------------
        __builtin_amdgcn_s_setprio(MLMC_COV_PRIO_MFMA);
        if constexpr (SLICED) {
            const int rowl = lane & 15;
#pragma unroll
            for (int kk = 0; kk < BATCH / 16; ++kk) {
                const int col = 4 * (wave + 4 * kk) + (lane >> 4);
                double d[T], sv[T], dd[T], ss[T], ds[T];
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    const double f = lds_f[(16 * i + rowl) * STRIDE + col];
                    d[i] = sv[i] = f;
                    if (PAIR) {
                        const double c = lds_c[(16 * i + rowl) * STRIDE + col];
                        d[i] = f - c;
                        sv[i] = f + c;
                    }
                    if (M0) {
                        dd[i] = d[i] * d[i];
                        ss[i] = sv[i] * sv[i];
                        ds[i] = d[i] * sv[i];
                    }
                }
            // symmetric Gram matrices (both operands the same vector: everything at level 0, G2 and D^T D with pairs) need
            // only their upper tiles; the lower ones are mirrored when the partial tiles are written
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j) {
                        const bool upper = j >= i;
                        if (M0) {
                            if (DO_G0 && (PAIR || upper)) acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[i], sv[j], acc[0][i][j], 0, 0, 0);
                            if (PAIR || upper) acc[1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(dd[i], ss[j], acc[1][i][j], 0, 0, 0);
                            if (PAIR && upper) acc[NG - 1][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ds[i], ds[j], acc[NG - 1][i][j], 0, 0, 0);
                        } else if (MODE == 1 || !PAIR) {
                            if (upper) acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[i], d[j], acc[0][i][j], 0, 0, 0);
                        } else {
                            acc[0][i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[i], sv[j], acc[0][i][j], 0, 0, 0);
                        }
                    }
            }
            __builtin_amdgcn_s_setprio(MLMC_COV_PRIO_EVAL);
            MLMC_COV_STAMP(2)
            __syncthreads();
            MLMC_COV_STAMP(3)
            continue;
        }
        const int arow = 16 * I + (lane & 15);
        // fixed trip count -> fully unrolled: the compiler hoists the next steps' LDS reads above the MFMAs
#pragma unroll
        for (int kk = 0; kk < BATCH / 4 / NSL; ++kk) {
            const int ks = kslice + kk * NSL;
            const int col = 4 * ks + (lane >> 4);
            double fa = lds_f[arow * STRIDE + col];
            double da = fa, sa = fa;
            if (PAIR) {
                double ca = lds_c[arow * STRIDE + col];
                da = fa - ca;
                sa = fa + ca;
            }
            const double a1 = da * da, a2 = da * sa;
#pragma unroll
            for (int J = 0; J < T; ++J) {
                const int brow = 16 * J + (lane & 15);
                double fb = (WIDE ? lds_fb : lds_f)[brow * STRIDE + col];
                double db = fb, sb = fb;
                if (PAIR) {
                    double cb = (WIDE ? lds_cb : lds_c)[brow * STRIDE + col];
                    db = fb - cb;
                    sb = fb + cb;
                }
                if (M0) {
                    if (DO_G0) acc[0][0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(da, sb, acc[0][0][J], 0, 0, 0);
                    if (PAIR) {
                        acc[1][0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, sb * sb, acc[1][0][J], 0, 0, 0);
                        acc[2][0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, db * sb, acc[2][0][J], 0, 0, 0);
                    } else {
                        acc[1][0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, fb * fb, acc[1][0][J], 0, 0, 0);
                    }
                } else {
                    acc[0][0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(da, MODE == 2 ? sb : db, acc[0][0][J], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(MLMC_COV_PRIO_EVAL);
        __syncthreads();
    }

#ifdef MLMC_PROF_COV
    if (lane == 0 && blockIdx.y == 0) {
        unsigned long long *q = g_prof_cov + ((size_t)blockIdx.x * 4 + wave) * 6;
        q[0] = prof_acc[0]; q[1] = prof_acc[1]; q[2] = prof_acc[2]; q[3] = prof_acc[3];
        q[4] = __builtin_amdgcn_s_memtime() - prof_t0;
        q[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xffffu) |   // HW_ID: SIMD in bits 5:4
               ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 16) |      // XCC_ID
               ((__builtin_amdgcn_s_memrealtime() - prof_r0) << 24);                                           // 100 MHz ticks
    }
#endif
    // ---------------- write the workgroup's partial tiles ----------------
    // partial row = (block, kslice); columns [g][row][col] with g in {G0, G1, G2} (MODE 0) or {G} (MODE 1)
    constexpr int NGOUT = M0 ? 3 : 1;
    // SLICED: one partial row per wave (its k-slice) holding all tiles; else one per (block, kslice), a wave writes row tile I
    double *__restrict__ prow = partials + (SLICED ? ((int64_t)blockIdx.x * 4 + wave) : ((int64_t)blockIdx.x * NSL + kslice)) * (NGOUT * NT * NT);
#pragma unroll
    for (int Iw = 0; Iw < TI; ++Iw) {
        const int It = SLICED ? Iw : I;
#pragma unroll
        for (int J = 0; J < T; ++J) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * It + (lane >> 4) + 4 * r;
                const int col = 16 * J + (lane & 15);
                // SLICED: lower tiles of the symmetric matrices were not computed: tile (J, I) transposed stands in
                const bool low = SLICED && J < It;
                // acc of tile (J, It) holds element (16 J + lane / 16 + 4 r, 16 It + lane % 16): it is written transposed
                const int mirror = (16 * It + (lane & 15)) * NT + (16 * J + (lane >> 4) + 4 * r);
                if (M0) {
                    if (PAIR) {
                        prow[0 * NT * NT + row * NT + col] = acc[0][Iw][J][r];
                        prow[1 * NT * NT + row * NT + col] = acc[1][Iw][J][r];
                        if (!low) prow[2 * NT * NT + row * NT + col] = acc[NG - 1][Iw][J][r];
                        else prow[2 * NT * NT + mirror] = acc[NG - 1][SLICED ? J : 0][It][r];
                    } else if (!low) {   // level 0: d = s = f  ->  G2 = G1 = (F.F)^T (F.F)
                        prow[0 * NT * NT + row * NT + col] = acc[0][Iw][J][r];
                        prow[1 * NT * NT + row * NT + col] = acc[1][Iw][J][r];
                        prow[2 * NT * NT + row * NT + col] = acc[1][Iw][J][r];
                    } else {
                        prow[0 * NT * NT + mirror] = acc[0][SLICED ? J : 0][It][r];
                        prow[1 * NT * NT + mirror] = acc[1][SLICED ? J : 0][It][r];
                        prow[2 * NT * NT + mirror] = acc[1][SLICED ? J : 0][It][r];
                    }
                } else if (low && (MODE == 1 || !PAIR)) {
                    prow[mirror] = acc[0][SLICED ? J : 0][It][r];
                } else {
                    prow[row * NT + col] = acc[0][Iw][J][r];
                }
            }
        }
    }
    if (pcounts) {
        n_keep = wave_sum_i(n_keep);
        n_rm = wave_sum_i(n_rm);
        if (lane == 0) { ldc[wave][0] = n_keep; ldc[wave][1] = n_rm; }      // non-evaluator waves carry zeros
        __syncthreads();
        if (threadIdx.x < 2) {
            const int v = ldc[0][threadIdx.x] + ldc[1][threadIdx.x] + ldc[2][threadIdx.x] + ldc[3][threadIdx.x];
            pcounts[(int64_t)blockIdx.x * 2 + threadIdx.x] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Diagonal 64 x 64 blocks (T = 4, one term window): wave-specialised variant that uses the symmetry of
// G2 = (D.S)^T (D.S) (and, at level 0, of G0 = F^T F and G1 = (F.F)^T (F.F); in MODE 1 of G = D^T D).
// Only the 10 upper tiles of a symmetric matrix are computed, spread 2 / 2 / 3 / 3 over the four waves; the mirrored
// tile is written with the partials.  MFMAs per 4 samples and wave: 10 / 10 / 11 / 11 instead of 12 (pair levels),
// 5 / 5 / 5 / 5 instead of 8 (level 0: the second matrix takes the lists in the order 3 / 3 / 2 / 2), 2 / 2 / 3 / 3 instead
// of 4 (MODE 1).  Waves 0-1 -- the ones with less matrix work -- run the evaluation phase.
// ------------------------------------------------------------------------------------------
// w0: (0,0) (0,1)   w1: (1,1) (1,2)   w2: (2,2) (2,3) (0,2)   w3: (3,3) (0,3) (1,3)
__host__ __device__ constexpr int sym_n(int w) { return w < 2 ? 2 : 3; }
__host__ __device__ constexpr int sym_i(int w, int t) { return t == 0 ? w : (w < 2 ? w : (w == 2 ? (t == 1 ? 2 : 0) : (t == 1 ? 0 : 1))); }
__host__ __device__ constexpr int sym_j(int w, int t) { return t == 0 ? w : (w < 2 ? w + 1 : (w == 2 ? (t == 1 ? 3 : 2) : 3)); }
// One evaluator wave (32-pair batches): it takes a single tile, w0: (0,0)   w1: (1,1) (1,2) (0,1)   w2: (2,2) (2,3) (0,2)
// w3: (3,3) (0,3) (1,3) -- 9 / 11 / 11 / 11 MFMAs per k-step at a pair level, the evaluation goes to the wave with 9.
__host__ __device__ constexpr int psym_n(int w) { return w == 0 ? 1 : 3; }
__host__ __device__ constexpr int psym_i(int w, int t) { return t == 0 ? w : (w == 1 ? (t == 1 ? 1 : 0) : (w == 2 ? (t == 1 ? 2 : 0) : (t == 1 ? 0 : 1))); }
__host__ __device__ constexpr int psym_j(int w, int t) { return t == 0 ? w : (w == 1 ? (t == 1 ? 2 : 1) : (w == 2 ? (t == 1 ? 3 : 2) : 3)); }
// 2 x 2 tile blocks (pair levels with variances): wave w = 2 a + b owns rows {2a, 2a+1} x columns {2b, 2b+1} of G0 and G1, and of
// the symmetric G2  w0: (0,0) (0,1) (1,1)   w1: (0,2) (0,3)   w2: (1,2) (1,3)   w3: (2,2) (2,3) (3,3) -- the diagonal waves then
// need the fine / coarse fragments of two row blocks only (4 LDS reads and 10 vector instructions per k-step instead of 8 and
// 15), the others 8 and 14; 11 / 10 / 10 / 11 MFMAs, wave 1 evaluates.
__host__ __device__ constexpr int bsym_n(int w) { return (w == 0 || w == 3) ? 3 : 2; }
__host__ __device__ constexpr int bsym_i(int w, int t) { return w == 0 ? (t == 2 ? 1 : 0) : (w == 1 ? 0 : (w == 2 ? 1 : (t == 2 ? 3 : 2))); }
__host__ __device__ constexpr int bsym_j(int w, int t) { return w == 0 ? (t == 0 ? 0 : 1) : (w == 1 ? 2 + t : (w == 2 ? 2 + t : (t == 0 ? 2 : 3))); }
// kind of list: 0 = two evaluators (2/2/3/3), 1 = one evaluator (1/3/3/3), 2 = 2 x 2 blocks (3/2/2/3)
__host__ __device__ constexpr int xsym_n(int kind, int w) { return kind == 2 ? bsym_n(w) : (kind == 1 ? psym_n(w) : sym_n(w)); }
__host__ __device__ constexpr int xsym_i(int kind, int w, int t) { return kind == 2 ? bsym_i(w, t) : (kind == 1 ? psym_i(w, t) : sym_i(w, t)); }
__host__ __device__ constexpr int xsym_j(int kind, int w, int t) { return kind == 2 ? bsym_j(w, t) : (kind == 1 ? psym_j(w, t) : sym_j(w, t)); }

template <int KIND, bool PAIR, int MODE, int BD, int W>
__device__ __forceinline__ void cov_t4_body(const BasisParams &bp, 
                                            const double *__restrict__ fine, const double *__restrict__ coarse,
                                            const uint8_t *__restrict__ mask, int64_t n, double *__restrict__ partials,
                                            int64_t *__restrict__ pcounts, double *__restrict__ lds_f,
                                            double *__restrict__ lds_c, int (*ldc)[2]) {
    constexpr int NT = 64;
    constexpr int TA = 64 * BD;
    constexpr int N_EVAL = TA + NT;
    constexpr bool BLK = PAIR && (MODE == 0 || MODE == 3) && MLMC_COV_BLK22 && COV_T4_BATCH == 32;   // 2 x 2 tile blocks per wave
    constexpr bool BLK2 = PAIR && MODE == 2 && MLMC_COV_BLK22 && COV_T4_BATCH == 32;  // the same for the mean-only G0
    constexpr int ONE = BLK ? 2 : ((PAIR && MLMC_COV_T4_ONE_LIST && COV_T4_BATCH == 32) ? 1 : 0);   // kind of tile list
    constexpr int NS = xsym_n(ONE, W);
    // level 0 with variances has TWO symmetric matrices: the second one takes the tile list of wave W + 2, so every wave
    // issues 2 + 3 = 5 MFMAs per k-step (the same list for both gave 6 / 6 / 4 / 4 and two waves waited at the barrier:
    // 1.88 -> 1.65 ms per 1e7 samples)
    constexpr int W1 = (MODE == 0 && !PAIR) ? (W + 2) % 4 : W;
    constexpr int NS1 = sym_n(W1);
    constexpr int NFULL = (MODE == 0 && PAIR) ? 2 : (((MODE == 2 || MODE == 3) && PAIR) ? 1 : 0);     // G0 (and G1): full row W
    constexpr int GV = MODE == 3 ? 0 : 1;          // accumulator set of G1
    constexpr int NSYMM = (MODE == 0 && !PAIR) ? 2 : 1;    // symmetric matrices handled through the tile list
    const int lane = threadIdx.x & 63;

    v4f64 accf[NFULL > 0 ? NFULL : 1][4];
    v4f64 accs[NSYMM][3];
#pragma unroll
    for (int g = 0; g < (NFULL > 0 ? NFULL : 1); ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) accf[g][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < NSYMM; ++g)
#pragma unroll
        for (int t = 0; t < 3; ++t) accs[g][t] = (v4f64){0.0, 0.0, 0.0, 0.0};

    // Batch: COV_T4_BATCH sample pairs, or twice as many samples at level 0 (one LDS array instead of two: the same bytes) --
    // COV_T4_BATCH / 32 evaluator waves either way (pairs: lane = (sample, fine | coarse); level 0: lane = sample), so the
    // evaluation phase is spread over the same share of the workgroup at both kinds of level.
    constexpr int BATCH = PAIR ? COV_T4_BATCH : 2 * COV_T4_BATCH;
    // row stride in doubles: == 2 (mod 32) makes the 64-bit fragment reads conflict-free; the 128-bit reads (16 lanes per LDS
    // cycle, four banks each) want == 4 (mod 32): 8 row + 4 g (mod 64) is then a different bank quad for every lane of a group
    constexpr int STRIDE = BATCH + MLMC_COV_B128_PAD;
    // The evaluators are the first waves: they own fewer tiles of a symmetric matrix than the others.  (Waves 2-3 as
    // evaluators of 64-pair batches: +5 % on a mean-only pair level, same-box A/B.  Which SIMD a wave runs on rotates from workgroup to
    // workgroup -- HW_ID histogram in tools/prof_cov.hip -- so no assignment balances the SIMDs of a CU exactly.)
    constexpr bool evaluator = BLK ? (W == 1) : (W < COV_T4_BATCH / 32);
    constexpr int EW = BLK ? 0 : W;              // index of this wave among the evaluators
    const int samp = PAIR ? (EW * 32 + (lane & 31)) : (EW * 64 + lane);
    const int psamp = cov_t4_pos(samp);          // where this sample's values go in the LDS rows
    const bool is_coarse = PAIR && (lane >> 5);
    const double *__restrict__ src = is_coarse ? coarse : fine;
    // (Measured and not adopted: writing d = f - c and s = f + c instead of f and c -- one half-wave exchange and one add per
    // term in the evaluator lanes save the four waves of phase 2 eight adds per k-step, but the exchange sits in the
    // evaluation's dependent chain: +9.6 % on a pair level, same-box A/B, tools/kbench.py.)
    double *__restrict__ dst = is_coarse ? lds_c : lds_f;
    int n_keep = 0, n_rm = 0;

    const int64_t n_batches = (n + BATCH - 1) / BATCH;
    int64_t batch = blockIdx.x;
    double xv = 0.0;
    uint8_t mv = 1;
    if (evaluator && batch < n_batches) {
        int64_t idx = batch * BATCH + samp;
        if (idx < n) { xv = src[idx]; if (mask) mv = mask[idx]; }
    }
#ifdef MLMC_PROF_COV
    unsigned long long prof_acc[4] = {0, 0, 0, 0}, prof_t = __builtin_amdgcn_s_memtime();
    const unsigned long long prof_t0 = prof_t, prof_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // rank of the workgroup among the COV_T4_WGS that share its CU: the dispatcher deals the workgroups of a full grid
    // (launch_cov_accum: COV_T4_WGS per CU) round-robin over the CUs, so the partners of block b are b + n_cu, b + 2 n_cu, ...
    // (checked with HW_ID in tools/prof_cov.hip); all four waves of a workgroup share the rank
    constexpr bool RANKS = (MODE == 0 || (MODE == 3 && MLMC_COV_VAR_RANKS)) && PAIR;      // rotating priorities (see MLMC_COV_MFMA_PRIO)
    const unsigned per_round = gridDim.x >= (unsigned)COV_T4_WGS ? gridDim.x / (unsigned)COV_T4_WGS : 1u;
    const unsigned prio_slot = (blockIdx.x / per_round) % (unsigned)COV_T4_WGS;
    MLMC_COV_EVAL_PRIO(RANKS);
    for (; batch < n_batches; batch += gridDim.x) {
        if (evaluator) {
            const int64_t idx = batch * BATCH + samp;
            const bool valid = idx < n;
            bool keep;
            double t = transform_value(bp, xv, keep);
            keep = keep && valid && (mv != 0);
            if (PAIR) {
                const int other = __shfl_xor((int)keep, 32, 64);
                keep = keep && (other != 0);
            }
            if (!is_coarse) { n_keep += (int)keep; n_rm += (int)(valid && !keep); }
            const int64_t nidx = (batch + gridDim.x) * BATCH + samp;
            if (nidx < n) { xv = src[nidx]; if (mask) mv = mask[nidx]; }
            TermGen<KIND> g;
            g.init(keep ? t : 0.0, keep ? 1.0 : 0.0, bp);
            if constexpr (KIND == MLMC_SPLINE) {
                cov_spline_store<NT, STRIDE, TA>(g, dst, psamp);
            } else {
#pragma unroll
                for (int i = 0; i < N_EVAL; ++i) {
                    const double q = g.next(i);
                    if (i >= TA) dst[(i - TA) * STRIDE + psamp] = q;
                }
            }
        }
        MLMC_COV_STAMP(0)
        const unsigned long long prio_clock = __builtin_amdgcn_s_memtime();   // arrives while the wave waits at the barrier
        __syncthreads();
        MLMC_COV_STAMP(1)
        MLMC_COV_MFMA_PRIO(RANKS, prio_slot, (unsigned)COV_T4_WGS, prio_clock);
        // (Measured and not adopted, same-box A/B: explicit register prefetch of the next two k-steps' fragments with the
        // instructions of two k-steps grouped as reads | vector | matrix by sched_group_barrier: -0.5 % with variances, +13 %
        // mean-only against the compiler's own interleaving.  A software pipeline with ONE workgroup per CU, two LDS images
        // and the next batch's recurrences spread over the k-steps of this one -- no evaluation phase, one barrier per batch
        // -- was 11 % slower at a pair level: a single wave per SIMD leaves every stall of the matrix stream exposed.  The same
        // pipeline with batches of 32 pairs, two 35 KB images per workgroup, two workgroups per CU and one evaluator wave
        // (9 / 11 / 11 / 11 MFMAs per k-step): +10 % with variances, +17 % mean-only -- the evaluator wave's in-order stream
        // of dependent recurrence steps between its matrix instructions paces the whole workgroup.  All levels of an estimate in
        // ONE launch -- every workgroup walks the levels one after the other, pair and level-0 bodies in one kernel, to pay
        // the ramp and the uneven end of a launch once instead of five times: +1.7 % (the combined kernel does not fit the 128
        // registers per lane of four workgroups per CU: 210 spilled VGPRs), and compiled in the same translation unit it
        // changed the register allocation of the stand-alone kernel, too; the four pair levels alone in one launch, from a
        // translation unit of its own: -0.3 % -- the end of a launch is not where the time goes.  A static schedule that gives the youngest
        // workgroup of a CU 15 batches where the others get 16: no change.)
        typedef double v2f64 __attribute__((ext_vector_type(2)));
        v2f64 f2[4], c2[4];
#pragma unroll
        for (int ks = 0; ks < BATCH / 4; ++ks) {
            const int col = 4 * ks + (lane >> 4);
            double d[4], sm[4];
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const int row = 16 * J + (lane & 15);
                double f, c = 0.0;
                if (MLMC_COV_B128) {
                    // (fully unrolled: `ks` is a constant) one 128-bit read per operand row and PAIR of k-steps
                    if ((ks & 1) == 0) {
                        const int base = row * STRIDE + 8 * (ks >> 1) + 2 * (lane >> 4);
                        f2[J] = *reinterpret_cast<const v2f64 *>(lds_f + base);
                        if (PAIR) c2[J] = *reinterpret_cast<const v2f64 *>(lds_c + base);
                    }
                    f = (ks & 1) ? f2[J].y : f2[J].x;
                    if (PAIR) c = (ks & 1) ? c2[J].y : c2[J].x;
                } else {
                    f = lds_f[row * STRIDE + col];
                    if (PAIR) c = lds_c[row * STRIDE + col];
                }
                d[J] = f;
                sm[J] = f;
                if (PAIR) {
                    d[J] = f - c;
                    sm[J] = f + c;
                }
            }
            if (BLK) {
                constexpr int RA = 2 * (W >> 1), CA = 2 * (W & 1);      // first row block, first column block
                double ds[4];
#pragma unroll
                for (int J = 0; J < 4; ++J) ds[J] = d[J] * sm[J];       // only the ones the tile list names survive
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const double dr2 = d[RA + r] * d[RA + r];
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (MODE == 0)
                            accf[0][2 * r + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[RA + r], sm[CA + c], accf[0][2 * r + c], 0, 0, 0);
                        accf[GV][2 * r + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(dr2, sm[CA + c] * sm[CA + c], accf[GV][2 * r + c], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < NS; ++t)
                    accs[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ds[xsym_i(ONE, W, t)], ds[xsym_j(ONE, W, t)], accs[0][t], 0, 0, 0);
            } else if ((MODE == 0 || MODE == 3) && PAIR) {
                const double dw2 = d[W] * d[W];
                double ds[4];
#pragma unroll
                for (int J = 0; J < 4; ++J) {
                    if (MODE == 0) accf[0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[W], sm[J], accf[0][J], 0, 0, 0);
                    accf[GV][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(dw2, sm[J] * sm[J], accf[GV][J], 0, 0, 0);
                    ds[J] = d[J] * sm[J];
                }
#pragma unroll
                for (int t = 0; t < NS; ++t)
                    accs[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(ds[xsym_i(ONE, W, t)], ds[xsym_j(ONE, W, t)], accs[0][t], 0, 0, 0);
            } else if (MODE == 0) {   // level 0: F^T F and (F.F)^T (F.F), both symmetric
                double f2[4];
#pragma unroll
                for (int J = 0; J < 4; ++J) f2[J] = d[J] * d[J];
#pragma unroll
                for (int t = 0; t < NS; ++t)
                    accs[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[xsym_i(ONE, W, t)], d[xsym_j(ONE, W, t)], accs[0][t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < NS1; ++t)
                    accs[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f2[sym_i(W1, t)], f2[sym_j(W1, t)], accs[1][t], 0, 0, 0);
            } else if (MODE == 3) {   // level 0, variance only: (F.F)^T (F.F), symmetric
                double f2[4];
#pragma unroll
                for (int J = 0; J < 4; ++J) f2[J] = d[J] * d[J];
#pragma unroll
                for (int t = 0; t < NS; ++t)
                    accs[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f2[xsym_i(ONE, W, t)], f2[xsym_j(ONE, W, t)], accs[0][t], 0, 0, 0);
            } else if (MODE == 2 && PAIR && BLK2) {   // covariance mean only: G0 = D^T S, a 2 x 2 block of tiles per wave
                constexpr int RA = 2 * (W >> 1), CA = 2 * (W & 1);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        accf[0][2 * r + c] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[RA + r], sm[CA + c], accf[0][2 * r + c], 0, 0, 0);
            } else if (MODE == 2 && PAIR) {   // covariance mean only: G0 = D^T S, row tile W
#pragma unroll
                for (int J = 0; J < 4; ++J)
                    accf[0][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[W], sm[J], accf[0][J], 0, 0, 0);
            } else {                  // MODE 1: D^T D;  MODE 2 at level 0: F^T F (d = f)
#pragma unroll
                for (int t = 0; t < NS; ++t)
                    accs[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(d[xsym_i(ONE, W, t)], d[xsym_j(ONE, W, t)], accs[0][t], 0, 0, 0);
            }
        }
        MLMC_COV_EVAL_PRIO(RANKS);
        MLMC_COV_STAMP(2)
        __syncthreads();
        MLMC_COV_STAMP(3)
    }
#ifdef MLMC_PROF_COV
    if (lane == 0 && blockIdx.y == 0) {
        unsigned long long *q = g_prof_cov + ((size_t)blockIdx.x * 4 + W) * 6;
        q[0] = prof_acc[0]; q[1] = prof_acc[1]; q[2] = prof_acc[2]; q[3] = prof_acc[3];
        q[4] = __builtin_amdgcn_s_memtime() - prof_t0;
        q[5] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xffffu) |   // HW_ID: SIMD in bits 5:4
               ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 16) |      // XCC_ID
               ((__builtin_amdgcn_s_memrealtime() - prof_r0) << 24);                                           // 100 MHz ticks
    }
#endif

    // ---- partial tiles: one row per block, columns [g][row][col]; symmetric tiles are mirrored here ----
    constexpr int NGOUT = cov_ng(MODE);
    double *__restrict__ prow = partials + (int64_t)blockIdx.x * (NGOUT * NT * NT);
    const int r0 = lane >> 4, c0 = lane & 15;
    if (MODE == 3 && PAIR) {        // [G1][G2]
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tr = BLK ? 2 * (W >> 1) + (J >> 1) : W, tc = BLK ? 2 * (W & 1) + (J & 1) : J;
                prow[(16 * tr + r0 + 4 * r) * NT + 16 * tc + c0] = accf[0][J][r];
            }
    }
    if (MODE == 0 && PAIR) {
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // row-of-tiles mapping: tile (W, J); 2 x 2 blocks: accumulator J = 2 r' + c' is tile (2 (W >> 1) + r', 2 (W & 1) + c')
                const int tr = BLK ? 2 * (W >> 1) + (J >> 1) : W, tc = BLK ? 2 * (W & 1) + (J & 1) : J;
                const int row = 16 * tr + r0 + 4 * r, col = 16 * tc + c0;
                prow[0 * NT * NT + row * NT + col] = accf[0][J][r];
                prow[1 * NT * NT + row * NT + col] = accf[1][J][r];
            }
    }
    if (MODE == 2 && PAIR) {
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tr = BLK2 ? 2 * (W >> 1) + (J >> 1) : W, tc = BLK2 ? 2 * (W & 1) + (J & 1) : J;
                prow[(16 * tr + r0 + 4 * r) * NT + 16 * tc + c0] = accf[0][J][r];
            }
    }
#pragma unroll
    for (int t = 0; t < ((MODE == 2 && PAIR) ? 0 : NS); ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * xsym_i(ONE, W, t) + r0 + 4 * r, col = 16 * xsym_j(ONE, W, t) + c0;
            if (MODE == 0 && PAIR) {
                prow[2 * NT * NT + row * NT + col] = accs[0][t][r];
                if (xsym_i(ONE, W, t) != xsym_j(ONE, W, t)) prow[2 * NT * NT + col * NT + row] = accs[0][t][r];
            } else if (MODE == 3) {   // pair levels: G2 = (D.S)^T (D.S) behind G1; level 0: G1 = G2 = (F.F)^T (F.F)
                const bool off = xsym_i(ONE, W, t) != xsym_j(ONE, W, t);
                prow[NT * NT + row * NT + col] = accs[0][t][r];
                if (off) prow[NT * NT + col * NT + row] = accs[0][t][r];
                if (!PAIR) {
                    prow[row * NT + col] = accs[0][t][r];
                    if (off) prow[col * NT + row] = accs[0][t][r];
                }
            } else if (MODE == 0) {   // level 0: G0 = F^T F (G1 = G2 below)
                prow[0 * NT * NT + row * NT + col] = accs[0][t][r];
                if (xsym_i(ONE, W, t) != xsym_j(ONE, W, t)) prow[0 * NT * NT + col * NT + row] = accs[0][t][r];
            } else {
                prow[row * NT + col] = accs[0][t][r];
                if (xsym_i(ONE, W, t) != xsym_j(ONE, W, t)) prow[col * NT + row] = accs[0][t][r];
            }
        }
    if (MODE == 0 && !PAIR) {   // level 0: G1 = G2 = (F.F)^T (F.F), tile list of wave W1
#pragma unroll
        for (int t = 0; t < NS1; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * sym_i(W1, t) + r0 + 4 * r, col = 16 * sym_j(W1, t) + c0;
                prow[1 * NT * NT + row * NT + col] = accs[1][t][r];
                prow[2 * NT * NT + row * NT + col] = accs[1][t][r];
                if (sym_i(W1, t) != sym_j(W1, t)) {
                    prow[1 * NT * NT + col * NT + row] = accs[1][t][r];
                    prow[2 * NT * NT + col * NT + row] = accs[1][t][r];
                }
            }
    }
    if (pcounts) {
        n_keep = wave_sum_i(n_keep);
        n_rm = wave_sum_i(n_rm);
        if (lane == 0 && W < 2) { ldc[W][0] = n_keep; ldc[W][1] = n_rm; }      // both evaluator waves (the coarse lanes carry zeros)
        __syncthreads();
        if (threadIdx.x < 2) pcounts[(int64_t)blockIdx.x * 2 + threadIdx.x] = ldc[0][threadIdx.x] + ldc[1][threadIdx.x];
    }
}

template <int KIND, bool PAIR, int MODE, int BD>
__global__ __launch_bounds__(256, COV_T4_WGS) void k_cov_accum_t4(BasisParams bp, 
                                                         const double *__restrict__ fine, const double *__restrict__ coarse,
                                                         const uint8_t *__restrict__ mask, int64_t n, int R,
                                                         double *__restrict__ partials, int64_t *__restrict__ pcounts) {
    __shared__ __attribute__((aligned(16))) double lds_f[64 * ((PAIR ? COV_T4_BATCH : 2 * COV_T4_BATCH) + MLMC_COV_B128_PAD)];   // term-major fine values (level 0: 128 samples)
    __shared__ __attribute__((aligned(16))) double lds_c[PAIR ? 64 * (COV_T4_BATCH + MLMC_COV_B128_PAD) : 1];                  // coarse values
    __shared__ int ldc[2][2];
    (void)R;
    // blockIdx.y = component of a vector quantity (see k_cov_accum): one partial row [3 or 1][64][64] per workgroup
    fine += (int64_t)blockIdx.y * n;
    if (PAIR) coarse += (int64_t)blockIdx.y * n;
    partials += (int64_t)blockIdx.y * gridDim.x * (cov_ng(MODE) * 64 * 64);
    if (blockIdx.y) pcounts = nullptr;
    // every wave runs its own specialisation (same barrier count in all of them); readfirstlane makes the wave index a scalar,
    // so this is a scalar branch -- as a per-lane switch the compiler predicates the four bodies with exec masks, and a
    // wave must never walk through another body's s_barrier with an empty mask
    switch (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) {
        case 0: cov_t4_body<KIND, PAIR, MODE, BD, 0>(bp, fine, coarse, mask, n, partials, pcounts, lds_f, lds_c, ldc); break;
        case 1: cov_t4_body<KIND, PAIR, MODE, BD, 1>(bp, fine, coarse, mask, n, partials, pcounts, lds_f, lds_c, ldc); break;
        case 2: cov_t4_body<KIND, PAIR, MODE, BD, 2>(bp, fine, coarse, mask, n, partials, pcounts, lds_f, lds_c, ldc); break;
        default: cov_t4_body<KIND, PAIR, MODE, BD, 3>(bp, fine, coarse, mask, n, partials, pcounts, lds_f, lds_c, ldc); break;
    }
}

// totals[g][row][col] (leading dimension RP) += sum over partial rows, fixed order.  16 columns per workgroup (one 128-byte
// line of every partial row), 64 row groups: a three-Gram 16 x 16 tile set already gives 48 workgroups (64 columns per
// workgroup left a 24-component quantity of small chunks waiting on 12 of them).
__global__ __launch_bounds__(1024) void k_reduce_cov(const double *__restrict__ partials, int nrows, int NT, int NG, int RP,
                                                    int roff, int coff, double *__restrict__ totals, int64_t comp_stride,
                                                    int mirror, const int64_t *__restrict__ pcounts, int nblocks,
                                                    int64_t *__restrict__ counts) {
    __shared__ double lds[64][17];
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int width = NG * NT * NT;
    if (pcounts && blockIdx.x == gridDim.x - 1) {      // one more workgroup: the sample counts of the launch (exact integer sums)
        if (blockIdx.y == 0 && threadIdx.x < 64) {
            int64_t a = 0, b = 0;
            for (int i = threadIdx.x; i < nblocks; i += 64) { a += pcounts[2 * i]; b += pcounts[2 * i + 1]; }
            for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
            if (threadIdx.x == 0) { counts[0] += a; counts[1] += b; }
        }
        return;
    }
    partials += (int64_t)blockIdx.y * nrows * width;       // blockIdx.y = component: its partial rows, its totals
    totals += (int64_t)blockIdx.y * comp_stride;
    const int col = blockIdx.x * 16 + c;
    double acc = 0.0;
    if (col < width)
        for (int b = g; b < nrows; b += 64) acc += partials[(int64_t)b * width + col];
    lds[g][c] = acc;
    __syncthreads();
    if (g == 0 && col < width) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 64; ++k) v += lds[k][c];
        const int gi = col / (NT * NT), rem = col % (NT * NT);
        const int row = roff + rem / NT, cc = coff + rem % NT;
        if (row < RP && cc < RP) {
            totals[(int64_t)gi * RP * RP + (int64_t)row * RP + cc] += v;
            // level 0: every matrix is symmetric, the block below the diagonal was not computed -- this one transposed stands in
            if (mirror) totals[(int64_t)gi * RP * RP + (int64_t)cc * RP + row] += v;
        }
    }
}

template <int KIND, int T, int MODE, int BI, int BJ>
static int launch_cov_t(const BasisParams &bp, bool pair, int blocks, int ncomp, const double *d_f, const double *d_c,
                        const uint8_t *d_mask, int64_t n, int R, double *partials, int64_t *pcounts) {
    hipStream_t st = rt().stream;
    if constexpr (T == 4 && BI == BJ) {   // diagonal 64 x 64 block: wave-specialised kernel (symmetric tiles used where they exist)
        if (pair)
            hipLaunchKernelGGL((k_cov_accum_t4<KIND, true, MODE, BI>), dim3(blocks, ncomp), dim3(256), 0, st, bp, d_f, d_c, d_mask, n, R, partials, pcounts);
        else
            hipLaunchKernelGGL((k_cov_accum_t4<KIND, false, MODE, BI>), dim3(blocks, ncomp), dim3(256), 0, st, bp, d_f, d_c, d_mask, n, R, partials, pcounts);
    } else {            // (in the else branch: the generic kernel is not instantiated for the combinations above)
        if (pair)
            hipLaunchKernelGGL((k_cov_accum<KIND, T, true, MODE, BI, BJ>), dim3(blocks, ncomp), dim3(256), 0, st, bp, d_f, d_c, d_mask, n, R, partials, pcounts, 0, 0);
        else
            hipLaunchKernelGGL((k_cov_accum<KIND, T, false, MODE, BI, BJ>), dim3(blocks, ncomp), dim3(256), 0, st, bp, d_f, d_c, d_mask, n, R, partials, pcounts, 0, 0);
    }
    MLMC_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int KIND, int MODE>
static int launch_cov_kind(const BasisParams &bp, int T, int bi, int bj, bool pair, int blocks, int ncomp, const double *d_f,
                           const double *d_c, const uint8_t *d_mask, int64_t n, int R, double *partials, int64_t *pcounts) {
#define MLMC_COV_ARGS bp, pair, blocks, ncomp, d_f, d_c, d_mask, n, R, partials, pcounts
    if (T == 1) return launch_cov_t<KIND, 1, MODE, 0, 0>(MLMC_COV_ARGS);
    if (T == 2) return launch_cov_t<KIND, 2, MODE, 0, 0>(MLMC_COV_ARGS);
    if (bi == 0 && bj == 0) return launch_cov_t<KIND, 4, MODE, 0, 0>(MLMC_COV_ARGS);
    if (bi == 0 && bj == 1) return launch_cov_t<KIND, 4, MODE, 0, 1>(MLMC_COV_ARGS);
    if (bi == 1 && bj == 0) return launch_cov_t<KIND, 4, MODE, 1, 0>(MLMC_COV_ARGS);
    if (bi == 1 && bj == 1) return launch_cov_t<KIND, 4, MODE, 1, 1>(MLMC_COV_ARGS);
#undef MLMC_COV_ARGS
    return fail("covariance: at most 128 moments");
}

// Covariance accumulation from materialised moment values [n][R1]: the generic kernel with VALS.  The path of moment
// functions that are not evaluated in registers -- TransformedMoments (any number of rows) and plain bases beyond the 128
// moments the compile-time term windows cover (up to the 512 the library accepts): the reference has no size limit
// (quantity_estimate.py:122-156).  More than 64 moments: 64 x 64 output blocks, the window offsets are kernel arguments.
// gram_mode 1: the difference Gram matrix D^T D of the UNDERLYING moments (values [n][a->R] in the accumulators' scaling) into
// the Gram slot of a MOMENTS accumulator of TransformedMoments (variance = diag(T G T^T)): symmetric, upper blocks only.
int launch_cov_from_values(mlmc_accum *a, int level, int comp, const double *d_vf, const double *d_vc, const uint8_t *d_mask,
                           int64_t n, bool count, int gram_mode) {
    if (n == 0) return 0;
    const bool diff_gram = gram_mode == 1;
    const int R = diff_gram ? a->R : a->Rout;
    hipStream_t st = rt().stream;
    const int T = (R <= 16) ? 1 : (R <= 32 ? 2 : 4);
    const int NT = 16 * T, NSL = 4 / T;
    const bool pair = d_vc != nullptr;
    const int NB = (R + 63) / 64;
    const int64_t n_batches = (n + COV_BATCH - 1) / COV_BATCH;
    const int NG = diff_gram ? 1 : 3;
    const size_t width = (size_t)NG * NT * NT;
    const BasisParams &bp = a->basis->p;
    double *totals = a->d_totals + ((int64_t)level * a->n_comp + comp) * a->int_width + (diff_gram ? 2 * (int64_t)R : 0);
    const bool symmetric = !pair || diff_gram;
    for (int bi = 0; bi < NB; ++bi)
        for (int bj = 0; bj < NB; ++bj) {
            if (symmetric && bj < bi) continue;   // symmetric matrices (level 0; D^T D): block (bj, bi) is mirrored by the reduction
            const bool wide = bi != bj;
            int blocks = rt().n_cu * (wide ? 1 : 2);          // two term windows: 135 KB of LDS, one workgroup per CU
            if (n_batches < blocks) blocks = (int)n_batches;
            const int n_slices = T <= 2 ? 4 : NSL;             // partial rows per workgroup (k_cov_accum: SLICED)
            if (int rc = ensure((void **)&a->d_partials, &a->partials_cap, sizeof(double) * (size_t)blocks * n_slices * width)) return rc;
            if (int rc = ensure((void **)&a->d_pcounts, &a->pcounts_cap, sizeof(int64_t) * (size_t)blocks * 2)) return rc;
            const bool do_count = count && bi == 0 && bj == 0 && !diff_gram;
            int64_t *pc = do_count ? a->d_pcounts : nullptr;
            if (int rc = timing_begin(a)) return rc;
#define MLMC_COV_VALS1(TT, BJJ, MODE)                                                                                           \
    do {                                                                                                                        \
        if (pair)                                                                                                               \
            hipLaunchKernelGGL((k_cov_accum<MLMC_IDENTITY, TT, true, MODE, 0, BJJ, true>), dim3(blocks), dim3(256), 0, st, bp,      \
                               d_vf, d_vc, d_mask, n, R, a->d_partials, pc, 64 * bi, 64 * bj);                                  \
        else                                                                                                                    \
            hipLaunchKernelGGL((k_cov_accum<MLMC_IDENTITY, TT, false, MODE, 0, BJJ, true>), dim3(blocks), dim3(256), 0, st, bp,     \
                               d_vf, d_vc, d_mask, n, R, a->d_partials, pc, 64 * bi, 64 * bj);                                  \
    } while (0)
#define MLMC_COV_VALS(TT, BJJ)                                                                                                  \
    do {                                                                                                                        \
        if (diff_gram) MLMC_COV_VALS1(TT, BJJ, 1); else MLMC_COV_VALS1(TT, BJJ, 0);                                             \
    } while (0)
            if (T == 1) MLMC_COV_VALS(1, 0);
            else if (T == 2) MLMC_COV_VALS(2, 0);
            else if (!wide) MLMC_COV_VALS(4, 0);
            else MLMC_COV_VALS(4, 1);
#undef MLMC_COV_VALS
#undef MLMC_COV_VALS1
            MLMC_HIP_CHECK(hipGetLastError());
            if (int rc = timing_end(a)) return rc;
            if (!diff_gram) {
                a->launches += 1;
                a->alg_bytes += (int64_t)n * (pair ? 16 : 8);
                // the generic kernel's tile lists: no symmetric savings inside a block except the SLICED (T <= 2) upper tiles
                a->mfma_flops += (int64_t)512 * n * (T <= 2 ? (pair ? 2 * T * T + T * (T + 1) / 2 : T * (T + 1)) : (pair ? 3 : 2) * 16);
            }
            hipLaunchKernelGGL(k_reduce_cov, dim3((unsigned)((width + 15) / 16) + (do_count ? 1u : 0u)), dim3(1024), 0, st, a->d_partials,
                               blocks * n_slices, NT, NG, a->RP, 64 * bi, 64 * bj, totals, (int64_t)0, (symmetric && wide) ? 1 : 0,
                               do_count ? a->d_pcounts : (const int64_t *)nullptr, blocks, a->d_counts + 2 * (int64_t)level);
            MLMC_HIP_CHECK(hipGetLastError());
        }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Mean of the covariance of SPLINE moments without the matrix cores.  At most four cubic B-splines are non-zero at a point, so
// phi_i phi_j vanishes identically for |i - j| > 3 (i, j >= 1; phi_0 = 1 gives row / column 0 = the plain moments): the R x R
// level sums are a 7-diagonal band plus one row, 14 numbers per value instead of a dense 128 x 128 Gram update.  Like
// k_spline_accum every wave keeps a private table in LDS -- [r][delta] = sum B_r B_(r+delta), delta = 0..3, and [r][4] = sum B_r --
// and adds to it with ds_add_f64 (fine values +, coarse values -; both in one span: the differences, 14 atomics per pair);
// the four tables are summed in a fixed order, the grid reduction scatters the band into the dense G0 (both triangles).
// 1.25e7 samples x 128 moments, mean only: the dense MFMA pass took 7.1 ms.
// ------------------------------------------------------------------------------------------
constexpr int BAND_W = 5;
template <bool PAIR>
__global__ __launch_bounds__(256) void k_spline_band_accum(BasisParams bp, const double *__restrict__ fine,
                                                          const double *__restrict__ coarse, const uint8_t *__restrict__ mask,
                                                          int64_t n, int R, double *__restrict__ partials,
                                                          int64_t *__restrict__ pcounts) {
    extern __shared__ double band_tab[];           // [4 waves][R + 8][BAND_W]
    __shared__ int ldc[4][2];
    const int RP8 = R + 8;
    fine += (int64_t)blockIdx.y * n;
    if (PAIR) coarse += (int64_t)blockIdx.y * n;
    partials += (int64_t)blockIdx.y * gridDim.x * (BAND_W * R);
    if (blockIdx.y) pcounts = nullptr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * RP8 * BAND_W; i += 256) band_tab[i] = 0.0;
    __syncthreads();
    double *__restrict__ tab = band_tab + (size_t)wave * RP8 * BAND_W;
    int n_keep = 0, n_rm = 0;
    const int64_t T = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += T) {
        bool kf, kc = true;
        const double tf = transform_value(bp, fine[i], kf);
        double tc = 0.0;
        if (PAIR) tc = transform_value(bp, coarse[i], kc);
        const bool keep = kf && kc && (!mask || mask[i] != 0);
        n_keep += (int)keep;
        n_rm += (int)!keep;
        if (!keep) continue;
        TermGen<MLMC_SPLINE> gf, gc;
        gf.init(tf, 1.0, bp);
        const double nf[4] = {gf.n0, gf.n1, gf.n2, gf.n3};
        double nc[4] = {0.0, 0.0, 0.0, 0.0};
        bool same = false;
        if (PAIR) {
            gc.init(tc, 1.0, bp);
            nc[0] = gc.n0; nc[1] = gc.n1; nc[2] = gc.n2; nc[3] = gc.n3;
            same = gc.k == gf.k;
        }
        // fine value (and, in the same span, minus the coarse value)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = gf.k + j;
            if (r < 1) continue;                                // B_0 is replaced by phi_0 = 1
            double *__restrict__ row = tab + (size_t)r * BAND_W;
            atomicAdd(&row[4], same ? nf[j] - nc[j] : nf[j]);
#pragma unroll
            for (int j2 = j; j2 < 4; ++j2)
                atomicAdd(&row[j2 - j], same ? __builtin_fma(nf[j], nf[j2], -(nc[j] * nc[j2])) : nf[j] * nf[j2]);
        }
        if (PAIR && !same) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = gc.k + j;
                if (r < 1) continue;
                double *__restrict__ row = tab + (size_t)r * BAND_W;
                atomicAdd(&row[4], -nc[j]);
#pragma unroll
                for (int j2 = j; j2 < 4; ++j2) atomicAdd(&row[j2 - j], -(nc[j] * nc[j2]));
            }
        }
    }
    n_keep = wave_sum_i(n_keep);
    n_rm = wave_sum_i(n_rm);
    if (lane == 0) { ldc[wave][0] = n_keep; ldc[wave][1] = n_rm; }
    __syncthreads();
    for (int j = threadIdx.x; j < BAND_W * R; j += 256) {
        const int o = j;                                         // [r][w], r < R
        partials[(int64_t)blockIdx.x * (BAND_W * R) + j] =
            ((band_tab[o] + band_tab[(size_t)RP8 * BAND_W + o]) + band_tab[(size_t)2 * RP8 * BAND_W + o]) + band_tab[(size_t)3 * RP8 * BAND_W + o];
    }
    if (pcounts && threadIdx.x < 2)
        pcounts[(int64_t)blockIdx.x * 2 + threadIdx.x] = ldc[0][threadIdx.x] + ldc[1][threadIdx.x] + ldc[2][threadIdx.x] + ldc[3][threadIdx.x];
}

// Band partials [nblocks][R][BAND_W] (per component) -> dense G0 [RP][RP]: fixed-order sums (16 columns per workgroup, 64 row
// groups, like k_reduce_cov), both triangles; row / column 0 from the plain sums; G0[0][0] += kept samples at level 0
// (phi_0 phi_0 = 1; the differences of a pair level vanish).  The last workgroup in x sums the sample counts.
__global__ __launch_bounds__(1024) void k_reduce_band(const double *__restrict__ partials, const int64_t *__restrict__ pcounts, int nblocks,
                                                     int R, int RP, double *__restrict__ totals, int64_t comp_stride,
                                                     int64_t *__restrict__ counts, int corner) {
    __shared__ double lds[64][17];
    __shared__ long long lc[16][2];
    const int width = BAND_W * R;
    const int n_col_groups = (width + 15) / 16;
    totals += (int64_t)blockIdx.y * comp_stride;
    if ((int)blockIdx.x == n_col_groups) {          // the counting workgroup: kept / removed samples of this launch
        long long a = 0, b = 0;
        for (int i = threadIdx.x; i < nblocks; i += 1024) { a += pcounts[2 * i]; b += pcounts[2 * i + 1]; }
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
        if ((threadIdx.x & 63) == 0) { lc[threadIdx.x >> 6][0] = a; lc[threadIdx.x >> 6][1] = b; }
        __syncthreads();
        if (threadIdx.x == 0) {
            a = 0; b = 0;
            for (int w = 0; w < 16; ++w) { a += lc[w][0]; b += lc[w][1]; }
            if (counts && blockIdx.y == 0) { counts[0] += a; counts[1] += b; }
            if (corner) totals[0] += (double)a;
        }
        return;
    }
    partials += (int64_t)blockIdx.y * nblocks * width;
    const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c;
    double acc = 0.0;
    if (col < width)
        for (int b = g; b < nblocks; b += 64) acc += partials[(int64_t)b * width + col];
    lds[g][c] = acc;
    __syncthreads();
    if (g == 0 && col < width) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 64; ++k) v += lds[k][c];
        const int r = col / BAND_W, w = col % BAND_W;
        if (r >= 1) {
            if (w == 4) {
                totals[r] += v;
                totals[(int64_t)r * RP] += v;
            } else if (r + w < R) {
                totals[(int64_t)r * RP + r + w] += v;
                if (w) totals[(int64_t)(r + w) * RP + r] += v;
            }
        }
    }
}

static int launch_spline_band(mlmc_accum *a, int level, int comp, const double *d_f, const double *d_c, const uint8_t *d_mask, int64_t n,
                              bool count, int ncomp) {
    hipStream_t st = rt().stream;
    const int R = a->R;
    const bool pair = d_c != nullptr;
    int blocks = rt().n_cu * 4;
    const int64_t want = (n + 255) / 256;
    if (want < blocks) blocks = (int)want;
    if (int rc = ensure((void **)&a->d_partials, &a->partials_cap, sizeof(double) * (size_t)blocks * BAND_W * R * ncomp)) return rc;
    if (int rc = ensure((void **)&a->d_pcounts, &a->pcounts_cap, sizeof(int64_t) * (size_t)blocks * 2)) return rc;
    const size_t lds = sizeof(double) * 4 * (size_t)(R + 8) * BAND_W;
    const BasisParams &bp = a->basis->p;
    double *totals = a->d_totals + ((int64_t)level * a->n_comp + comp) * a->int_width;
    if (int rc = timing_begin(a)) return rc;
    // the kept count is needed for G0[0][0] at level 0 even when another kernel (k_mask) has counted the samples already
    if (pair)
        hipLaunchKernelGGL(k_spline_band_accum<true>, dim3(blocks, ncomp), dim3(256), lds, st, bp, d_f, d_c, d_mask, n, R, a->d_partials, a->d_pcounts);
    else
        hipLaunchKernelGGL(k_spline_band_accum<false>, dim3(blocks, ncomp), dim3(256), lds, st, bp, d_f, d_c, d_mask, n, R, a->d_partials, a->d_pcounts);
    MLMC_HIP_CHECK(hipGetLastError());
    if (int rc = timing_end(a)) return rc;
    a->launches += 1;
    a->alg_bytes += (int64_t)n * (pair ? 16 : 8) * ncomp;
    hipLaunchKernelGGL(k_reduce_band, dim3((BAND_W * R + 15) / 16 + 1, ncomp), dim3(1024), 0, st, a->d_partials, a->d_pcounts, blocks, R, a->RP,
                       totals, a->int_width, count ? a->d_counts + 2 * (int64_t)level : (int64_t *)nullptr, pair ? 0 : 1);
    MLMC_HIP_CHECK(hipGetLastError());
    return 0;
}

// 16 x 16 output tiles one sample (pair) contributes to in one block launch -- the kernels' tile lists above:
// gram_mode 0 = G0, G1, G2 (level 0: two symmetric matrices), 1 = D^T D (symmetric), 2 = G0 only (level 0: F^T F, symmetric),
// 3 = G1, G2 (level 0: one symmetric matrix).
// x 512 = executed matrix-core flops per sample (mlmc_accum_kernel_flops).
static int cov_tiles_per_sample(int T, bool diagonal, bool pair, int gram_mode) {
    const int full = T * T, upper = T * (T + 1) / 2;
    if (!diagonal) return (gram_mode == 0 ? (pair ? 3 : 2) : (gram_mode == 3 ? (pair ? 2 : 1) : 1)) * full;      // two term windows: no symmetry inside the block
    if (gram_mode == 0) return pair ? 2 * full + upper : 2 * upper;
    if (gram_mode == 3) return pair ? full + upper : upper;
    if (gram_mode == 1 || !pair) return upper;
    return full;
}

int launch_cov_accum(mlmc_accum *a, int level, int comp, const double *d_f, const double *d_c, const uint8_t *d_mask,
                     int64_t n, bool count, int gram_mode, int ncomp) {
    // ncomp > 1: components comp .. comp + ncomp - 1 of a vector quantity ([M][n] arrays) in ONE launch (grid.y)
    const bool diff_gram_only = gram_mode == 1;   // gram_mode: 0 = G0, G1, G2; 1 = D^T D into the moments' Gram slot; 2 = G0 only
    if (n == 0) return 0;
    const int R = a->R;
    if (a->basis->p.kind == MLMC_SPLINE && gram_mode == 2 && R <= SPLINE_BAND_MAX_R)
        return launch_spline_band(a, level, comp, d_f, d_c, d_mask, n, count, ncomp);      // banded: no matrix cores needed
    if (R > 128) return fail("covariance accumulation in registers covers 128 moments; larger bases go through launch_cov_from_values");
    hipStream_t st = rt().stream;
    const int T = (R <= 16) ? 1 : (R <= 32 ? 2 : 4);
    const int NT = 16 * T, NSL = 4 / T;
    const int NB = (R + 63) / 64;          // 64 x 64 output blocks per dimension
    const bool pair = d_c != nullptr;
    const int64_t bsz = T == 4 ? (pair ? COV_T4_BATCH : 2 * COV_T4_BATCH) : cov_batch(T, false, false, pair);
    const int64_t n_batches = (n + bsz - 1) / bsz;
    const BasisParams &bp = a->basis->p;
    double *totals0 = a->d_totals + ((int64_t)level * a->n_comp + comp) * a->int_width + (diff_gram_only ? 2 * (int64_t)R : 0);
    for (int bi = 0; bi < NB; ++bi)
        for (int bj = 0; bj < NB; ++bj) {
            if (!pair && bj < bi) continue;   // level 0: symmetric matrices, block (bj, bi) is mirrored by the reduction
            // variance only (gram_mode 3): the 64-term kernel of the diagonal blocks leaves [G1][G2] in its partial rows, the
            // generic kernel keeps the three-matrix layout with an empty G0
            const bool two_slots = gram_mode == 3 && T == 4 && bi == bj;
            const int NG = gram_mode == 3 ? (two_slots ? 2 : 3) : cov_ng(gram_mode);
            const size_t width = (size_t)NG * NT * NT;
            double *totals = totals0 + (two_slots ? (int64_t)a->RP * a->RP : 0);
            // workgroups per CU: one for the two-window blocks (135 KB of LDS), four for a single 16-term tile (33 KB each:
            // -11..15 % time against two), two otherwise
            int blocks = rt().n_cu * ((bi != bj) ? ((pair && MLMC_COV_WIDE_BATCH < 64) ? 2 : 1) : (T == 1 ? 4 : (T == 4 ? COV_T4_WGS : 2)));
            // small chunks: at least four batches per workgroup -- every workgroup leaves NSL partial matrices behind and the
            // reduction reads them all (a 24-component quantity of 10^5 samples spent more time there than in the MFMAs)
            if ((n_batches + 3) / 4 < blocks) blocks = (int)((n_batches + 3) / 4);
            const int n_slices = T <= 2 ? 4 : NSL;     // partial rows per workgroup (k_cov_accum: SLICED)
            if (int rc = ensure((void **)&a->d_partials, &a->partials_cap, sizeof(double) * (size_t)blocks * n_slices * width * ncomp)) return rc;
            if (int rc = ensure((void **)&a->d_pcounts, &a->pcounts_cap, sizeof(int64_t) * (size_t)blocks * 2)) return rc;
            const bool do_count = count && bi == 0 && bj == 0;
            int64_t *pc = do_count ? a->d_pcounts : nullptr;
            const bool timed = !diff_gram_only;
            if (timed) if (int rc = timing_begin(a)) return rc;
            int rc;
#define MLMC_COV_DISPATCH(KIND)                                                                                               \
    rc = gram_mode == 3 ? launch_cov_kind<KIND, 3>(bp, T, bi, bj, pair, blocks, ncomp, d_f, d_c, d_mask, n, R, a->d_partials, pc) \
       : gram_mode == 2 ? launch_cov_kind<KIND, 2>(bp, T, bi, bj, pair, blocks, ncomp, d_f, d_c, d_mask, n, R, a->d_partials, pc) \
       : diff_gram_only ? launch_cov_kind<KIND, 1>(bp, T, bi, bj, pair, blocks, ncomp, d_f, d_c, d_mask, n, R, a->d_partials, pc) \
                        : launch_cov_kind<KIND, 0>(bp, T, bi, bj, pair, blocks, ncomp, d_f, d_c, d_mask, n, R, a->d_partials, pc)
#define MLMC_COV_DISPATCH012(KIND)                                                                                            \
    rc = gram_mode == 2 ? launch_cov_kind<KIND, 2>(bp, T, bi, bj, pair, blocks, ncomp, d_f, d_c, d_mask, n, R, a->d_partials, pc) \
       : diff_gram_only ? launch_cov_kind<KIND, 1>(bp, T, bi, bj, pair, blocks, ncomp, d_f, d_c, d_mask, n, R, a->d_partials, pc) \
                        : launch_cov_kind<KIND, 0>(bp, T, bi, bj, pair, blocks, ncomp, d_f, d_c, d_mask, n, R, a->d_partials, pc)
            if (gram_mode == 3 && bp.kind != MLMC_LEGENDRE && bp.kind != MLMC_MONOMIAL)
                return fail("covariance: the variance-only pass exists for the polynomial families");
            switch (bp.kind) {
                case MLMC_LEGENDRE: MLMC_COV_DISPATCH(MLMC_LEGENDRE); break;
                case MLMC_MONOMIAL: MLMC_COV_DISPATCH(MLMC_MONOMIAL); break;
                case MLMC_FOURIER: MLMC_COV_DISPATCH012(MLMC_FOURIER); break;      // (no product linearisation here: no MODE 3 code)
                case MLMC_SPLINE: MLMC_COV_DISPATCH012(MLMC_SPLINE); break;
                default: return fail("covariance: unsupported basis kind");
            }
#undef MLMC_COV_DISPATCH
#undef MLMC_COV_DISPATCH012
            if (rc) return rc;
            if (timed) {
                if (int rc2 = timing_end(a)) return rc2;
                a->launches += 1;
                a->alg_bytes += (int64_t)n * (pair ? 16 : 8) * ncomp;
                a->mfma_flops += (int64_t)512 * cov_tiles_per_sample(T, bi == bj, pair, gram_mode) * n * ncomp;
            }
            hipLaunchKernelGGL(k_reduce_cov, dim3((unsigned)((width + 15) / 16) + (do_count ? 1u : 0u), ncomp), dim3(1024), 0, st, a->d_partials,
                               blocks * n_slices, NT, NG, a->RP, 64 * bi, 64 * bj, totals, a->int_width, (!pair && bi != bj) ? 1 : 0,
                               do_count ? a->d_pcounts : (const int64_t *)nullptr, blocks, a->d_counts + 2 * (int64_t)level);
            MLMC_HIP_CHECK(hipGetLastError());
        }
    return 0;
}

// out_s[lc][i][j] = 1/2 c_i c_j (G0_ij + G0_ji);  out_sp[lc][i][j] = 1/4 c_i^2 c_j^2 (G1_ij + G1_ji + 2 G2_ij)
__global__ void k_cov_finalize(const double *__restrict__ totals, const double *__restrict__ scale_c, int R, int RP,
                               int64_t int_width, double *__restrict__ out_s, double *__restrict__ out_sp,
                               const int64_t *__restrict__ counts, int n_levels, int64_t *__restrict__ out_n,
                               double *__restrict__ out_nd, int mean_only, const int64_t *__restrict__ counts2) {
    const int lc = blockIdx.y;
    if (lc == 0 && blockIdx.x == 0)
        for (int l = threadIdx.x; l < n_levels; l += blockDim.x) {
            // counts2: the chunks that never saw a covariance kernel (level 0 through the linearised second moments) were counted
            // by the moments kernel of the inner accumulator
            const int64_t kept = counts[2 * l] + (counts2 ? counts2[2 * l] : 0), removed = counts[2 * l + 1] + (counts2 ? counts2[2 * l + 1] : 0);
            out_n[l] = kept;
            out_n[n_levels + l] = removed;
            out_nd[l] = (double)kept;                        // exact below 2^53: lets one fp64 all-reduce carry the counts
            out_nd[n_levels + l] = (double)removed;
        }
    const double *G0 = totals + (int64_t)lc * int_width;
    const double *G1 = G0 + (int64_t)RP * RP;
    const double *G2 = G1 + (int64_t)RP * RP;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= R * R) return;
    const int i = idx / R, j = idx % R;
    const double ci = scale_c ? scale_c[i] : 1.0, cj = scale_c ? scale_c[j] : 1.0;   // nullptr: sums of true values
    const double cc = ci * cj;
    out_s[(int64_t)lc * R * R + idx] = 0.5 * cc * (G0[i * RP + j] + G0[j * RP + i]);
    out_sp[(int64_t)lc * R * R + idx] = mean_only ? __builtin_nan("")   // MLMC_MODE_MEAN_ONLY: not accumulated
                                                  : 0.25 * (cc * cc) * ((G1[i * RP + j] + G1[j * RP + i]) + 2.0 * G2[i * RP + j]);
}

// Mean of the covariance through the product linearisation: out_s[lc][i][j] = sum_k c_ijk (scale_k S_k[lc]), S = level sums of
// the 2 R - 1 (scaled) moments of the inner accumulator.  The table is k-major: consecutive threads read consecutive entries.
// A thread owns one (i, j) for up to LIN_LC (level, component) rows: every table entry is read once per LIN_LC rows.  parity_step 2
// (Legendre): c_ijk = 0 unless k = i + j (mod 2) (products), c2_ijk = 0 for odd k (squares: `squares` != 0) -- the other half of
// the table is not even read.
constexpr int LIN_LC = 8;
__global__ void k_cov_lin_mean(const double *__restrict__ prod, const double *__restrict__ lin_totals, const double *__restrict__ lin_scale,
                               int R, int K, int64_t lin_width, double *__restrict__ out_s, int n_lc, int parity_step, int squares) {
    extern __shared__ double m_s[];     // [LIN_LC][K] true-value sums of these (level, component) rows
    const int lc0 = blockIdx.y * LIN_LC;
    for (int q = threadIdx.x; q < LIN_LC * K; q += blockDim.x) {
        const int l = q / K, k = q % K;
        m_s[q] = lc0 + l < n_lc ? lin_scale[k] * lin_totals[(int64_t)(lc0 + l) * lin_width + k] : 0.0;
    }
    __syncthreads();
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int RR = R * R;
    if (idx >= RR) return;
    double acc[LIN_LC];
#pragma unroll
    for (int l = 0; l < LIN_LC; ++l) acc[l] = 0.0;
    const int k0 = (parity_step == 2 && !squares) ? ((idx / R + idx % R) & 1) : 0;
    for (int k = k0; k < K; k += parity_step) {
        const double c = prod[(int64_t)k * RR + idx];
#pragma unroll
        for (int l = 0; l < LIN_LC; ++l) acc[l] = __builtin_fma(c, m_s[l * K + k], acc[l]);
    }
#pragma unroll
    for (int l = 0; l < LIN_LC; ++l)
        if (lc0 + l < n_lc) out_s[(int64_t)(lc0 + l) * RR + idx] += acc[l];     // (+ the sums of the chunks that went the direct way)
}

int launch_cov_finalize(mlmc_accum *a) {
    if (a->lin && a->lin_used) {
        if (int rc = flush_moments(a->lin)) return rc;
    }
    const bool lin0 = a->lin0 && a->lin0_used;
    if (lin0) {
        if (int rc = flush_moments(a->lin0)) return rc;
    }
    const dim3 lin_grid((a->R * a->R + 255) / 256, (a->n_levels * a->n_comp + LIN_LC - 1) / LIN_LC);
    const int pstep = a->basis->p.kind == MLMC_LEGENDRE ? 2 : 1;
    const int n_lc = a->n_levels * a->n_comp;
    const bool vals = a->cov_from_values;              // TransformedMoments / more than 128 moments: accumulated from true values
    const int R = vals ? a->Rout : a->R;
    hipLaunchKernelGGL(k_cov_finalize, dim3((R * R + 255) / 256, n_lc), dim3(256), 0, rt().stream, a->d_totals,
                       vals ? (const double *)nullptr : a->basis->d_scale, R,
                       a->RP, a->int_width, a->d_out_s, a->d_out_sp, a->d_counts, a->n_levels, a->d_out_n, a->d_out_nd,
                       a->mean_only ? 1 : 0, lin0 ? a->lin0->d_counts : (const int64_t *)nullptr);
    MLMC_HIP_CHECK(hipGetLastError());
    if (lin0) {
        // level-0 chunks that went without a matrix pass: sum f_i f_j from the first 2 R - 1 extended sums, sum (f_i f_j)^2 from all
        // 4 R - 3 (at level 0 the finalize formula 1/4 (G1 + G1^T + 2 G2) is this very sum: both kinds of chunk add up)
        hipLaunchKernelGGL(k_cov_lin_mean, lin_grid, dim3(256), sizeof(double) * LIN_LC * a->lin_K, rt().stream,
                           a->d_lin_prod, a->lin0->d_totals, a->lin0_basis->d_scale, R, a->lin_K, a->lin0->int_width, a->d_out_s, n_lc, pstep, 0);
        hipLaunchKernelGGL(k_cov_lin_mean, lin_grid, dim3(256), sizeof(double) * LIN_LC * a->lin0_K, rt().stream,
                           a->d_lin0_prod, a->lin0->d_totals, a->lin0_basis->d_scale, R, a->lin0_K, a->lin0->int_width, a->d_out_sp, n_lc, pstep, 1);
        MLMC_HIP_CHECK(hipGetLastError());
    }
    if (a->lin && a->lin_used) {       // chunks without G0 (gram_mode 3): their share of the means comes from the extended moments
        hipLaunchKernelGGL(k_cov_lin_mean, lin_grid, dim3(256), sizeof(double) * LIN_LC * a->lin_K, rt().stream,
                           a->d_lin_prod, a->lin->d_totals, a->lin_basis->d_scale, R, a->lin_K, a->lin->int_width, a->d_out_s, n_lc, pstep, 0);
        MLMC_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

}  // namespace mlmc
