// Maximum-entropy density reconstruction on the device (gfx950).
//
// Reference: mlmc/tool/simple_distribution.py:50-94,127-152,198-327 (SimpleDistribution) and
// mlmc/tool/distribution.py:85-157,236-419 (Distribution): minimise
//     F(l) = sum_i mu_i l_i / s_i + int_a^b exp(-sum_i phi_i(x) l_i / s_i) dx  (+ penalties)
// with SciPy trust-ncg / trust-exact on a quadrature rebuilt from QUADPACK's adaptive sub-intervals.
// Here: a fixed composite Gauss-Legendre rule (n_intervals x degree points), the scaled basis matrix
// Phi[q][i] = phi_i(x_q) / s_i resident in HBM, and a damped Newton iteration whose pieces are kernels:
//   k_me_density : rho_q w_q = w_q exp(clip(-Phi_q . l, +-200)), F partial sums
//   k_me_grad    : g = mu/s - Phi^T (rho w)
//   k_me_hessian : H = Phi^T diag(rho w) Phi on the fp64 matrix cores (v_mfma_f64_16x16x4_f64)
//   k_me_penalty : end-point decay + stabilisation terms of distribution.py:354-359,375-380,404-412
//   k_me_solve   : Cholesky of H + tau I in LDS (one workgroup), p = -(H + tau I)^-1 g
// F is strictly convex, so Newton + Armijo backtracking converges to the same multipliers as the reference's
// trust-region iterations (which are not pinned by any reference test).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_basis.hpp"

namespace mlmc {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int ME_MAX_R = 128;

// rho w and the integral: one thread per quadrature point.  out_sum[0] += sum_q rho_q w_q (block partials, then
// a fixed-order final sum in k_me_finish).
__global__ void k_me_density(const double *__restrict__ Phi, const double *__restrict__ w, const double *__restrict__ lam,
                             int Q, int R1, double *__restrict__ rhow, double *__restrict__ block_sums) {
    __shared__ double l_s[ME_MAX_R];
    __shared__ double red[4];
    for (int i = threadIdx.x; i < R1; i += blockDim.x) l_s[i] = lam[i];
    __syncthreads();
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    double v = 0.0;
    if (q < Q) {
        double power = 0.0;
        const double *row = Phi + (int64_t)q * R1;
        for (int i = 0; i < R1; ++i) power = __builtin_fma(row[i], l_s[i], power);
        power = -power;
        power = fmin(fmax(power, -200.0), 200.0);        // simple_distribution.py:256
        v = w[q] * exp(power);
        rhow[q] = v;
    }
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// g_i = mu_i/s_i - sum_q Phi[q][i] rho_q w_q : one 256-thread block per moment i (fixed-order block reduction).
// Block 0 also sets scal[0] = F = mu~.l + sum rho w, scal[1] = int rho phi_0 s_0, scal[2] = sum rho w.
// grad == 0: functional only (line search) -- a single block.
__global__ __launch_bounds__(256) void k_me_grad(const double *__restrict__ Phi, const double *__restrict__ rhow,
                                                 const double *__restrict__ mu_s, const double *__restrict__ lam,
                                                 const double *__restrict__ sigma, int Q, int R1,
                                                 const double *__restrict__ block_sums, int n_blocks, int grad,
                                                 double *__restrict__ g, double *__restrict__ scal) {
    __shared__ double red[4];
    const int i = blockIdx.x;
    if (grad) {
        double acc = 0.0;
        for (int q = threadIdx.x; q < Q; q += 256) acc = __builtin_fma(Phi[(int64_t)q * R1 + i], rhow[q], acc);
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double tot = ((red[0] + red[1]) + red[2]) + red[3];
            g[i] = mu_s[i] - tot;
            if (i == 0) scal[1] = tot * sigma[0];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 64) {
        double integral = 0.0;
        for (int b = 0; b < n_blocks; ++b) integral += block_sums[b];
        double lin = 0.0;
        for (int k = 0; k < R1; ++k) lin = __builtin_fma(mu_s[k], lam[k], lin);
        scal[0] = lin + integral;
        scal[2] = integral;
    }
}

// H tile (ti, tj >= ti): 4 waves split the quadrature points, partial tiles summed through LDS.
__global__ __launch_bounds__(256) void k_me_hessian(const double *__restrict__ Phi, const double *__restrict__ rhow, int Q,
                                                    int R1, int T, double *__restrict__ H) {
    // decode upper-triangular tile index
    int t = blockIdx.x, ti = 0;
    while (t >= T - ti) { t -= T - ti; ++ti; }
    const int tj = ti + t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ia = 16 * ti + (lane & 15), ib = 16 * tj + (lane & 15);
    const bool va = ia < R1, vb = ib < R1;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
    const int steps = (Q + 3) / 4;
    for (int ks = wave; ks < steps; ks += 4) {
        const int q = 4 * ks + (lane >> 4);
        double a = 0.0, b = 0.0;
        if (q < Q) {
            const double rw = rhow[q];
            if (va) a = Phi[(int64_t)q * R1 + ia] * rw;
            if (vb) b = Phi[(int64_t)q * R1 + ib];
        }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    __shared__ double red[4][256];
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = r * 64 + lane;
            const double v = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
            const int row = 16 * ti + (lane >> 4) + 4 * r, col = 16 * tj + (lane & 15);
            if (row < R1 && col < R1) {
                H[(int64_t)row * R1 + col] = v;
                H[(int64_t)col * R1 + row] = v;
            }
        }
    }
}

// penalties of the older solver (distribution.py): end-point decay (coef * |fun| * max(e.l, 0)^2) and stabilisation
// (0.5 stab |l_prev - l|^2); formulas for gradient / Hessian exactly as the reference writes them (:375-380, :404-412).
__global__ void k_me_penalty(const double *__restrict__ end_diff, const double *__restrict__ lam,
                             const double *__restrict__ prev, int n_prev, double stab, double coef,
                             const double *__restrict__ mu_s, const double *__restrict__ sigma, int R1,
                             double *__restrict__ g, double *__restrict__ H, double *__restrict__ scal) {
    __shared__ double ed[2];
    __shared__ double lin_s;
    if (threadIdx.x < 2) {
        double d = 0.0;
        for (int k = 0; k < R1; ++k) d = __builtin_fma(end_diff[threadIdx.x * R1 + k], lam[k], d);
        ed[threadIdx.x] = d;
    }
    if (threadIdx.x == 2) {
        double lin = 0.0;
        for (int k = 0; k < R1; ++k) lin = __builtin_fma(mu_s[k], lam[k], lin);
        lin_s = lin;
    }
    __syncthreads();
    const double p0 = fmax(ed[0], 0.0), p1 = fmax(ed[1], 0.0);
    const double fun_g = lin_s + scal[1];                                  // distribution.py:376
    const double fun_h = lin_s + H[0] * sigma[0] * sigma[0];              // distribution.py:401-402
    __syncthreads();
    for (int i = threadIdx.x; i < R1; i += blockDim.x) {
        double gi = g[i] + fabs(fun_g) * coef * 2.0 * (p0 * end_diff[i] + p1 * end_diff[R1 + i]);
        if (i < n_prev) gi += stab * (lam[i] - prev[i]);
        g[i] = gi;
    }
    for (int idx = threadIdx.x; idx < R1 * R1; idx += blockDim.x) {
        const int i = idx / R1, j = idx % R1;
        double h = H[idx];
        if (ed[0] > 0) h += fabs(fun_h) * coef * 2.0 * end_diff[i] * end_diff[j];
        if (ed[1] > 0) h += fabs(fun_h) * coef * 2.0 * end_diff[R1 + i] * end_diff[R1 + j];
        if (i == j) h += stab;
        H[idx] = h;
    }
    if (threadIdx.x == 0) {
        double f = scal[0];
        f = f + fabs(f) * coef * (p0 * p0 + p1 * p1);                      // distribution.py:355-357
        double sq = 0.0;
        for (int k = 0; k < n_prev; ++k) sq += (prev[k] - lam[k]) * (prev[k] - lam[k]);
        scal[0] = f + 0.5 * stab * sq;                                     // distribution.py:358-359
    }
}

// Cholesky of H + tau I in LDS, p = -(H + tau I)^-1 g.  scal[3] = ||g||_2, scal[4] = g.p, scal[5] = 1 if not SPD.
__global__ __launch_bounds__(256) void k_me_solve(const double *__restrict__ H, const double *__restrict__ g, int R1, double tau,
                                                  double *__restrict__ p, double *__restrict__ scal) {
    extern __shared__ double sm[];
    const int ld = R1 + 1;
    double *A = sm;                 // [R1][ld]
    double *y = sm + (size_t)R1 * ld;
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    for (int idx = threadIdx.x; idx < R1 * R1; idx += blockDim.x) {
        const int i = idx / R1, j = idx % R1;
        A[i * ld + j] = H[idx] + (i == j ? tau : 0.0);
    }
    __syncthreads();
    for (int k = 0; k < R1; ++k) {
        if (threadIdx.x == 0) {
            const double d = A[k * ld + k];
            if (!(d > 0.0)) bad = 1;
            A[k * ld + k] = sqrt(d > 0.0 ? d : 1.0);
        }
        __syncthreads();
        const double dk = A[k * ld + k];
        for (int i = k + 1 + threadIdx.x; i < R1; i += blockDim.x) A[i * ld + k] /= dk;
        __syncthreads();
        const int m = R1 - k - 1;      // trailing update of the lower triangle
        for (int idx = threadIdx.x; idx < m * m; idx += blockDim.x) {
            const int i = k + 1 + idx / m, j = k + 1 + idx % m;
            if (j <= i) A[i * ld + j] -= A[i * ld + k] * A[j * ld + k];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // forward: L y = -g ; backward: L^T p = y
        for (int i = 0; i < R1; ++i) {
            double v = -g[i];
            for (int j = 0; j < i; ++j) v -= A[i * ld + j] * y[j];
            y[i] = v / A[i * ld + i];
        }
        for (int i = R1 - 1; i >= 0; --i) {
            double v = y[i];
            for (int j = i + 1; j < R1; ++j) v -= A[j * ld + i] * y[j];
            y[i] = v / A[i * ld + i];
        }
        double gn = 0.0, gp = 0.0;
        for (int i = 0; i < R1; ++i) { p[i] = y[i]; gn += g[i] * g[i]; gp += g[i] * y[i]; }
        scal[3] = sqrt(gn);
        scal[4] = gp;
        scal[5] = bad ? 1.0 : 0.0;
    }
}

__global__ void k_me_axpy(const double *__restrict__ lam, const double *__restrict__ p, double alpha, int R1,
                          double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R1) out[i] = __builtin_fma(alpha, p[i], lam[i]);
}

__global__ void k_me_scale_cols(double *__restrict__ Phi, const double *__restrict__ sigma, int Q, int R1) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (int64_t)Q * R1) Phi[idx] /= sigma[idx % R1];
}


// ------------------------------------------------------------------------------------------
// The damped Newton iteration as ONE cooperative launch (SimpleDistribution's functional; with the end-point decay and
// stabilisation penalties of the older Distribution solver when the options carry them).
// The kernel-per-step formulation above pays launch latency and one host round trip per Newton step and per
// line-search trial (~160 us per step); the problem itself is tiny.  Here NB workgroups each own a slice of QS
// quadrature points: the slice of the scaled basis sits in LDS for the whole solve, every workgroup computes its
// partial integral / gradient / Hessian from LDS, the partials are summed in a fixed order after a grid barrier, and
// EVERY workgroup then runs the same Cholesky solve and takes the same decisions (regularisation, Armijo backtracking
// -- four step lengths per barrier -- and convergence), so nothing has to be broadcast.  Same algorithm and constants
// as the host loop of mlmc_maxent_solve, which stays as the fallback (and as the validation path: MLMC_MAXENT_STEPWISE=1).
// With penalties the reference's gradient is not the gradient of its functional (distribution.py:375-380 scales the decay
// term by |fun|), so -- as in the host loop -- a step is accepted when it lowers the gradient NORM: the trial point's
// gradient needs the full partial sums anyway, and an accepted trial's sums are the next iterate's, whatever the step length.
// The grid barrier spins on an agent-scope generation counter with a bounded number of polls: if the workgroups are
// ever not co-resident the kernel gives up (result[5] = 1) instead of hanging, and the host falls back.
// ------------------------------------------------------------------------------------------
constexpr int COOP_THREADS = 256;
constexpr int COOP_MAX_BLOCKS = 128;

struct CoopArgs {
    const double *Phi;       // [Q][R1] scaled basis (row-major)
    const double *w;         // [Q]
    const double *mu_s;      // [R1]
    const double *sigma;     // [R1]
    double *lam;             // [R1] in: start, out: multipliers
    double *g_out;           // [R1]
    double *H_out;           // [R1][R1]
    double *result;          // [8]: nit, success, F, gnorm, moment0, aborted
    double *part;            // [NB][E] partials, E = npairs + R1 + 1
    double *tot;             // [E]
    double *part_ls;         // [NB][4]
    unsigned *bar;           // [2]: arrival counter (monotonic), generation
    int Q, R1, NB, QS, max_it;
    double tol;
    // penalised functional (distribution.py:354-359, :375-380, :404-412); use_pen == 0: none of these is read
    const double *end_diff;  // [2][R1] end-point derivative estimates / sigma
    const double *prev;      // [n_prev] multipliers of the previous stage
    int n_prev, use_pen;
    double stab, coef;
};

__device__ __forceinline__ double readlane_f64(double v, int lane) {   // lane must be wave-uniform
    const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(v), lane);
    const unsigned hi = __builtin_amdgcn_readlane((int)__double2hiint(v), lane);
    return __hiloint2double((int)hi, (int)lo);
}

__device__ __forceinline__ bool coop_barrier(unsigned *bar, unsigned nb, unsigned &gen, int *abort_s) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();                                     // release this workgroup's global writes (agent scope)
        ++gen;
        const unsigned arrived = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == nb * gen) {
            __hip_atomic_store(&bar[1], gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int polls = 0;
            while (__hip_atomic_load(&bar[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gen) {
                __builtin_amdgcn_s_sleep(4);
                if (++polls > (1 << 19)) { *abort_s = 1; break; }   // ~0.1 s: the grid is not co-resident
            }
        }
        __threadfence();                                     // acquire the other workgroups' writes
    }
    __syncthreads();
    return *abort_s == 0;
}

__global__ __launch_bounds__(COOP_THREADS) void k_me_coop(CoopArgs A) {
    extern __shared__ double sm[];
    const int R1 = A.R1, Q = A.Q, QS = A.QS, NB = A.NB;
    const int R1p = R1 | 1, ld = R1 + 1;
    const int npairs = R1 * (R1 + 1) / 2, E = npairs + R1 + 1;
    double *ph = sm;                                   // [QS][R1p]  slice of Phi (whole solve)
    double *lam = ph + (size_t)QS * R1p;               // [R1]
    double *trial = lam + R1;                          // [R1]  (not used: step lengths are applied on the fly)
    double *g = trial + R1;                            // [R1]
    double *pdir = g + R1;                             // [R1]
    double *y = pdir + R1;                             // [R1]
    double *rw = y + R1;                               // [4][QS] rho w of the slice (4 step lengths in the line search)
    double *wsl = rw + 4 * QS;                         // [QS] weights of the slice
    double *red = wsl + QS;                            // [16]
    double *mus = red + 16;                            // [R1] mu_i / sigma_i
    double *lsum = mus + R1;                           // [4][COOP_MAX_BLOCKS] line-search partials of all workgroups
    double *endd = lsum + 4 * COOP_MAX_BLOCKS;         // [2][R1] end-point derivative rows (penalised functional)
    double *prv = endd + 2 * R1;                       // [R1] previous stage's multipliers, zero beyond n_prev
    double *Lm = prv + R1;                             // [R1][ld] Cholesky matrix; aliased by psi [QS][R1p] in phase A
    double *psi = Lm;
    __shared__ int abort_s, bad_s;
    const int tid = threadIdx.x, b = blockIdx.x;
    const int q0 = b * QS, nq = max(0, min(QS, Q - q0));
    unsigned gen = 0;
    if (tid == 0) abort_s = 0;
    for (int idx = tid; idx < QS * R1; idx += COOP_THREADS) {
        const int q = idx / R1, i = idx % R1;
        ph[q * R1p + i] = q < nq ? A.Phi[(int64_t)(q0 + q) * R1 + i] : 0.0;
    }
    for (int q = tid; q < QS; q += COOP_THREADS) wsl[q] = q < nq ? A.w[q0 + q] : 0.0;
    for (int i = tid; i < R1; i += COOP_THREADS) { lam[i] = A.lam[i]; pdir[i] = 0.0; mus[i] = A.mu_s[i]; }
    const bool pen = A.use_pen != 0;
    const int n_prev = A.n_prev;
    if (pen) {
        for (int i = tid; i < 2 * R1; i += COOP_THREADS) endd[i] = A.end_diff[i];
        for (int i = tid; i < R1; i += COOP_THREADS) prv[i] = i < n_prev ? A.prev[i] : 0.0;
    }
    __syncthreads();
    // penalty scalars at the point `pt`: red[8] = e_left . pt, red[9] = e_right . pt, red[11] = sum_{k < n_prev} (prev_k - pt_k)^2
    auto pen_scalars = [&](const double *pt) {
        if (tid < 64) {
            double e0 = 0.0, e1 = 0.0, sq = 0.0;
            for (int k = tid; k < R1; k += 64) {
                e0 = __builtin_fma(endd[k], pt[k], e0);
                e1 = __builtin_fma(endd[R1 + k], pt[k], e1);
                if (k < n_prev) sq = __builtin_fma(prv[k] - pt[k], prv[k] - pt[k], sq);
            }
            e0 = wave_sum(e0);
            e1 = wave_sum(e1);
            sq = wave_sum(sq);
            if (tid == 0) { red[8] = e0; red[9] = e1; red[11] = sq; }
        }
        __syncthreads();
    };

    // rho w of the slice for up to 4 step lengths alpha_k = alpha0 / 2^k along pdir; 8 lanes share one (point, k)
    auto slice_density = [&](double alpha0, int n_alpha) {
        for (int item = tid >> 3; item < QS * n_alpha; item += COOP_THREADS / 8) {
            const int k = item / QS, q = item % QS, sub = tid & 7;
            const double alpha = alpha0 / (double)(1 << k);
            double power = 0.0;
            for (int i = sub; i < R1; i += 8) power = __builtin_fma(ph[q * R1p + i], __builtin_fma(alpha, pdir[i], lam[i]), power);
            power += __shfl_xor(power, 1, 64);
            power += __shfl_xor(power, 2, 64);
            power += __shfl_xor(power, 4, 64);
            if (sub == 0) rw[k * QS + q] = wsl[q] * exp(fmin(fmax(-power, -200.0), 200.0));
        }
        __syncthreads();
    };

    int nit = 0, success = 0, ok = 1;
    double tau = 0.0, F = 0.0, gnorm = 0.0, moment0 = 0.0, gp = 0.0;
    // `spec`: the partials of this round are taken at lam + pdir, the full Newton step -- if it passes the Armijo test
    // (the usual case near the solution) they ARE the next iterate's integral / gradient / Hessian, so an accepted
    // step costs no line-search round trip at all; otherwise the step lengths 1/2, 1/4, ... are tried four per barrier.
    bool spec = false;
    double alpha_try = 1.0, pen_ed0 = 0.0, pen_ed1 = 0.0, pen_fun_h = 0.0;     // penalised: step length of the trial point
    int ls_count = 0;
    for (int it = 0; it <= A.max_it && ok;) {
#ifdef MLMC_PROF_COOP
        unsigned long long st_[8];
        st_[0] = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- phase A: partial integral, gradient and Hessian of this slice at lam (or at lam + pdir) ----
        slice_density(spec ? alpha_try : 0.0, 1);
        for (int idx = tid; idx < QS * R1; idx += COOP_THREADS) {
            const int q = idx / R1, i = idx % R1;
            psi[q * R1p + i] = ph[q * R1p + i] * rw[q];
        }
        __syncthreads();
        double *mine = A.part + (size_t)b * E;
        for (int p = tid; p < npairs; p += COOP_THREADS) {
            // upper triangle, row-major: p = i (2 R1 - i + 1) / 2 + (j - i)
            int i = (int)floor(((double)(2 * R1 + 1) - sqrt((double)(2 * R1 + 1) * (2 * R1 + 1) - 8.0 * p)) * 0.5);
            while (i * (2 * R1 - i + 1) / 2 > p) --i;
            while ((i + 1) * (2 * R1 - i) / 2 <= p) ++i;
            const int j = i + (p - i * (2 * R1 - i + 1) / 2);
            double acc = 0.0;
            for (int q = 0; q < QS; ++q) acc = __builtin_fma(psi[q * R1p + i], ph[q * R1p + j], acc);
            mine[p] = acc;
        }
        for (int i = tid; i < R1; i += COOP_THREADS) {
            double acc = 0.0;
            for (int q = 0; q < QS; ++q) acc += psi[q * R1p + i];
            mine[npairs + i] = acc;
        }
        if (tid == 0) {
            double acc = 0.0;
            for (int q = 0; q < QS; ++q) acc += rw[q];
            mine[npairs + R1] = acc;
        }
#ifdef MLMC_PROF_COOP
        st_[1] = __builtin_amdgcn_s_memrealtime();
#endif
        if (!coop_barrier(A.bar, NB, gen, &abort_s)) { ok = 0; break; }
#ifdef MLMC_PROF_COOP
        st_[2] = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- phase B: fixed-order sums of the partials, spread over the grid ----
        // (8 lanes per entry: the partials of the other workgroups come from remote L2 / HBM, keep the loads parallel)
        for (int e = b * (COOP_THREADS / 8) + (tid >> 3); e < E; e += NB * (COOP_THREADS / 8)) {
            double acc = 0.0;
            for (int k = tid & 7; k < NB; k += 8) acc += A.part[(size_t)k * E + e];
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if ((tid & 7) == 0) A.tot[e] = acc;
        }
#ifdef MLMC_PROF_COOP
        st_[3] = __builtin_amdgcn_s_memrealtime();
#endif
        if (!coop_barrier(A.bar, NB, gen, &abort_s)) { ok = 0; break; }
#ifdef MLMC_PROF_COOP
        st_[4] = __builtin_amdgcn_s_memrealtime();
#endif
        if (spec && pen) {
            // the totals belong to trial = lam + alpha_try pdir: its gradient (with penalties) decides (same in every workgroup)
            for (int i = tid; i < R1; i += COOP_THREADS) trial[i] = __builtin_fma(alpha_try, pdir[i], lam[i]);
            __syncthreads();
            pen_scalars(trial);
            double lin_t = 0.0;
            for (int k = 0; k < R1; ++k) lin_t = __builtin_fma(mus[k], trial[k], lin_t);
            const double p0 = fmax(red[8], 0.0), p1 = fmax(red[9], 0.0);
            const double fun_g = lin_t + A.tot[npairs] * A.sigma[0];                          // distribution.py:376
            if (tid < 64) {
                double a = 0.0;
                for (int i = tid; i < R1; i += 64) {
                    double gi = (mus[i] - A.tot[npairs + i]) + fabs(fun_g) * A.coef * 2.0 * (p0 * endd[i] + p1 * endd[R1 + i]);
                    if (i < n_prev) gi += A.stab * (trial[i] - prv[i]);
                    a = __builtin_fma(gi, gi, a);
                }
                a = wave_sum(a);
                if (tid == 0) red[13] = sqrt(a);
            }
            __syncthreads();
            const double gt = red[13];
            if (!(gt == gt && gt < gnorm)) {
                alpha_try *= 0.5;
                if (++ls_count < 40) continue;                               // the next round evaluates the shorter step
                spec = false;
                tau = (tau == 0.0) ? 1e-8 * (1.0 + fabs(F)) : tau * 100.0;
                if (tau > 1e20) break;
                continue;                                                   // re-evaluate at lam with the larger shift
            }
            for (int i = tid; i < R1; i += COOP_THREADS) lam[i] = trial[i];
            __syncthreads();
            tau = (alpha_try == 1.0) ? tau * 0.1 : tau;
            if (tau < 1e-14) tau = 0.0;
            ++nit;
            spec = false;                                                   // the totals ARE those of the new lam: on to phase C
        } else if (spec) {
            // the totals belong to lam + pdir: Armijo test of the full step (same in every workgroup)
            spec = false;
            const double lin0 = F - red[4];                                 // mu~ . lam  (red[4]: integral at lam)
            const double Ft = (lin0 + red[3]) + A.tot[npairs + R1];
            bool accepted = Ft == Ft && Ft <= F + 1e-4 * gp;
            double alpha = 1.0;
            if (!accepted) {
                // ---- backtracking: step lengths alpha / 2^k, four per barrier ----
                alpha = 0.5;
                for (int batch = 0; batch < 10 && !accepted && ok; ++batch) {
                    slice_density(alpha, 4);
                    if (tid < 4) {
                        double acc = 0.0;
                        for (int q = 0; q < QS; ++q) acc += rw[tid * QS + q];
                        A.part_ls[(size_t)(batch & 1) * NB * 4 + (size_t)b * 4 + tid] = acc;
                    }
                    if (!coop_barrier(A.bar, NB, gen, &abort_s)) { ok = 0; break; }
                    for (int idx = tid; idx < NB * 4; idx += COOP_THREADS)      // one parallel sweep over the remote partials
                        lsum[(idx & 3) * COOP_MAX_BLOCKS + (idx >> 2)] = A.part_ls[(size_t)(batch & 1) * NB * 4 + idx];
                    __syncthreads();
                    for (int k = 0; k < 4; ++k) {
                        const double ak = alpha / (double)(1 << k);
                        double integral = 0.0;
                        for (int kb = 0; kb < NB; ++kb) integral += lsum[k * COOP_MAX_BLOCKS + kb];
                        const double Fk = __builtin_fma(ak, red[3], lin0) + integral;
                        if (Fk == Fk && Fk <= F + 1e-4 * ak * gp) { accepted = true; alpha = ak; break; }
                    }
                    __syncthreads();
                    if (!accepted) alpha /= 16.0;
                }
                if (!ok) break;
            }
            if (!accepted) {
                tau = (tau == 0.0) ? 1e-8 * (1.0 + fabs(F)) : tau * 100.0;
                if (tau > 1e20) break;
                continue;                                                   // re-evaluate at lam with the larger shift
            }
            __syncthreads();
            for (int i = tid; i < R1; i += COOP_THREADS) lam[i] = __builtin_fma(alpha, pdir[i], lam[i]);
            __syncthreads();
            tau = (alpha == 1.0) ? tau * 0.1 : tau;
            if (tau < 1e-14) tau = 0.0;
            ++nit;
            if (alpha != 1.0) continue;                                     // the totals are not those of the new lam
        }
        // ---- phase C (identical in every workgroup): gradient, L D L^T of H + tau I, Newton direction ----
        for (int i = tid; i < R1; i += COOP_THREADS) g[i] = mus[i] - A.tot[npairs + i];
        for (int idx = tid; idx < R1 * R1; idx += COOP_THREADS) {
            int i = idx / R1, j = idx % R1;
            if (i > j) { const int t = i; i = j; j = t; }
            Lm[(idx / R1) * ld + (idx % R1)] = A.tot[i * (2 * R1 - i + 1) / 2 + (j - i)] + ((idx / R1) == (idx % R1) ? tau : 0.0);
        }
        if (tid == 0) bad_s = 0;
        __syncthreads();
#ifdef MLMC_PROF_COOP
        st_[6] = __builtin_amdgcn_s_memrealtime();
#endif
        {
            double lin = 0.0;
            for (int k = 0; k < R1; ++k) lin = __builtin_fma(mus[k], lam[k], lin);
            F = lin + A.tot[npairs + R1];
            moment0 = A.tot[npairs] * A.sigma[0];
            if (pen) {      // k_me_penalty's formulas at lam, applied by every workgroup to its own copy
                pen_scalars(lam);
                pen_ed0 = red[8];
                pen_ed1 = red[9];
                const double p0 = fmax(pen_ed0, 0.0), p1 = fmax(pen_ed1, 0.0);
                const double fun_g = lin + moment0;                                             // distribution.py:376
                pen_fun_h = lin + A.tot[0] * A.sigma[0] * A.sigma[0];                           // distribution.py:401-402
                for (int i = tid; i < R1; i += COOP_THREADS) {
                    double gi = g[i] + fabs(fun_g) * A.coef * 2.0 * (p0 * endd[i] + p1 * endd[R1 + i]);
                    if (i < n_prev) gi += A.stab * (lam[i] - prv[i]);
                    g[i] = gi;
                }
                for (int idx = tid; idx < R1 * R1; idx += COOP_THREADS) {
                    const int i = idx / R1, j = idx % R1;
                    double h = 0.0;
                    if (pen_ed0 > 0) h += fabs(pen_fun_h) * A.coef * 2.0 * endd[i] * endd[j];
                    if (pen_ed1 > 0) h += fabs(pen_fun_h) * A.coef * 2.0 * endd[R1 + i] * endd[R1 + j];
                    if (i == j) h += A.stab;
                    Lm[i * ld + j] += h;
                }
                F = F + fabs(F) * A.coef * (p0 * p0 + p1 * p1);                                 // distribution.py:355-357
                F = F + 0.5 * A.stab * red[11];                                                 // distribution.py:358-359
                __syncthreads();
            }
        }
        // L D L^T in place, ONE workgroup barrier per column: step k only reads column k (never scales it), so the
        // trailing update A_ij -= A_ik A_jk / d_k needs no second barrier; afterwards Lm[i][k] = L_ik d_k (i > k) and
        // Lm[k][k] = d_k.  Positive definite <=> every pivot d_k > 0.
        for (int k = 0; k < R1; ++k) {
            const double d = Lm[k * ld + k];
            if (!(d > 0.0)) {                       // same value in every thread: uniform exit
                if (tid == 0) bad_s = 1;
                break;
            }
            const double inv_d = 1.0 / d;
            if (tid == 0) y[k] = inv_d;
            for (int i = k + 1 + (tid >> 4); i < R1; i += 16) {          // 16 x 16 thread tile over the trailing block
                const double lik = Lm[i * ld + k] * inv_d;
                for (int j = k + 1 + (tid & 15); j <= i; j += 16) Lm[i * ld + j] = __builtin_fma(-lik, Lm[j * ld + k], Lm[i * ld + j]);
            }
            __syncthreads();
        }
        __syncthreads();
#ifdef MLMC_PROF_COOP
        st_[7] = __builtin_amdgcn_s_memrealtime();
#endif
        if (tid < 64) {
            // triangular solves by one wave, column oriented: lane j carries entries j and j + 64 of the running
            // right-hand side, every step broadcasts one finished entry (no workgroup barriers, no divisions)
            const int j0 = tid, j1 = tid + 64;
            const bool spd = bad_s == 0;
            double v0 = j0 < R1 ? -g[j0] : 0.0, v1 = j1 < R1 ? -g[j1] : 0.0;
            const double id0 = j0 < R1 ? y[j0] : 0.0, id1 = j1 < R1 ? y[j1] : 0.0;       // 1 / d_j
            // the matrix entries and 1 / d_i of step i + 1 are requested while step i's broadcast chain runs: only the
            // readlane -> fma dependency stays on the critical path, not an LDS round trip per step
            const int jc0 = j0 < R1 ? j0 : 0, jc1 = j1 < R1 ? j1 : 0;
            const bool two = R1 > 64;                              // lanes carry a second entry only beyond 64 moments
            if (spd) {
                double a0 = Lm[jc0 * ld + 0], a1 = two ? Lm[jc1 * ld + 0] : 0.0, yi = y[0];
                for (int i = 0; i < R1; ++i) {                     // L u = -g, then z = D^-1 u (z_i replaces u_i)
                    const int in = i + 1 < R1 ? i + 1 : i;
                    const double n0 = Lm[jc0 * ld + in], n1 = two ? Lm[jc1 * ld + in] : 0.0, yn = y[in];
                    const double zi = readlane_f64(i < 64 ? v0 : v1, i & 63) * yi;
                    if (j0 == i) v0 = zi;
                    if (j1 == i) v1 = zi;
                    if (j0 > i && j0 < R1) v0 = __builtin_fma(-a0, zi, v0);
                    if (j1 > i && j1 < R1) v1 = __builtin_fma(-a1, zi, v1);
                    a0 = n0; a1 = n1; yi = yn;
                }
                double b0 = Lm[(R1 - 1) * ld + jc0] * id0, b1 = two ? Lm[(R1 - 1) * ld + jc1] * id1 : 0.0;
                for (int i = R1 - 1; i >= 0; --i) {                // L^T p = z: p_j = z_j - sum_{i > j} L_ij p_i, L_ij = Lm[i][j] / d_j
                    const int in = i > 0 ? i - 1 : 0;
                    const double n0 = Lm[in * ld + jc0] * id0, n1 = two ? Lm[in * ld + jc1] * id1 : 0.0;
                    const double pi = readlane_f64(i < 64 ? v0 : v1, i & 63);
                    if (j0 < i) v0 = __builtin_fma(-b0, pi, v0);
                    if (j1 < i) v1 = __builtin_fma(-b1, pi, v1);
                    b0 = n0; b1 = n1;
                }
            }
            if (j0 < R1) pdir[j0] = v0;
            if (j1 < R1) pdir[j1] = v1;
            double gn = (j0 < R1 ? g[j0] * g[j0] : 0.0) + (j1 < R1 ? g[j1] * g[j1] : 0.0);
            double gpv = (j0 < R1 ? g[j0] * v0 : 0.0) + (j1 < R1 ? g[j1] * v1 : 0.0);
            double lpv = (j0 < R1 ? mus[j0] * v0 : 0.0) + (j1 < R1 ? mus[j1] * v1 : 0.0);
            gn = wave_sum(gn);
            gpv = wave_sum(gpv);
            lpv = wave_sum(lpv);
            if (tid == 0) {
                red[0] = sqrt(gn);
                red[1] = gpv;
                red[2] = bad_s ? 1.0 : 0.0;
                red[3] = lpv;                                    // mu~ . p: the linear part of F along the direction
                red[4] = A.tot[npairs + R1];                     // integral at lam
            }
        }
        __syncthreads();
#ifdef MLMC_PROF_COOP
        st_[5] = __builtin_amdgcn_s_memrealtime();
        if (b == 0 && tid == 0 && it == 3)
            for (int k = 0; k < 8; ++k) A.result[8 + k] = (double)(st_[k] - st_[0]) / 100.0;   // us
#endif
        gnorm = red[0];
        gp = red[1];
        const bool not_spd = red[2] != 0.0;
        if (!(gnorm == gnorm) || !(F == F)) break;
        if (gnorm < A.tol) { success = 1; break; }
        if (it == A.max_it) break;
        ++it;
        if (not_spd || !(gp < 0.0)) {
            tau = (tau == 0.0) ? 1e-10 * (1.0 + fabs(F)) : tau * 100.0;
            if (tau > 1e20) break;
            continue;
        }
        spec = true;                                             // next round: partials at lam + pdir
        alpha_try = 1.0;
        ls_count = 0;
    }
    __syncthreads();
    if (b == 0) {
        for (int i = tid; i < R1; i += COOP_THREADS) { A.lam[i] = lam[i]; A.g_out[i] = g[i]; }
        if (ok)
            for (int idx = tid; idx < R1 * R1; idx += COOP_THREADS) {
                int i = idx / R1, j = idx % R1;
                if (i > j) { const int t = i; i = j; j = t; }
                double h = A.tot[i * (2 * R1 - i + 1) / 2 + (j - i)];
                if (pen) {
                    const int r = idx / R1, c = idx % R1;
                    if (pen_ed0 > 0) h += fabs(pen_fun_h) * A.coef * 2.0 * endd[r] * endd[c];
                    if (pen_ed1 > 0) h += fabs(pen_fun_h) * A.coef * 2.0 * endd[R1 + r] * endd[R1 + c];
                    if (r == c) h += A.stab;
                }
                A.H_out[idx] = h;
            }
        if (tid == 0) {
            A.result[0] = (double)nit;
            A.result[1] = (double)success;
            A.result[2] = F;
            A.result[3] = gnorm;
            A.result[4] = moment0;
            A.result[5] = ok ? 0.0 : 1.0;
        }
    }
}

// density(x) = exp(clip(-sum_r c_r Q_r(x), +-200)), c = effective coefficients in the underlying (scaled) family
template <int KIND>
__global__ void k_density(BasisParams bp, const double *__restrict__ c, int R,
                          const double *__restrict__ x, int64_t n, double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool keep;
    const double t = transform_value(bp, x[i], keep);
    TermGen<KIND> g;
    g.init(keep ? t : 0.0, 1.0, bp);
    double power = 0.0;
    for (int r = 0; r < R; ++r) power = __builtin_fma(g.next(r), c[r], power);
    power = fmin(fmax(-power, -200.0), 200.0);
    out[i] = keep ? exp(power) : __builtin_nan("");
}

// integral of the density over [lo_i, hi_i] with a `deg`-point Gauss-Legendre rule (nodes/weights on [-1, 1])
template <int KIND>
__global__ void k_density_integrate(BasisParams bp, const double *__restrict__ c, int R,
                                    const double *__restrict__ lo, const double *__restrict__ hi, int64_t n,
                                    const double *__restrict__ nodes, const double *__restrict__ wts, int deg,
                                    double *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = lo[i], b = hi[i];
    const double half = 0.5 * (b - a), mid = 0.5 * (b + a);
    double acc = 0.0;
    for (int k = 0; k < deg; ++k) {
        bool keep;
        const double t = transform_value(bp, __builtin_fma(half, nodes[k], mid), keep);
        TermGen<KIND> g;
        g.init(keep ? t : 0.0, 1.0, bp);
        double power = 0.0;
        for (int r = 0; r < R; ++r) power = __builtin_fma(g.next(r), c[r], power);
        power = fmin(fmax(-power, -200.0), 200.0);
        acc = __builtin_fma(wts[k], keep ? exp(power) : __builtin_nan(""), acc);
    }
    out[i] = acc * half;
}

// Gauss-Legendre nodes / weights on [-1, 1] (Newton on P_n, host, long double)
static void gauss_legendre(int n, std::vector<double> &x, std::vector<double> &w) {
    x.assign(n, 0.0);
    w.assign(n, 0.0);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int i = 0; i < (n + 1) / 2; ++i) {
        long double z = cosl(pi * (i + 0.75L) / (n + 0.5L));
        long double pp = 1.0L;
        for (int it = 0; it < 100; ++it) {
            long double p1 = 1.0L, p2 = 0.0L;
            for (int j = 1; j <= n; ++j) {
                long double p3 = p2;
                p2 = p1;
                p1 = ((2.0L * j - 1.0L) * z * p2 - (j - 1.0L) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0L);
            long double dz = p1 / pp;
            z -= dz;
            if (fabsl(dz) < 1e-19L) break;
        }
        x[i] = (double)(-z);
        x[n - 1 - i] = (double)z;
        w[i] = w[n - 1 - i] = (double)(2.0L / ((1.0L - z * z) * pp * pp));
    }
}

// effective coefficients in the underlying scaled family: c_r = scale_c[r] * sum_j T[j][r] lambda_j / sigma_j
static std::vector<double> effective_coeffs(const mlmc_basis *b, const double *lambda, const double *sigma, int R1) {
    const int R = b->p.size;
    std::vector<double> c(R, 0.0);
    if (b->out_size > 0) {
        for (int j = 0; j < R1; ++j) {
            const double lj = lambda[j] / sigma[j];
            for (int r = 0; r < R; ++r) c[r] += b->matrix[(size_t)j * R + r] * lj;
        }
    } else {
        for (int r = 0; r < R1; ++r) c[r] = lambda[r] / sigma[r];
    }
    for (int r = 0; r < R; ++r) c[r] *= b->scale_c[r];
    return c;
}

// One device allocation per call, carved into 256-byte aligned sub-buffers.
struct DevPool {
    char *base = nullptr;
    size_t used = 0, cap = 0;
    bool owner = true;
    ~DevPool() { if (base && owner) (void)hipFree(base); }
    int reserve(size_t bytes) { MLMC_HIP_CHECK(hipMalloc((void **)&base, bytes)); cap = bytes; return 0; }
    // view of the process-wide workspace of the max-entropy solver (grow-only; solves are serialised on the stream)
    int reserve_shared(size_t bytes) {
        static char *g_base = nullptr;
        static size_t g_cap = 0;
        if (bytes > g_cap) {
            MLMC_HIP_CHECK(wait_stream(rt().stream));
            if (g_base) (void)hipFree(g_base);
            g_base = nullptr;
            g_cap = 0;
            MLMC_HIP_CHECK(hipMalloc((void **)&g_base, bytes));
            g_cap = bytes;
        }
        base = g_base;
        cap = g_cap;
        owner = false;
        return 0;
    }
    static size_t pad(size_t n_doubles) { return ((n_doubles * sizeof(double) + 255) / 256) * 256 + 256; }
    double *take(size_t n_doubles) {
        double *p = (double *)(base + used);
        used += pad(n_doubles);
        return used <= cap ? p : nullptr;
    }
};
struct DevBuf {   // view into a DevPool
    double *p = nullptr;
    double *d() const { return p; }
};

}  // namespace mlmc

using namespace mlmc;

extern "C" {

int mlmc_maxent_solve(const mlmc_basis *b, const double *mu, const double *sigma, int32_t R1, double a, double bnd_b,
                      const mlmc_maxent_opts *opts, const double *prev_lambda, int32_t n_prev, double *lambda_io,
                      double *grad_out, double *hess_out, mlmc_maxent_info *info) {
    MLMC_API_GUARD;
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!b || !mu || !sigma || !opts || !lambda_io || !info) return fail("mlmc_maxent_solve: null argument");
    const int max_out = b->out_size > 0 ? b->out_size : b->p.size;
    if (R1 <= 0 || R1 > max_out) return fail("mlmc_maxent_solve: R1 out of range");
    if (R1 > ME_MAX_R) return fail("mlmc_maxent_solve: at most 128 moments");
    if (!(bnd_b > a)) return fail("mlmc_maxent_solve: empty domain");
    for (int i = 0; i < R1; ++i)
        if (!(sigma[i] > 0.0)) return fail("mlmc_maxent_solve: moment standard errors must be positive");
    hipStream_t st = rt().stream;
    const int deg = opts->gauss_degree > 0 ? opts->gauss_degree : 21;
    const int nint = opts->n_intervals > 0 ? opts->n_intervals : 64;
    const int Q = deg * nint;
    const int max_it = opts->max_it > 0 ? opts->max_it : 100;
    const double tol = opts->tol > 0 ? opts->tol : 1e-8;
    if (n_prev < 0 || n_prev > R1 || (n_prev > 0 && !prev_lambda)) return fail("mlmc_maxent_solve: bad prev_lambda");

    // ---- quadrature points / weights (host), basis matrix Phi (device) ----
    std::vector<double> gx, gw;
    gauss_legendre(deg, gx, gw);
    std::vector<double> xq(Q), wq(Q);
    const double h = (bnd_b - a) / nint;
    for (int k = 0; k < nint; ++k) {
        const double lo = a + k * h, hi = (k == nint - 1) ? bnd_b : a + (k + 1) * h;
        for (int j = 0; j < deg; ++j) {
            xq[k * deg + j] = (gx[j] + 1.0) / 2.0 * (hi - lo) + lo;      // simple_distribution.py:227
            wq[k * deg + j] = gw[j] * (hi - lo) / 2.0;                   // :228
        }
    }
    const int T = (R1 + 15) / 16;
    const int n_dblocks = (Q + 255) / 256;
    DevPool pool;
    // cooperative solver geometry: QS quadrature points per workgroup, NB workgroups
    int QS = R1 <= 64 ? 32 : 16;
    while ((Q + QS - 1) / QS > COOP_MAX_BLOCKS) QS += 8;
    const int NB = (Q + QS - 1) / QS;
    const int R1p = R1 | 1;
    const size_t coop_E = (size_t)R1 * (R1 + 1) / 2 + R1 + 1;
    const size_t coop_alias = std::max((size_t)R1 * (R1 + 1), (size_t)QS * R1p);
    const size_t coop_lds = sizeof(double) * ((size_t)QS * R1p + 9 * (size_t)R1 + 5 * (size_t)QS + 16 + 4 * COOP_MAX_BLOCKS + coop_alias);
    const bool stepwise_forced = getenv("MLMC_MAXENT_STEPWISE") != nullptr;             // validation aid (tests compare both paths)
    const bool coop = !stepwise_forced && coop_lds <= 156 * 1024 && NB <= rt().n_cu;
    const size_t sizes[] = {(size_t)R1, (size_t)Q, (size_t)Q, (size_t)Q * R1, (size_t)Q, (size_t)R1, (size_t)R1, (size_t)R1, (size_t)R1,
                            (size_t)R1 * R1, (size_t)R1, (size_t)R1, 8, (size_t)n_dblocks, (size_t)2 * R1, (size_t)R1 + 1, 4, (size_t)4 * R1,
                            coop ? (size_t)NB * coop_E : 1, coop ? coop_E : 1, (size_t)2 * COOP_MAX_BLOCKS * 4, 16, 8,
                            b->out_size > 0 ? (size_t)Q * b->p.size : 1, (size_t)2 * Q + 3 * (size_t)R1,
                            16 + 2 * (size_t)R1 + (size_t)R1 * R1};
    size_t total = 0;
    for (size_t sz : sizes) total += DevPool::pad(sz);
    if (pool.reserve_shared(total)) return 1;
    DevBuf d_dir, d_x, d_w, d_Phi, d_rhow, d_lam, d_trial, d_p, d_g, d_H, d_mus, d_sig, d_scal, d_bs, d_end, d_prev, d_endpts, d_endphi,
        d_part, d_tot, d_pls, d_res, d_bar, d_evtmp, d_in, d_outblk;
    DevBuf *bufs[] = {&d_dir, &d_x, &d_w, &d_Phi, &d_rhow, &d_lam, &d_trial, &d_p, &d_g, &d_H, &d_mus, &d_sig, &d_scal, &d_bs, &d_end,
                      &d_prev, &d_endpts, &d_endphi, &d_part, &d_tot, &d_pls, &d_res, &d_bar, &d_evtmp, &d_in, &d_outblk};
    for (int k = 0; k < 26; ++k) {
        bufs[k]->p = pool.take(sizes[k]);
        if (!bufs[k]->p) return fail("mlmc_maxent_solve: internal pool overflow");
    }
    // inputs travel as ONE block through a pinned staging buffer: x | w | mu / sigma | sigma | lambda;
    // results come back the same way: result[16] | lambda | gradient | Hessian
    const size_t n_in = 2 * (size_t)Q + 3 * (size_t)R1, n_out = 16 + 2 * (size_t)R1 + (size_t)R1 * R1;
    static double *h_stage = nullptr;
    static size_t h_stage_cap = 0;
    if (n_in + n_out > h_stage_cap) {
        MLMC_HIP_CHECK(wait_stream(st));
        if (h_stage) (void)hipHostFree(h_stage);
        h_stage = nullptr;
        h_stage_cap = 0;
        MLMC_HIP_CHECK(hipHostMalloc((void **)&h_stage, sizeof(double) * (n_in + n_out), hipHostMallocDefault));
        h_stage_cap = n_in + n_out;
    }
    d_x.p = d_in.p;
    d_w.p = d_in.p + Q;
    d_mus.p = d_in.p + 2 * (size_t)Q;
    d_sig.p = d_mus.p + R1;
    d_lam.p = d_sig.p + R1;
    d_res.p = d_outblk.p;
    d_g.p = d_outblk.p + 16 + R1;
    d_H.p = d_g.p + R1;
    std::memcpy(h_stage, xq.data(), sizeof(double) * Q);
    std::memcpy(h_stage + Q, wq.data(), sizeof(double) * Q);
    for (int i = 0; i < R1; ++i) h_stage[2 * (size_t)Q + i] = mu[i] / sigma[i];
    std::memcpy(h_stage + 2 * (size_t)Q + R1, sigma, sizeof(double) * R1);
    std::memcpy(h_stage + 2 * (size_t)Q + 2 * (size_t)R1, lambda_io, sizeof(double) * R1);
    MLMC_HIP_CHECK(hipMemcpyAsync(d_in.p, h_stage, sizeof(double) * n_in, hipMemcpyHostToDevice, st));
    if (n_prev > 0) MLMC_HIP_CHECK(hipMemcpyAsync(d_prev.p, prev_lambda, sizeof(double) * n_prev, hipMemcpyHostToDevice, st));
    if (int rc = launch_eval(b, d_x.d(), Q, R1, d_Phi.d(), b->out_size > 0 ? d_evtmp.d() : nullptr)) return rc;
    hipLaunchKernelGGL(k_me_scale_cols, dim3(((size_t)Q * R1 + 255) / 256), dim3(256), 0, st, d_Phi.d(), d_sig.d(), Q, R1);
    MLMC_HIP_CHECK(hipGetLastError());

    // end-point derivative estimates (simple_distribution.py:240-252 / distribution.py:326-338), eps = 1e-10
    const bool use_pen = (opts->penalty_coef != 0.0) || (opts->stab_penalty != 0.0);
    if (use_pen) {
        const double eps = 1e-10;
        double pts[4] = {a + eps, a, bnd_b, bnd_b - eps};
        std::vector<double> phi(4 * (size_t)R1), ed(2 * (size_t)R1, 0.0);
        MLMC_HIP_CHECK(hipMemcpyAsync(d_endpts.p, pts, sizeof(pts), hipMemcpyHostToDevice, st));
        if (int rc = launch_eval(b, d_endpts.d(), 4, R1, d_endphi.d())) return rc;
        MLMC_HIP_CHECK(hipMemcpyAsync(phi.data(), d_endphi.p, sizeof(double) * 4 * R1, hipMemcpyDeviceToHost, st));
        MLMC_HIP_CHECK(wait_stream(st));
        for (int i = 0; i < R1; ++i) {
            if (opts->decay_left) ed[i] = (phi[i] - phi[R1 + i]) / eps / sigma[i];
            if (opts->decay_right) ed[R1 + i] = (-phi[2 * R1 + i] + phi[3 * R1 + i]) / eps / sigma[i];
        }
        MLMC_HIP_CHECK(hipMemcpyAsync(d_end.p, ed.data(), sizeof(double) * 2 * R1, hipMemcpyHostToDevice, st));
    }

    // ---- the whole iteration as one cooperative launch (k_me_coop); falls back to the step-by-step loop below ----
    if (coop) {
        MLMC_HIP_CHECK(hipFuncSetAttribute((const void *)k_me_coop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)coop_lds));
        MLMC_HIP_CHECK(hipMemsetAsync(d_bar.p, 0, 64, st));
        CoopArgs ca;
        ca.Phi = d_Phi.d(); ca.w = d_w.d(); ca.mu_s = d_mus.d(); ca.sigma = d_sig.d();
        ca.lam = d_lam.d(); ca.g_out = d_g.d(); ca.H_out = d_H.d(); ca.result = d_res.d();
        ca.part = d_part.d(); ca.tot = d_tot.d(); ca.part_ls = d_pls.d(); ca.bar = (unsigned *)d_bar.p;
        ca.Q = Q; ca.R1 = R1; ca.NB = NB; ca.QS = QS; ca.max_it = max_it; ca.tol = tol;
        ca.end_diff = d_end.d(); ca.prev = d_prev.d(); ca.n_prev = n_prev; ca.use_pen = use_pen ? 1 : 0;
        ca.stab = opts->stab_penalty; ca.coef = opts->penalty_coef;
        hipLaunchKernelGGL(k_me_coop, dim3(NB), dim3(COOP_THREADS), coop_lds, st, ca);
        MLMC_HIP_CHECK(hipGetLastError());
        // the final multipliers are written next to the result block: one copy brings everything back
        hipLaunchKernelGGL(k_me_axpy, dim3((R1 + 127) / 128), dim3(128), 0, st, d_lam.d(), d_lam.d(), 0.0, R1, d_outblk.p + 16);
        double *res = h_stage + n_in;
        MLMC_HIP_CHECK(hipMemcpyAsync(res, d_outblk.p, sizeof(double) * n_out, hipMemcpyDeviceToHost, st));
        MLMC_HIP_CHECK(wait_stream(st));
#ifdef MLMC_PROF_COOP
        fprintf(stderr, "coop stamps (us): A %.1f | bar1 %.1f | B %.1f | bar2 %.1f | C %.1f = load %.1f + ldl %.1f + solve %.1f (NB %d QS %d R1 %d)\n", res[9], res[10] - res[9],
                res[11] - res[10], res[12] - res[11], res[13] - res[12], res[14] - res[12], res[15] - res[14], res[13] - res[15], NB, QS, R1);
#endif
        if (res[5] == 0.0) {
            std::memcpy(lambda_io, res + 16, sizeof(double) * R1);
            if (grad_out) std::memcpy(grad_out, res + 16 + R1, sizeof(double) * R1);
            if (hess_out) std::memcpy(hess_out, res + 16 + 2 * (size_t)R1, sizeof(double) * (size_t)R1 * R1);
            info->nit = (int)res[0];
            info->success = (int)res[1];
            info->fun = res[2];
            info->grad_norm = res[3];
            info->moment0 = res[4];
            info->n_quad = Q;
            info->reserved = 0;
            return 0;
        }
        // the grid was not co-resident (another tenant held the CUs): start again from the caller's multipliers, step by step
        MLMC_HIP_CHECK(hipMemcpyAsync(d_lam.p, lambda_io, sizeof(double) * R1, hipMemcpyHostToDevice, st));
    }

    double scal[8];
    auto eval_F = [&](const double *d_l, int grad) -> int {   // scal[0] = F (penalties excluded) at d_l; gradient if asked
        hipLaunchKernelGGL(k_me_density, dim3(n_dblocks), dim3(256), 0, st, d_Phi.d(), d_w.d(), d_l, Q, R1, d_rhow.d(), d_bs.d());
        hipLaunchKernelGGL(k_me_grad, dim3(grad ? R1 : 1), dim3(256), 0, st, d_Phi.d(), d_rhow.d(), d_mus.d(), d_l, d_sig.d(), Q, R1,
                           d_bs.d(), n_dblocks, grad, d_g.d(), d_scal.d());
        MLMC_HIP_CHECK(hipGetLastError());
        return 0;
    };
    auto eval_full = [&](const double *d_l, double tau) -> int {
        if (int rc = eval_F(d_l, 1)) return rc;
        hipLaunchKernelGGL(k_me_hessian, dim3(T * (T + 1) / 2), dim3(256), 0, st, d_Phi.d(), d_rhow.d(), Q, R1, T, d_H.d());
        if (use_pen)
            hipLaunchKernelGGL(k_me_penalty, dim3(1), dim3(256), 0, st, d_end.d(), d_l, d_prev.d(), n_prev, opts->stab_penalty,
                               opts->penalty_coef, d_mus.d(), d_sig.d(), R1, d_g.d(), d_H.d(), d_scal.d());
        const size_t lds = sizeof(double) * ((size_t)R1 * (R1 + 1) + R1);
        hipLaunchKernelGGL(k_me_solve, dim3(1), dim3(256), lds, st, d_H.d(), d_g.d(), R1, tau, d_p.d(), d_scal.d());
        MLMC_HIP_CHECK(hipGetLastError());
        MLMC_HIP_CHECK(hipMemcpyAsync(scal, d_scal.p, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
        MLMC_HIP_CHECK(wait_stream(st));
        return 0;
    };
    {   // the Cholesky kernel needs up to 128*129*8 + 1 KiB of dynamic LDS
        const size_t lds = sizeof(double) * ((size_t)R1 * (R1 + 1) + R1);
        MLMC_HIP_CHECK(hipFuncSetAttribute((const void *)k_me_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }

    int nit = 0, success = 0;
    double tau = 0.0;
    double F = 0.0, gnorm = 0.0;
    for (int it = 0; it <= max_it; ++it) {
        if (int rc = eval_full(d_lam.d(), tau)) return rc;
        F = scal[0];
        gnorm = scal[3];
        if (!(gnorm == gnorm) || !(F == F)) break;                    // NaN: give up (success = 0)
        if (gnorm < tol) { success = 1; break; }
        if (it == max_it) break;
        if (scal[5] != 0.0 || !(scal[4] < 0.0)) {                     // not SPD / not a descent direction: regularise
            tau = (tau == 0.0) ? 1e-10 * (1.0 + fabs(F)) : tau * 100.0;
            if (tau > 1e20) break;
            continue;
        }
        // Armijo backtracking on F (exact when no penalties; with penalties the reference's own approximate gradient
        // is used, so fall back to decreasing the gradient norm)
        const double gp = scal[4];
        MLMC_HIP_CHECK(hipMemcpyAsync(d_dir.p, d_p.p, sizeof(double) * R1, hipMemcpyDeviceToDevice, st));
        double alpha = 1.0;
        bool accepted = false;
        for (int ls = 0; ls < 40; ++ls) {
            hipLaunchKernelGGL(k_me_axpy, dim3((R1 + 127) / 128), dim3(128), 0, st, d_lam.d(), d_dir.d(), alpha, R1, d_trial.d());
            double Ft, gt;
            if (!use_pen) {
                if (int rc = eval_F(d_trial.d(), 0)) return rc;
                double s2[8];
                MLMC_HIP_CHECK(hipMemcpyAsync(s2, d_scal.p, sizeof(double) * 8, hipMemcpyDeviceToHost, st));
                MLMC_HIP_CHECK(wait_stream(st));
                Ft = s2[0];
                if (Ft == Ft && Ft <= F + 1e-4 * alpha * gp) accepted = true;
            } else {
                double keep[8];
                for (int k = 0; k < 8; ++k) keep[k] = scal[k];
                if (int rc = eval_full(d_trial.d(), tau)) return rc;
                gt = scal[3];
                for (int k = 0; k < 8; ++k) scal[k] = keep[k];
                if (gt == gt && gt < gnorm) accepted = true;
                (void)Ft;
            }
            if (accepted) break;
            alpha *= 0.5;
        }
        if (!accepted) {
            tau = (tau == 0.0) ? 1e-8 * (1.0 + fabs(F)) : tau * 100.0;
            if (tau > 1e20) break;
            continue;
        }
        MLMC_HIP_CHECK(hipMemcpyAsync(d_lam.p, d_trial.p, sizeof(double) * R1, hipMemcpyDeviceToDevice, st));
        tau = (alpha == 1.0) ? tau * 0.1 : tau;
        if (tau < 1e-14) tau = 0.0;
        ++nit;
    }
    MLMC_HIP_CHECK(hipMemcpyAsync(lambda_io, d_lam.p, sizeof(double) * R1, hipMemcpyDeviceToHost, st));
    if (grad_out) MLMC_HIP_CHECK(hipMemcpyAsync(grad_out, d_g.p, sizeof(double) * R1, hipMemcpyDeviceToHost, st));
    if (hess_out) MLMC_HIP_CHECK(hipMemcpyAsync(hess_out, d_H.p, sizeof(double) * (size_t)R1 * R1, hipMemcpyDeviceToHost, st));
    MLMC_HIP_CHECK(wait_stream(st));
    info->nit = nit;
    info->success = success;
    info->fun = F;
    info->grad_norm = gnorm;
    info->moment0 = scal[1];
    info->n_quad = Q;
    info->reserved = 0;
    return 0;
}

int mlmc_density_eval(const mlmc_basis *b, const double *lambda, const double *sigma, int32_t R1, const double *x, int64_t n,
                      double *out, int mem_kind) {
    MLMC_API_GUARD;
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!b || !lambda || !sigma || (n > 0 && (!x || !out))) return fail("mlmc_density_eval: null argument");
    const int max_out = b->out_size > 0 ? b->out_size : b->p.size;
    if (R1 <= 0 || R1 > max_out) return fail("mlmc_density_eval: R1 out of range");
    if (n == 0) return 0;
    hipStream_t st = rt().stream;
    const int R = b->p.size;
    std::vector<double> c = effective_coeffs(b, lambda, sigma, R1);
    const int Reff = b->out_size > 0 ? R : R1;
    DevPool pool;
    if (pool.reserve(DevPool::pad(R) + 2 * DevPool::pad((size_t)n))) return 1;
    DevBuf d_c, d_x, d_o;
    d_c.p = pool.take(R);
    MLMC_HIP_CHECK(hipMemcpyAsync(d_c.p, c.data(), sizeof(double) * R, hipMemcpyHostToDevice, st));
    const double *xd = x;
    double *od = out;
    if (mem_kind == MLMC_HOST) {
        d_x.p = pool.take((size_t)n);
        d_o.p = pool.take((size_t)n);
        MLMC_HIP_CHECK(hipMemcpyAsync(d_x.p, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
        xd = d_x.d();
        od = d_o.d();
    }
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    switch (b->p.kind) {
        case MLMC_LEGENDRE: hipLaunchKernelGGL(k_density<MLMC_LEGENDRE>, grid, block, 0, st, b->p, d_c.d(), Reff, xd, n, od); break;
        case MLMC_MONOMIAL: hipLaunchKernelGGL(k_density<MLMC_MONOMIAL>, grid, block, 0, st, b->p, d_c.d(), Reff, xd, n, od); break;
        case MLMC_FOURIER: hipLaunchKernelGGL(k_density<MLMC_FOURIER>, grid, block, 0, st, b->p, d_c.d(), Reff, xd, n, od); break;
        case MLMC_SPLINE: hipLaunchKernelGGL(k_density<MLMC_SPLINE>, grid, block, 0, st, b->p, d_c.d(), Reff, xd, n, od); break;
        default: return fail("mlmc_density_eval: unsupported basis kind");
    }
    MLMC_HIP_CHECK(hipGetLastError());
    if (mem_kind == MLMC_HOST) MLMC_HIP_CHECK(hipMemcpyAsync(out, d_o.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    MLMC_HIP_CHECK(wait_stream(st));
    return 0;
}

int mlmc_density_integrate(const mlmc_basis *b, const double *lambda, const double *sigma, int32_t R1, const double *lo,
                           const double *hi, int64_t n, int32_t degree, double *out) {
    MLMC_API_GUARD;
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!b || !lambda || !sigma || (n > 0 && (!lo || !hi || !out))) return fail("mlmc_density_integrate: null argument");
    const int max_out = b->out_size > 0 ? b->out_size : b->p.size;
    if (R1 <= 0 || R1 > max_out) return fail("mlmc_density_integrate: R1 out of range");
    if (degree <= 0 || degree > 64) return fail("mlmc_density_integrate: degree must be in 1..64");
    if (n == 0) return 0;
    hipStream_t st = rt().stream;
    const int R = b->p.size;
    std::vector<double> c = effective_coeffs(b, lambda, sigma, R1);
    const int Reff = b->out_size > 0 ? R : R1;
    std::vector<double> gx, gw;
    gauss_legendre(degree, gx, gw);
    DevPool pool;
    if (pool.reserve(DevPool::pad(R) + 3 * DevPool::pad((size_t)n) + 2 * DevPool::pad(degree))) return 1;
    DevBuf d_c, d_lo, d_hi, d_o, d_gx, d_gw;
    d_c.p = pool.take(R);
    d_lo.p = pool.take((size_t)n);
    d_hi.p = pool.take((size_t)n);
    d_o.p = pool.take((size_t)n);
    d_gx.p = pool.take(degree);
    d_gw.p = pool.take(degree);
    MLMC_HIP_CHECK(hipMemcpyAsync(d_c.p, c.data(), sizeof(double) * R, hipMemcpyHostToDevice, st));
    MLMC_HIP_CHECK(hipMemcpyAsync(d_lo.p, lo, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    MLMC_HIP_CHECK(hipMemcpyAsync(d_hi.p, hi, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    MLMC_HIP_CHECK(hipMemcpyAsync(d_gx.p, gx.data(), sizeof(double) * degree, hipMemcpyHostToDevice, st));
    MLMC_HIP_CHECK(hipMemcpyAsync(d_gw.p, gw.data(), sizeof(double) * degree, hipMemcpyHostToDevice, st));
    const dim3 grid((unsigned)((n + 127) / 128)), block(128);
    switch (b->p.kind) {
        case MLMC_LEGENDRE: hipLaunchKernelGGL(k_density_integrate<MLMC_LEGENDRE>, grid, block, 0, st, b->p, d_c.d(), Reff, d_lo.d(), d_hi.d(), n, d_gx.d(), d_gw.d(), degree, d_o.d()); break;
        case MLMC_MONOMIAL: hipLaunchKernelGGL(k_density_integrate<MLMC_MONOMIAL>, grid, block, 0, st, b->p, d_c.d(), Reff, d_lo.d(), d_hi.d(), n, d_gx.d(), d_gw.d(), degree, d_o.d()); break;
        case MLMC_FOURIER: hipLaunchKernelGGL(k_density_integrate<MLMC_FOURIER>, grid, block, 0, st, b->p, d_c.d(), Reff, d_lo.d(), d_hi.d(), n, d_gx.d(), d_gw.d(), degree, d_o.d()); break;
        case MLMC_SPLINE: hipLaunchKernelGGL(k_density_integrate<MLMC_SPLINE>, grid, block, 0, st, b->p, d_c.d(), Reff, d_lo.d(), d_hi.d(), n, d_gx.d(), d_gw.d(), degree, d_o.d()); break;
        default: return fail("mlmc_density_integrate: unsupported basis kind");
    }
    MLMC_HIP_CHECK(hipGetLastError());
    MLMC_HIP_CHECK(hipMemcpyAsync(out, d_o.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    MLMC_HIP_CHECK(wait_stream(st));
    return 0;
}

}  // extern "C"
