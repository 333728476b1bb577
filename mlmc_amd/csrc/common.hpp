// Shared host/device declarations of libmlmc_hip.so (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mlmc_hip.h"

namespace mlmc {

constexpr int WAVE = 64;
constexpr int ACC_THREADS = 256;   // 4 waves, one per SIMD of a CU
#ifndef MLMC_TERMS_PER_PASS
#define MLMC_TERMS_PER_PASS 64
#endif
constexpr int MAX_TERMS_PER_PASS = MLMC_TERMS_PER_PASS;
constexpr int SPLINE_BAND_MAX_R = 256;   // banded mean-only covariance of spline moments (cov.hip): 4 x 5 x (R + 8) doubles of LDS

// ---- error plumbing ---------------------------------------------------------------------
void set_error(const std::string &msg);
int fail(const std::string &msg);
#define MLMC_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return ::mlmc::fail(std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

// One lock for the whole C ABI.  The library works on ONE stream with shared workspaces (the Runtime, the scratch and
// staging buffers of an accumulator, the grow-only pools of the solver, the occupancy caches), so its entry points are
// serialised: host threads may call in freely (ctypes releases the GIL), their calls queue up here.  Recursive because
// the one-call forms (mlmc_accum_estimate, ...) go through other entry points.  Every extern "C" function that touches the
// device takes it first (MLMC_API_GUARD); mlmc_last_error is thread-local and needs none.
std::recursive_mutex &api_mutex();
// HIP's current device is a property of the calling host thread (default 0), while the library's stream, scratch and
// modules live on the device mlmc_init bound: every entry point re-binds the calling thread (a thread-local store, no
// driver call when it is already current), so worker threads of a rank with LOCAL_RANK > 0 allocate and launch on the
// right GPU.
void bind_thread_to_device();
struct ApiGuard {
    std::lock_guard<std::recursive_mutex> lock;
    ApiGuard() : lock(api_mutex()) { bind_thread_to_device(); }
};
#define MLMC_API_GUARD ::mlmc::ApiGuard mlmc_api_guard_

struct Runtime {
    bool ready = false;
    int device = -1;
    int flags = 0;
    int n_cu = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipDeviceProp_t prop;
};
Runtime &rt();
// Wait for the stream with a bounded busy poll before blocking: the runtime's own wait spins for ~50 us and then sleeps on
// the completion interrupt, whose wake-up (20-60 us, more from a deep CPU idle state) would dominate a 0.2 ms estimate.
hipError_t wait_stream(hipStream_t st);

// ---- device-visible basis parameters (passed by value to kernels) -------------------------
struct BasisParams {
    int kind;
    int size;      // R of the underlying family
    double shift, scale, ref0, ref1;
    int is_log, is_clip;
    double x_lo, x_hi;   // is_log && is_clip: keep <=> x_lo <= x <= x_hi on the raw value (host-bisected, mlmc_hip.h)
};

}  // namespace mlmc

struct mlmc_basis {
    mlmc::BasisParams p;
    int out_size = 0;                 // 0 = no transform
    std::vector<double> matrix;       // [out_size][size] host copy
    std::vector<double> scale_c;      // host: P_i = scale_c[i] * Q_i (Legendre: leading coefficients; else 1)
    double *d_scale = nullptr;        // device: scale_c [size]
    double *d_matrix = nullptr;       // device: matrix [out_size][size]
};

struct PendingSeg {   // a pushed chunk waiting for the next accumulation launch
    const double *fine, *coarse;
    const uint8_t *mask;
    int64_t n;
    int level, comp;
    bool count;
};

struct mlmc_accum {
    const mlmc_basis *basis = nullptr;
    int n_levels = 0, mode = 0, n_comp = 1;
    bool mean_only = false;       // MLMC_MODE_MEAN_ONLY: the second moments (sp) are not accumulated
    bool cov_from_values = false; // COV: accumulated from materialised moment values (TransformedMoments; plain bases with R > 128)
    bool mean_only_plain = false; // ... requested for MOMENTS of a plain basis: honoured by the 65..256-term split kernel (sp = NaN)
    int R = 0;            // underlying family size
    int Rout = 0;         // rows per component seen by the caller (transform applied)
    int64_t K = 0;        // caller rows per level = n_comp * Rout (* Rout)
    // internal totals per (level, component): MOMENTS: [2][R] (sum d, sum d^2) (+ [R][R] diff Gram if transform)
    //                                         COV: [3][RP][RP] (G0, G1, G2)
    int64_t int_width = 0;        // doubles per (level, comp)
    void *d_state = nullptr;      // one allocation: totals | counts (reset = one memset)
    size_t state_bytes = 0;
    double *d_totals = nullptr;   // [n_levels][n_comp][int_width]
    int64_t *d_counts = nullptr;  // [n_levels][2]  (kept, removed)
    // scratch
    double *d_partials = nullptr; size_t partials_cap = 0;
    int64_t *d_pcounts = nullptr; size_t pcounts_cap = 0;
    double *d_stage_f = nullptr, *d_stage_c = nullptr; size_t stage_cap = 0;
    uint8_t *d_mask = nullptr; size_t mask_cap = 0;
    double *d_vals_f = nullptr, *d_vals_c = nullptr; size_t vals_cap = 0;   // materialised moment values (COV of TransformedMoments)
    double *d_vals_tmp = nullptr; size_t vals_tmp_cap = 0;                  // underlying values of a transformed basis, one side at a time
    void *d_out = nullptr;        // finalize outputs, one allocation: n[L] | n_rm[L] (int64) | n, n_rm as fp64 [2L] | s[L*K] | sp[L*K]
    size_t out_bytes = 0;
    void *h_out = nullptr;        // pinned host mirror of d_out (device-mapped)
    // MOMENTS with a plain basis and one component: k_reduce_partials writes the finished rows straight into h_out
    bool host_outputs = false;
    double *packed_target = nullptr;   // set for the duration of a flush started by mlmc_accum_finalize_packed(MLMC_DEVICE)
    double *h_out_s = nullptr, *h_out_sp = nullptr; int64_t *h_out_n = nullptr;   // device-side addresses of h_out's parts
    std::vector<char> level_flushed;   // levels whose rows in h_out are current (set by flush_moments, cleared by reset)
    double *d_out_s = nullptr, *d_out_sp = nullptr, *d_out_nd = nullptr; int64_t *d_out_n = nullptr;
    // timing
    std::vector<hipEvent_t> ev;   // pairs (start, stop)
    size_t ev_used = 0;
    double ms_total = 0;
    int64_t launches = 0, alg_bytes = 0;
    int64_t mfma_flops = 0;   // executed matrix-core flops of the timed covariance launches (mlmc_accum_kernel_flops)
    int RP = 0;  // COV: R padded to 16
    // COV with variances of 17..128 plain Legendre / monomial moments: the MEAN of the covariance comes from the level sums of
    // the 2 R - 1 moments of the product linearisation (phi_i phi_j = sum_k c_ijk phi_k) -- an inner mean-only MOMENTS
    // accumulator over the same chunks -- and the matrix cores compute G1, G2 only (26 instead of 42 tiles per pair)
    bool lin_eligible = false, lin0_eligible = false;   // decided at create; the inner accumulators appear with the first large chunk
    mlmc_accum *lin = nullptr;
    mlmc_basis *lin_basis = nullptr;      // the family's member of size lin_K = 2 R - 1
    double *d_lin_prod = nullptr;         // [lin_K][R * R]: c_ijk, k-major (shared per family and size, not owned)
    int lin_K = 0;
    // ... and at LEVEL 0 (one value per sample) the variance linearises as well: (phi_i phi_j)^2 = sum_k c2_ijk phi_k, k < 4 R - 3, so a
    // large level-0 chunk of <= 64 moments needs no matrix pass at all -- a second inner accumulator over the size-(4 R - 3) member
    // (two windows of the mean-only kernel) gives sum f_i f_j (its first 2 R - 1 sums) and sum (f_i f_j)^2
    mlmc_accum *lin0 = nullptr;
    mlmc_basis *lin0_basis = nullptr;
    double *d_lin0_prod = nullptr;        // [lin0_K][R * R]: c2_ijk, k-major
    int lin0_K = 0;
    bool lin0_used = false;
    int64_t lin_min_n = 0;                // chunks with fewer samples keep all three Gram matrices on the matrix cores (the
                                          // extra launches of the auxiliary pass cost more than they save); sums are additive
    bool lin_used = false;                // a chunk of this estimate went the linearised way
    std::vector<PendingSeg> pending;   // MOMENTS: chunks gathered into one launch (flushed by finalize / conflicts)
};

namespace mlmc {
// moments.hip
int launch_eval(const mlmc_basis *b, const double *d_x, int64_t n, int size, double *d_out, double *scratch = nullptr);
int launch_eval_scaled_base(const mlmc_basis *b, const double *d_x, int64_t n, double *d_out);
int launch_mask(const mlmc_accum *a, const double *d_f, const double *d_c, int64_t n, uint8_t *d_mask, int64_t *d_counts_level);
int launch_moments_accum(mlmc_accum *a, int level, int comp, const double *d_f, const double *d_c, const uint8_t *d_mask,
                         int64_t n, bool count, bool defer);
int flush_moments(mlmc_accum *a);
int launch_moments_finalize(mlmc_accum *a);
// cov.hip
int launch_cov_accum(mlmc_accum *a, int level, int comp, const double *d_f, const double *d_c, const uint8_t *d_mask,
                     int64_t n, bool count, int gram_mode, int ncomp = 1);
int launch_cov_finalize(mlmc_accum *a);
// c_ijk of the product linearisation, k-major [2 R - 1][R * R]; false: the family has none here
bool product_table(int kind, int R, std::vector<double> &out);
// c2_ijk of (phi_i phi_j)^2 = sum_k c2_ijk phi_k, k-major [4 R - 3][R * R]
bool square_product_table(int kind, int R, std::vector<double> &out);
int launch_cov_from_values(mlmc_accum *a, int level, int comp, const double *d_vf, const double *d_vc, const uint8_t *d_mask,
                           int64_t n, bool count, int gram_mode = 0);
int ensure(void **p, size_t *cap, size_t bytes);
int timing_begin(mlmc_accum *a);
int timing_end(mlmc_accum *a);
}  // namespace mlmc
