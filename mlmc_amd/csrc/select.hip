// Order statistics / percentiles of a sample array on the device (gfx950).
//
// Reference: Estimate.estimate_domain (mlmc/estimator.py:275-302) removes NaNs from the fine samples of a level and
// calls np.percentile(fine, [100 q, 100 (1 - q)]) (NumPy "linear" method: interpolation between the two neighbouring
// order statistics).  Here the two neighbours are found exactly by a most-significant-digit radix select on an
// order-preserving 64-bit key (11-bit digits, per-block LDS histograms, 6 passes over the data), and the
// interpolation is done with NumPy's own formula, so the result is bit-identical to np.percentile.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace mlmc {

constexpr int SEL_BITS = 11;
constexpr int SEL_BINS = 1 << SEL_BITS;

__device__ __forceinline__ unsigned long long order_key(double x) {
    unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);   // monotone in x for all non-NaN values (-0.0 < +0.0)
}

// histogram of the digit at `shift` over the keys whose bits above (shift + SEL_BITS) equal `prefix`
__global__ __launch_bounds__(256) void k_select_hist(const double *__restrict__ x, int64_t n, unsigned long long prefix,
                                                     int shift, int width, int first, unsigned int *__restrict__ hist,
                                                     unsigned long long *__restrict__ n_valid) {
    __shared__ unsigned int lh[SEL_BINS];
    for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const int hi_shift = shift + width;
    const unsigned mask = (1u << width) - 1u;
    unsigned long long valid = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        if (v != v) continue;                       // NaNs are removed (estimator.py:298)
        ++valid;
        const unsigned long long key = order_key(v);
        const bool match = first || ((key >> hi_shift) == prefix);
        if (match) atomicAdd(&lh[(unsigned)(key >> shift) & mask], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
    if (first) {   // exact integer count, order independent
        for (int off = 32; off >= 1; off >>= 1) valid += __shfl_xor(valid, off, 64);
        if ((threadIdx.x & 63) == 0 && valid) atomicAdd(n_valid, valid);
    }
}

static double key_to_double(unsigned long long key) {
    unsigned long long u = (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;
    double d;
    std::memcpy(&d, &u, sizeof(d));
    return d;
}

// k-th smallest (0-based) non-NaN value of d_x[0..n)
static int select_kth(const double *d_x, int64_t n, int64_t k, unsigned int *d_hist, unsigned long long *d_nvalid, double *out,
                      int64_t *n_valid_out) {
    hipStream_t st = rt().stream;
    std::vector<unsigned int> hist(SEL_BINS);
    unsigned long long prefix = 0;
    int blocks = rt().n_cu * 4;
    const int64_t want = (n + 255) / 256;
    if (want < blocks) blocks = (int)(want > 0 ? want : 1);
    // digits from the top: bits [53, 64), [42, 53), [31, 42), [20, 31), [9, 20), then the last 9 bits
    const int shifts[6] = {53, 42, 31, 20, 9, 0};
    for (int pass = 0; pass < 6; ++pass) {
        const int shift = shifts[pass];
        const int width = (pass == 5) ? 9 : SEL_BITS;
        MLMC_HIP_CHECK(hipMemsetAsync(d_hist, 0, sizeof(unsigned int) * SEL_BINS, st));
        if (pass == 0) MLMC_HIP_CHECK(hipMemsetAsync(d_nvalid, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_select_hist, dim3(blocks), dim3(256), 0, st, d_x, n, prefix, shift, width, pass == 0 ? 1 : 0, d_hist, d_nvalid);
        MLMC_HIP_CHECK(hipGetLastError());
        MLMC_HIP_CHECK(hipMemcpyAsync(hist.data(), d_hist, sizeof(unsigned int) * SEL_BINS, hipMemcpyDeviceToHost, st));
        if (pass == 0) {
            unsigned long long nv = 0;
            MLMC_HIP_CHECK(hipMemcpyAsync(&nv, d_nvalid, sizeof(nv), hipMemcpyDeviceToHost, st));
            MLMC_HIP_CHECK(hipStreamSynchronize(st));
            if (n_valid_out) *n_valid_out = (int64_t)nv;
            if (k < 0 || (unsigned long long)k >= nv) return fail("order statistic index out of range");
        } else {
            MLMC_HIP_CHECK(hipStreamSynchronize(st));
        }
        int64_t cum = 0;
        int digit = -1;
        const int nbins = 1 << width;
        for (int b = 0; b < nbins; ++b) {
            if (k < cum + (int64_t)hist[b]) { digit = b; break; }
            cum += hist[b];
        }
        if (digit < 0) return fail("radix select: inconsistent histogram");
        k -= cum;
        prefix = (prefix << width) | (unsigned long long)digit;
    }
    *out = key_to_double(prefix);
    return 0;
}

}  // namespace mlmc

using namespace mlmc;

extern "C" int mlmc_percentiles(const double *x, int64_t n, const double *q_percent, int32_t nq, double *out, int64_t *n_valid,
                                int mem_kind) {
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!x || !q_percent || !out || nq <= 0) return fail("mlmc_percentiles: bad argument");
    if (n <= 0) return fail("mlmc_percentiles: empty input");
    hipStream_t st = rt().stream;
    double *d_x = nullptr;
    bool own = false;
    if (mem_kind == MLMC_HOST) {
        MLMC_HIP_CHECK(hipMalloc(&d_x, sizeof(double) * (size_t)n));
        own = true;
        hipError_t e = hipMemcpyAsync(d_x, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { (void)hipFree(d_x); return fail("mlmc_percentiles: H2D copy failed"); }
    } else {
        d_x = const_cast<double *>(x);
    }
    unsigned int *d_hist = nullptr;
    if (hipMalloc(&d_hist, sizeof(unsigned int) * SEL_BINS + 64) != hipSuccess) {
        if (own) (void)hipFree(d_x);
        return fail("mlmc_percentiles: hipMalloc failed");
    }
    unsigned long long *d_nvalid = (unsigned long long *)(d_hist + SEL_BINS);
    int rc = 0;
    int64_t nv = 0;
    double lowest = 0.0;
    rc = select_kth(d_x, n, 0, d_hist, d_nvalid, &lowest, &nv);    // also counts the non-NaN values
    for (int i = 0; i < nq && !rc; ++i) {
        // NumPy: q = percent / 100 ; virtual index = (n - 1) * q ; linear interpolation between the neighbours
        const double q = q_percent[i] / 100.0;
        if (!(q >= 0.0 && q <= 1.0)) { rc = fail("mlmc_percentiles: percentiles must be in [0, 100]"); break; }
        const double vi = (double)(nv - 1) * q;
        int64_t prev = (int64_t)std::floor(vi);
        double gamma = vi - (double)prev;
        if (prev < 0) { prev = 0; gamma = 0.0; }
        int64_t next = prev + 1;
        if (next > nv - 1) next = nv - 1;
        double a = lowest, b;
        if (prev > 0) rc = select_kth(d_x, n, prev, d_hist, d_nvalid, &a, nullptr);
        if (rc) break;
        b = a;
        if (next != prev) rc = select_kth(d_x, n, next, d_hist, d_nvalid, &b, nullptr);
        if (rc) break;
        const double diff = b - a;
        double r = a + diff * gamma;                      // numpy.lib._function_base_impl._lerp
        if (gamma >= 0.5) r = b - diff * (1.0 - gamma);
        out[i] = r;
    }
    if (n_valid) *n_valid = nv;
    (void)hipFree(d_hist);
    if (own) (void)hipFree(d_x);
    return rc;
}

// ------------------------------------------------------------------------------------------
// Bootstrap sub-sampling: gather of k uniformly drawn columns (with replacement) of a resident chunk.
// Philox4x32-10 (Salmon et al., SC'11) keyed by the seed, counter = draw index: every draw is independent of the launch
// geometry.  index = high 64 bits of (64 random bits x n): uniform on [0, n) up to a bias below n / 2^64.
// ------------------------------------------------------------------------------------------
namespace mlmc {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ uint64_t philox_u64(uint64_t counter, uint64_t seed) {
    uint32_t c[4] = {(uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return ((uint64_t)c[1] << 32) | c[0];
}

__global__ __launch_bounds__(256) void k_subsample_gather(const double *__restrict__ fine, const double *__restrict__ coarse,
                                                          int n_rows, int64_t n, int64_t k, uint64_t seed,
                                                          double *__restrict__ fine_out, double *__restrict__ coarse_out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const int64_t idx = (int64_t)__umul64hi(philox_u64((uint64_t)j, seed), (uint64_t)n);
    for (int r = 0; r < n_rows; ++r) {
        fine_out[(int64_t)r * k + j] = fine[(int64_t)r * n + idx];
        if (coarse) coarse_out[(int64_t)r * k + j] = coarse[(int64_t)r * n + idx];
    }
}

}  // namespace mlmc

extern "C" int mlmc_subsample_gather(const double *fine, const double *coarse, int32_t n_rows, int64_t n, int64_t k,
                                     uint64_t seed, double *fine_out, double *coarse_out) {
    using namespace mlmc;
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!fine || !fine_out || (coarse && !coarse_out)) return fail("mlmc_subsample_gather: null argument");
    if (n_rows < 1 || n < 1 || k < 0) return fail("mlmc_subsample_gather: sizes out of range");
    if (k == 0) return 0;
    hipLaunchKernelGGL(k_subsample_gather, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, rt().stream, fine, coarse, n_rows, n, k,
                       seed, fine_out, coarse_out);
    MLMC_HIP_CHECK(hipGetLastError());
    return 0;
}
