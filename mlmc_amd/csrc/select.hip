// Order statistics / percentiles of a sample array on the device (gfx950).
//
// Reference: Estimate.estimate_domain (mlmc/estimator.py:275-302) removes NaNs from the fine samples of a level and
// calls np.percentile(fine, [100 q, 100 (1 - q)]) (NumPy "linear" method: interpolation between the two neighbouring
// order statistics).  Here the two neighbours are found exactly by a most-significant-digit radix select on an
// order-preserving 64-bit key (11-bit digits, per-block LDS histograms; two digit passes over the data for all ranks, one
// pass that collects the few values left under the 22-bit prefixes for a host finish -- six digit passes with heavy ties), and the
// interpolation is done with NumPy's own formula, so the result is bit-identical to np.percentile.
#include <algorithm>
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

constexpr int SEL_MAX_PREFIX = 4;   // distinct key prefixes followed in one pass (LDS: 4 x 2048 counters = 32 KB)

struct SelPrefixes {
    unsigned long long prefix[SEL_MAX_PREFIX];
    int n;
};

// One pass of the most-significant-digit radix select for up to SEL_MAX_PREFIX targets at once: histogram of the digit
// at `shift` over the keys whose higher bits equal prefix[t] (first pass: no prefix, a single histogram of all non-NaN
// values -- its total is the number of valid values).
__global__ __launch_bounds__(256) void k_select_hist(const double *__restrict__ x, int64_t n, SelPrefixes pf, int shift,
                                                     int width, int first, unsigned int *__restrict__ hist) {
    __shared__ unsigned int lh[SEL_MAX_PREFIX][SEL_BINS];
    for (int i = threadIdx.x; i < SEL_MAX_PREFIX * SEL_BINS; i += blockDim.x) (&lh[0][0])[i] = 0;
    __syncthreads();
    const int hi_shift = shift + width;
    const unsigned mask = (1u << width) - 1u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        if (v != v) continue;                       // NaNs are removed (estimator.py:298)
        const unsigned long long key = order_key(v);
        const unsigned digit = (unsigned)(key >> shift) & mask;
        if (first) {
            atomicAdd(&lh[0][digit], 1u);
        } else {
            const unsigned long long hi = key >> hi_shift;
#pragma unroll
            for (int t = 0; t < SEL_MAX_PREFIX; ++t)
                if (t < pf.n && hi == pf.prefix[t]) atomicAdd(&lh[t][digit], 1u);
        }
    }
    __syncthreads();
    const int n_hist = first ? 1 : pf.n;
    for (int i = threadIdx.x; i < n_hist * SEL_BINS; i += blockDim.x) {
        const unsigned v = (&lh[0][0])[i];
        if (v) atomicAdd(&hist[i], v);
    }
}

// After two digit passes a prefix of 22 bits usually holds a handful of values: the keys under up to SEL_MAX_PREFIX prefixes
// are appended to a small buffer (order irrelevant) and the selection finishes on the host -- three scans of the data
// instead of six.  Heavily repeated values overflow `cap`; the caller then keeps going digit by digit.
__global__ __launch_bounds__(256) void k_select_collect(const double *__restrict__ x, int64_t n, SelPrefixes pf, int hi_shift,
                                                        unsigned long long *__restrict__ out, unsigned int *__restrict__ count,
                                                        unsigned int cap) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        if (v != v) continue;
        const unsigned long long key = order_key(v);
        const unsigned long long hi = key >> hi_shift;
        bool hit = false;
#pragma unroll
        for (int t = 0; t < SEL_MAX_PREFIX; ++t) hit = hit || (t < pf.n && hi == pf.prefix[t]);
        if (hit) {
            const unsigned pos = atomicAdd(count, 1u);
            if (pos < cap) out[pos] = key;
        }
    }
}

static double key_to_double(unsigned long long key) {
    unsigned long long u = (key >> 63) ? (key & 0x7fffffffffffffffull) : ~key;
    double d;
    std::memcpy(&d, &u, sizeof(d));
    return d;
}

// The ranks[0..nr) smallest (0-based, ascending or not) non-NaN values of d_x[0..n): three passes over the data (six with
// heavy ties)
// (one shared top-digit pass, then five passes that follow all distinct prefixes together).  ranks may be filled in
// after the first pass: `plan(nv)` is called with the number of valid values and returns them.
template <typename Plan>
static int select_ranks(const double *d_x, int64_t n, Plan plan, std::vector<int64_t> &ranks, std::vector<double> &values,
                        int64_t *n_valid_out) {
    hipStream_t st = rt().stream;
    static unsigned int *d_hist = nullptr;
    static unsigned int *h_hist = nullptr;          // pinned
    if (!d_hist) {
        MLMC_HIP_CHECK(hipMalloc(&d_hist, sizeof(unsigned int) * SEL_MAX_PREFIX * SEL_BINS));
        MLMC_HIP_CHECK(hipHostMalloc((void **)&h_hist, sizeof(unsigned int) * SEL_MAX_PREFIX * SEL_BINS, hipHostMallocDefault));
    }
    int blocks = rt().n_cu * 4;
    const int64_t want = (n + 255) / 256;
    if (want < blocks) blocks = (int)(want > 0 ? want : 1);
    // digits from the top: bits [53, 64), [42, 53), [31, 42), [20, 31), [9, 20), then the last 9 bits
    const int shifts[6] = {53, 42, 31, 20, 9, 0};
    std::vector<unsigned long long> prefix;   // per rank
    std::vector<int64_t> k;                   // remaining rank inside the prefix
    for (int pass = 0; pass < 6; ++pass) {
        const int shift = shifts[pass];
        const int width = (pass == 5) ? 9 : SEL_BITS;
        const int nbins = 1 << width;
        // distinct prefixes of this pass, SEL_MAX_PREFIX at a time
        std::vector<unsigned long long> uniq;
        if (pass == 0) uniq.push_back(0);
        else
            for (unsigned long long p : prefix)
                if (std::find(uniq.begin(), uniq.end(), p) == uniq.end()) uniq.push_back(p);
        std::vector<std::vector<unsigned int>> hists(uniq.size());
        for (size_t g0 = 0; g0 < uniq.size(); g0 += SEL_MAX_PREFIX) {
            SelPrefixes pf;
            pf.n = (int)std::min<size_t>(SEL_MAX_PREFIX, uniq.size() - g0);
            for (int t = 0; t < SEL_MAX_PREFIX; ++t) pf.prefix[t] = t < pf.n ? uniq[g0 + t] : 0;
            MLMC_HIP_CHECK(hipMemsetAsync(d_hist, 0, sizeof(unsigned int) * pf.n * SEL_BINS, st));
            hipLaunchKernelGGL(k_select_hist, dim3(blocks), dim3(256), 0, st, d_x, n, pf, shift, width, pass == 0 ? 1 : 0, d_hist);
            MLMC_HIP_CHECK(hipGetLastError());
            MLMC_HIP_CHECK(hipMemcpyAsync(h_hist, d_hist, sizeof(unsigned int) * pf.n * SEL_BINS, hipMemcpyDeviceToHost, st));
            MLMC_HIP_CHECK(wait_stream(st));
            for (int t = 0; t < pf.n; ++t) hists[g0 + t].assign(h_hist + (size_t)t * SEL_BINS, h_hist + (size_t)t * SEL_BINS + nbins);
        }
        if (pass == 0) {
            int64_t nv = 0;
            for (unsigned int c : hists[0]) nv += c;
            if (n_valid_out) *n_valid_out = nv;
            ranks = plan(nv);
            for (int64_t r : ranks)
                if (r < 0 || r >= nv) return fail("order statistic index out of range");
            prefix.assign(ranks.size(), 0);
            k = ranks;
        }
        std::vector<int64_t> under(ranks.size(), 0);      // values that share the rank's prefix after this pass
        for (size_t r = 0; r < ranks.size(); ++r) {
            const size_t u = pass == 0 ? 0 : (size_t)(std::find(uniq.begin(), uniq.end(), prefix[r]) - uniq.begin());
            const std::vector<unsigned int> &h = hists[u];
            int64_t cum = 0;
            int digit = -1;
            for (int bin = 0; bin < nbins; ++bin) {
                if (k[r] < cum + (int64_t)h[bin]) { digit = bin; break; }
                cum += h[bin];
            }
            if (digit < 0) return fail("radix select: inconsistent histogram");
            k[r] -= cum;
            prefix[r] = (prefix[r] << width) | (unsigned long long)digit;
            under[r] = h[digit];
        }
        if (pass == 1) {   // 22 bits fixed: finish on the host when the candidates are few
            constexpr unsigned CAP = 1u << 16;
            static unsigned long long *d_cand = nullptr, *h_cand = nullptr;
            static unsigned int *d_ccount = nullptr, *h_ccount = nullptr;
            if (!d_cand) {
                MLMC_HIP_CHECK(hipMalloc(&d_cand, sizeof(unsigned long long) * CAP));
                MLMC_HIP_CHECK(hipMalloc(&d_ccount, sizeof(unsigned int)));
                MLMC_HIP_CHECK(hipHostMalloc((void **)&h_cand, sizeof(unsigned long long) * CAP, hipHostMallocDefault));
                MLMC_HIP_CHECK(hipHostMalloc((void **)&h_ccount, sizeof(unsigned int), hipHostMallocDefault));
            }
            std::vector<unsigned long long> up;            // distinct prefixes and the number of values under each
            int64_t total = 0;
            for (size_t r = 0; r < ranks.size(); ++r)
                if (std::find(up.begin(), up.end(), prefix[r]) == up.end()) { up.push_back(prefix[r]); total += under[r]; }
            if (total <= (int64_t)CAP) {
                std::vector<unsigned long long> cand;
                for (size_t g0 = 0; g0 < up.size(); g0 += SEL_MAX_PREFIX) {
                    SelPrefixes pf;
                    pf.n = (int)std::min<size_t>(SEL_MAX_PREFIX, up.size() - g0);
                    for (int t = 0; t < SEL_MAX_PREFIX; ++t) pf.prefix[t] = t < pf.n ? up[g0 + t] : 0;
                    MLMC_HIP_CHECK(hipMemsetAsync(d_ccount, 0, sizeof(unsigned int), st));
                    hipLaunchKernelGGL(k_select_collect, dim3(blocks), dim3(256), 0, st, d_x, n, pf, shift, d_cand, d_ccount, CAP);
                    MLMC_HIP_CHECK(hipGetLastError());
                    MLMC_HIP_CHECK(hipMemcpyAsync(h_ccount, d_ccount, sizeof(unsigned int), hipMemcpyDeviceToHost, st));
                    MLMC_HIP_CHECK(wait_stream(st));
                    const unsigned got = *h_ccount;
                    if (got > CAP) return fail("radix select: candidate buffer overflow");
                    if (got) {
                        MLMC_HIP_CHECK(hipMemcpyAsync(h_cand, d_cand, sizeof(unsigned long long) * got, hipMemcpyDeviceToHost, st));
                        MLMC_HIP_CHECK(wait_stream(st));
                        cand.insert(cand.end(), h_cand, h_cand + got);
                    }
                }
                for (size_t r = 0; r < ranks.size(); ++r) {
                    std::vector<unsigned long long> mine;
                    for (unsigned long long key : cand)
                        if ((key >> shift) == prefix[r]) mine.push_back(key);
                    if (k[r] < 0 || (size_t)k[r] >= mine.size()) return fail("radix select: inconsistent candidates");
                    std::nth_element(mine.begin(), mine.begin() + k[r], mine.end());
                    prefix[r] = mine[(size_t)k[r]];
                }
                break;                                      // prefix[r] now is the full key of rank r
            }
        }
    }
    values.resize(ranks.size());
    for (size_t r = 0; r < ranks.size(); ++r) values[r] = key_to_double(prefix[r]);
    return 0;
}

}  // namespace mlmc

using namespace mlmc;

extern "C" int mlmc_percentiles(const double *x, int64_t n, const double *q_percent, int32_t nq, double *out, int64_t *n_valid,
                                int mem_kind) {
    MLMC_API_GUARD;
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!x || !q_percent || !out || nq <= 0) return fail("mlmc_percentiles: bad argument");
    if (n <= 0) return fail("mlmc_percentiles: empty input");
    for (int i = 0; i < nq; ++i)
        if (!(q_percent[i] >= 0.0 && q_percent[i] <= 100.0)) return fail("mlmc_percentiles: percentiles must be in [0, 100]");
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
    // NumPy: q = percent / 100 ; virtual index = (n - 1) * q ; linear interpolation between the neighbours
    std::vector<double> gamma(nq);
    auto plan = [&](int64_t nv) {
        std::vector<int64_t> ranks;
        for (int i = 0; i < nq; ++i) {
            const double vi = (double)(nv - 1) * (q_percent[i] / 100.0);
            int64_t prev = (int64_t)std::floor(vi);
            double g = vi - (double)prev;
            if (prev < 0) { prev = 0; g = 0.0; }
            int64_t next = prev + 1;
            if (next > nv - 1) next = nv - 1;
            gamma[i] = g;
            ranks.push_back(prev);
            ranks.push_back(next < 0 ? 0 : next);
        }
        return ranks;
    };
    std::vector<int64_t> ranks;
    std::vector<double> values;
    int64_t nv = 0;
    int rc = select_ranks(d_x, n, plan, ranks, values, &nv);
    if (!rc && nv == 0) rc = fail("order statistic index out of range");
    for (int i = 0; i < nq && !rc; ++i) {
        const double a = values[2 * i], b = values[2 * i + 1];
        const double diff = b - a;
        double r = a + diff * gamma[i];                   // numpy.lib._function_base_impl._lerp
        if (gamma[i] >= 0.5) r = b - diff * (1.0 - gamma[i]);
        out[i] = r;
    }
    if (n_valid) *n_valid = nv;
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
    MLMC_API_GUARD;
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
