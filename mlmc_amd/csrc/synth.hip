// On-device SynthSimulation samples, bit-compatible with the reference's sampling chain:
//   sample id "L{level:02d}_S{index:07d}"                                   mlmc/sampler.py:120
//   seed = first little-endian uint32 of md5(id)                            mlmc/sampling_pool.py:75-84
//   y = scipy.stats.norm().rvs(2) from numpy RandomState(seed)              mlmc/sim/synth_simulation.py:75-90
//       = MT19937 seeded by init_genrand(seed), two legacy polar Box-Muller normals (numpy/random/src/legacy:
//         legacy_gauss, mt19937_next_double); the first call returns f * x2, the second the cached f * x1
//   result = y + h sqrt(1e-4 + |y|), rows [quantity 2][time 3][location 2][component 2], "+ location index"
//       on levels with a coarse simulation                                  synth_simulation.py:37-46,94-131
// One thread per sample.  MT19937 is never materialised: the first twist needs key[k], key[k+1] and key[k+397] of the
// freshly seeded state, and the seeding recurrence key[p+1] = 1812433253 (key[p] ^ key[p] >> 30) + p + 1 can be run
// as two cursors (one from 0, one from 397) in step with the outputs -- O(1) registers instead of a 2.5 KB state.
// The only step that is not exactly specified by IEEE-754 is log(): the result is identical to NumPy's whenever both
// libm's round the same way (measured in tests/test_gpu_parity.py::test_synth_generation).
#include "common.hpp"

namespace mlmc {

__device__ __constant__ uint32_t kMd5K[64] = {
    0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af,
    0xffff5bb1, 0x895cd7be, 0x6b901122, 0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa,
    0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6, 0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8,
    0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60, 0xbebfbc70,
    0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97,
    0xab9423a7, 0xfc93a039, 0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1,
    0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391};
__device__ __constant__ uint8_t kMd5S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                                             14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                                             4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};

// first 32-bit word of md5("L%02d_S%07d" % (level, index)); the id is shorter than 56 bytes: one block
__device__ uint32_t sample_seed(int level, uint64_t index) {
    uint8_t msg[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) msg[i] = 0;
    int len = 0;
    msg[len++] = 'L';
    if (level >= 100) msg[len++] = (uint8_t)('0' + (level / 100) % 10);
    msg[len++] = (uint8_t)('0' + (level / 10) % 10);
    msg[len++] = (uint8_t)('0' + level % 10);
    msg[len++] = '_';
    msg[len++] = 'S';
    int digits = 7;
    for (uint64_t t = index / 10000000ull; t > 0; t /= 10) ++digits;
    for (int d = digits - 1; d >= 0; --d) {
        msg[len + d] = (uint8_t)('0' + index % 10);
        index /= 10;
    }
    len += digits;
    msg[len] = 0x80;
    const uint32_t bits = (uint32_t)len * 8u;
    msg[56] = (uint8_t)bits;
    msg[57] = (uint8_t)(bits >> 8);
    uint32_t M[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
        M[i] = (uint32_t)msg[4 * i] | ((uint32_t)msg[4 * i + 1] << 8) | ((uint32_t)msg[4 * i + 2] << 16) | ((uint32_t)msg[4 * i + 3] << 24);
    uint32_t a = 0x67452301u, b = 0xefcdab89u, c = 0x98badcfeu, d = 0x10325476u;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        uint32_t f;
        int g;
        if (i < 16) { f = (b & c) | (~b & d); g = i; }
        else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
        else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
        else { f = c ^ (b | ~d); g = (7 * i) & 15; }
        f = f + a + kMd5K[i] + M[g];
        a = d;
        d = c;
        c = b;
        const int s = kMd5S[i];
        b = b + ((f << s) | (f >> (32 - s)));
    }
    return 0x67452301u + a;
}

struct Mt19937Head {   // the first <= 227 outputs of MT19937 after init_genrand(seed), generated in order
    uint32_t lo, hi;   // key[k], key[k + 397] of the seeded state
    int k;
    __device__ __forceinline__ static uint32_t step(uint32_t s, uint32_t pos) { return 1812433253u * (s ^ (s >> 30)) + pos + 1u; }
    __device__ void seed(uint32_t sd) {
        lo = sd;
        uint32_t s = sd;
        for (uint32_t pos = 0; pos < 397; ++pos) s = step(s, pos);
        hi = s;
        k = 0;
    }
    __device__ uint32_t next() {
        const uint32_t lo1 = step(lo, (uint32_t)k);
        const uint32_t y = (lo & 0x80000000u) | (lo1 & 0x7fffffffu);
        uint32_t v = hi ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        hi = step(hi, (uint32_t)(397 + k));
        lo = lo1;
        ++k;
        v ^= v >> 11;
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= v >> 18;
        return v;
    }
    __device__ double next_double() {   // mt19937_next_double
        const int32_t a = (int32_t)(next() >> 5), b = (int32_t)(next() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

constexpr int SYNTH_MAX_ROWS = 24;
struct SynthRows {
    double *out[SYNTH_MAX_ROWS];
    int row[SYNTH_MAX_ROWS];
};

__global__ __launch_bounds__(256) void k_synth_seeds(int level, int64_t first, int64_t n, uint32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sample_seed(level, (uint64_t)(first + i));
}

__global__ __launch_bounds__(256) void k_synth(int level, int64_t first, int64_t n, double h_fine, double h_coarse, double loc,
                                               double scale, int n_rows, SynthRows rows, int *__restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Mt19937Head mt;
    mt.seed(sample_seed(level, (uint64_t)(first + i)));
    double x1, x2, r2;
    int it = 0;
    do {   // legacy_gauss: polar Box-Muller, 4 generator outputs per trial, acceptance pi / 4
        x1 = 2.0 * mt.next_double() - 1.0;
        x2 = 2.0 * mt.next_double() - 1.0;
        r2 = x1 * x1 + x2 * x2;
        ++it;
    } while ((r2 >= 1.0 || r2 == 0.0) && it < 56);      // 56 trials = 224 outputs < 227 (probability 1e-38 of leaving this way)
    if (r2 >= 1.0 || r2 == 0.0) atomicAdd(overflow, 1);
    const double f = sqrt(-2.0 * log(r2) / r2);
    double y[2] = {f * x2, f * x1};
#pragma unroll
    for (int c = 0; c < 2; ++c) y[c] = y[c] * scale + loc;   // scipy rv_continuous.rvs: vals * scale + loc (two roundings)
    const bool has_coarse = h_coarse != 0.0;
    double fine[2], coarse[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        fine[c] = y[c] + h_fine * sqrt(1e-4 + fabs(y[c]));
        coarse[c] = has_coarse ? y[c] + h_coarse * sqrt(1e-4 + fabs(y[c])) : 0.0;
    }
    for (int r = 0; r < n_rows; ++r) {
        const int row = rows.row[r];
        const int comp = row & 1, place = (row >> 1) & 1;
        if (has_coarse) {
            double2 v = make_double2(fine[comp] + (double)place, coarse[comp] + (double)place);
            reinterpret_cast<double2 *>(rows.out[r])[i] = v;
        } else {
            rows.out[r][i] = fine[comp];
        }
    }
}

}  // namespace mlmc

extern "C" int mlmc_synth_seeds(int32_t level_id, int64_t first_sample, int64_t n, uint32_t *seeds_host) {
    MLMC_API_GUARD;
    using namespace mlmc;
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!seeds_host || level_id < 0 || level_id > 999 || first_sample < 0 || n < 0) return fail("mlmc_synth_seeds: bad argument");
    if (n == 0) return 0;
    uint32_t *d = nullptr;
    MLMC_HIP_CHECK(hipMalloc(&d, sizeof(uint32_t) * (size_t)n));
    hipLaunchKernelGGL(k_synth_seeds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, rt().stream, (int)level_id, first_sample, n, d);
    hipError_t e = hipMemcpyAsync(seeds_host, d, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, rt().stream);
    if (e == hipSuccess) e = wait_stream(rt().stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(std::string("mlmc_synth_seeds: ") + hipGetErrorString(e));
    return 0;
}

extern "C" int mlmc_synth_generate(int32_t level_id, int64_t first_sample, int64_t n, double fine_step, double coarse_step,
                                   double loc, double scale, int32_t n_rows, const int32_t *rows, double *const *out) {
    MLMC_API_GUARD;
    using namespace mlmc;
    if (!rt().ready) return fail("mlmc_init has not been called (no HIP device bound)");
    if (!rows || !out) return fail("mlmc_synth_generate: null argument");
    if (level_id < 0 || level_id > 999 || first_sample < 0 || n < 0) return fail("mlmc_synth_generate: level / sample range out of range");
    if (n_rows < 1 || n_rows > SYNTH_MAX_ROWS) return fail("mlmc_synth_generate: 1..24 rows");
    if (n == 0) return 0;
    SynthRows tab;
    for (int r = 0; r < n_rows; ++r) {
        if (rows[r] < 0 || rows[r] >= SYNTH_MAX_ROWS || !out[r]) return fail("mlmc_synth_generate: bad row");
        tab.row[r] = rows[r];
        tab.out[r] = out[r];
    }
    static int *d_overflow = nullptr;
    if (!d_overflow) {
        MLMC_HIP_CHECK(hipMalloc(&d_overflow, sizeof(int)));
        MLMC_HIP_CHECK(hipMemset(d_overflow, 0, sizeof(int)));
    }
    hipLaunchKernelGGL(k_synth, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, rt().stream, (int)level_id, first_sample, n, fine_step,
                       coarse_step, loc, scale, (int)n_rows, tab, d_overflow);
    MLMC_HIP_CHECK(hipGetLastError());
    return 0;
}
