// Diagnostic (not part of the library): does a light fp64 vector kernel on a SECOND stream fill the issue slots the 64-term
// variance-only covariance kernel leaves idle?  The covariance kernel holds 4 workgroups per CU (148 KB of LDS, 64 VGPRs x 4 waves
// per SIMD); the vector kernel takes 256 threads per CU, no LDS, <= 160 VGPRs.  Times: covariance alone, vector alone, both at once.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I include -I mlmc_amd/csrc tools/ubench_overlap.hip
//        mlmc_amd/csrc/api.hip mlmc_amd/csrc/moments.hip mlmc_amd/csrc/maxent.hip mlmc_amd/csrc/select.hip mlmc_amd/csrc/expr.hip
//        mlmc_amd/csrc/expr_jit.hip mlmc_amd/csrc/synth.hip -ldl -o tools/ubench_overlap
#include "../mlmc_amd/csrc/cov.hip"
#include <cstdio>
#include <random>
#include <vector>

using namespace mlmc;

// 4 independent recurrence chains per lane (like the mean-only moments kernel), `terms` steps per sample pair, 64 accumulators
__global__ __launch_bounds__(256, 1) void k_vec(const double *__restrict__ f, const double *__restrict__ c, int64_t n, int terms,
                                                double *__restrict__ out, int prio) {
    (void)prio; __builtin_amdgcn_s_setprio(0);
    double acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.0;
    const int64_t T = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += 2 * T) {
        const int64_t j = i + T < n ? i + T : i;
        double xf0 = f[i], xc0 = c[i], xf1 = f[j], xc1 = c[j];
        double a0 = 1, b0 = 0, a1 = 1, b1 = 0, a2 = 1, b2 = 0, a3 = 1, b3 = 0;
        for (int p = 0; p < terms; p += 64) {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                const double g = 0.2499 + 1e-4 * k;
                double q0 = __builtin_fma(xf0, a0, -(g * b0)); b0 = a0; a0 = q0;
                double q1 = __builtin_fma(xc0, a1, -(g * b1)); b1 = a1; a1 = q1;
                double q2 = __builtin_fma(xf1, a2, -(g * b2)); b2 = a2; a2 = q2;
                double q3 = __builtin_fma(xc1, a3, -(g * b3)); b3 = a3; a3 = q3;
                acc[k] += (q0 - q1);
                acc[k] += (q2 - q3);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc[i];
    out[(int64_t)blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char **argv) {
    const int64_t n = 10000000;
    const int vec_blocks = argc > 1 ? atoi(argv[1]) : 256;
    std::vector<double> h(n), hc(n);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    for (int64_t i = 0; i < n; ++i) { h[i] = 0.3 * nd(g); hc[i] = h[i] + 0.01 * nd(g); }
    double *f, *c, *partials, *out;
    (void)hipMalloc(&f, sizeof(double) * n); (void)hipMalloc(&c, sizeof(double) * n);
    (void)hipMemcpy(f, h.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(c, hc.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    const int blocks = 256 * COV_T4_WGS;
    (void)hipMalloc(&partials, sizeof(double) * (size_t)blocks * 3 * 64 * 64);
    (void)hipMalloc(&out, sizeof(double) * 256 * 1024);
    BasisParams bp{};
    bp.kind = MLMC_LEGENDRE; bp.size = 64; bp.shift = -3.7190164854556804; bp.scale = 2.0 / (2 * 3.7190164854556804);
    bp.ref0 = -1; bp.ref1 = 1; bp.is_log = 0; bp.is_clip = 1;
    hipStream_t sa, sb;
    (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, e2, e3;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2); (void)hipEventCreate(&e3);
    auto cov = [&]() {
        hipLaunchKernelGGL((k_cov_accum_t4<MLMC_LEGENDRE, true, 3, 0>), dim3(blocks), dim3(256), 0, sa, bp, f, c, nullptr, n, 64, partials, nullptr);
    };
    for (int terms : {128, 256}) {
        for (int prio : {0}) {
            auto vec = [&]() { hipLaunchKernelGGL(k_vec, dim3(vec_blocks), dim3(256), 0, sb, f, c, n / 4, terms, out, prio); };
            float t_cov = 0, t_vec = 0, t_both = 0, t_cov_in_both = 0;
            for (int it = 0; it < 6; ++it) {
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, sa); for (int k = 0; k < 4; ++k) cov(); (void)hipEventRecord(e1, sa); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&t_cov, e0, e1);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, sb); for (int k = 0; k < 4; ++k) vec(); (void)hipEventRecord(e1, sb); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&t_vec, e0, e1);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, sa); (void)hipEventRecord(e2, sb);
                for (int k = 0; k < 4; ++k) { cov(); vec(); }
                (void)hipEventRecord(e1, sa); (void)hipEventRecord(e3, sb);
                (void)hipEventSynchronize(e1); (void)hipEventSynchronize(e3);
                (void)hipEventElapsedTime(&t_cov_in_both, e0, e1);
                float t_vec_in_both; (void)hipEventElapsedTime(&t_vec_in_both, e2, e3);
                t_both = t_cov_in_both > t_vec_in_both ? t_cov_in_both : t_vec_in_both;
            }
            printf("vector kernel: %d blocks, %d terms, prio %d: 4 x cov alone %.3f ms, 4 x vec alone %.3f ms, both at once %.3f ms (cov stream %.3f)  sum %.3f\n",
                   vec_blocks, terms, prio, t_cov, t_vec, t_both, t_cov_in_both, t_cov + t_vec);
        }
    }
    return 0;
}
