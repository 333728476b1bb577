// Diagnostic: sustained shader clock under an fp64 VALU load (delta s_memtime / delta s_memrealtime x 100 MHz)
// and cycles per fp64 wave-instruction, as a function of how long the load has been running.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void k(double *out, unsigned long long *stamps, double a, double b, int iters) {
    double x[8], y[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { x[c] = a + c + threadIdx.x; y[c] = b + c; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < 8; ++c) { double t = a * y[c]; t = __builtin_fma(b, x[c], -t); y[c] = x[c]; x[c] = t; }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += x[c] + y[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    int blocks = 512;
    double *d; unsigned long long *st;
    (void)hipMalloc(&d, sizeof(double) * blocks * 256);
    (void)hipMalloc(&st, sizeof(unsigned long long) * 2 * blocks);
    std::vector<unsigned long long> h(2 * blocks);
    for (int iters : {200, 2000, 20000, 200000, 200000, 200000}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        k<<<blocks, 256>>>(d, st, 1.0000001, 0.5, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(h.data(), st, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        std::vector<double> clk, cyc;
        for (int b = 0; b < blocks; ++b) { clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 100.0); cyc.push_back((double)h[2*b]); }
        std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
        double winstr_per_simd = 2.0 * iters * 8 * 8 * 2;   // 2 waves/SIMD x iters x 8 x 8 chains x 2 instr
        printf("iters %7d  kernel %.3f ms  clock median %.0f MHz (min %.0f max %.0f)  cycles/wave-instr/SIMD %.2f  TFLOPs-equivalent(all-fma) %.1f\n", iters, ms,
               clk[blocks / 2], clk.front(), clk.back(), cyc[blocks / 2] / winstr_per_simd,
               (double)blocks * 256 * iters * 128 * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
