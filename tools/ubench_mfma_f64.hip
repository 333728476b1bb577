// Micro-benchmark: v_mfma_f64_16x16x4_f64 issue rate on gfx950, alone and with fp64 VALU work / LDS stores / LDS loads
// placed between the matrix instructions of the SAME wave, at one and two waves per SIMD.  Answers: does vector fp64 work
// of the covariance kernel's evaluation phase fit into the shadow of its own matrix instructions?
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_mfma_f64.hip -o tools/ubench_mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4f64 __attribute__((ext_vector_type(4)));

// NACC independent accumulators, each receiving RUN consecutive MFMAs before the next one takes over; per MFMA: NV
// dependent-chain fp64 VALU pairs (mul + fma of a recurrence), NW LDS stores, NR LDS loads (consumed one group later)
template <int NACC, int NV, int NW, int NR, int RUN = 1>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters, unsigned long long *clk) {
    __shared__ double lds[64 * 66];
    v4f64 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double p1 = a + threadIdx.x * 1e-9, p2 = b, x = 0.999;
    double q1 = b + threadIdx.x * 1e-9, q2 = a;
    double va = a, vb = b, sink = 0.0;
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = a;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NACC * RUN; ++u) {
            // inline asm pins the accumulators to VGPRs: with the builtin hipcc parks loop-carried accumulators in AGPRs and
            // copies all of them in and out every iteration (176 v_accvgpr moves at 11 accumulators), which swamps the figure
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[u / RUN]) : "v"(va), "v"(vb));
#pragma unroll
            for (int v = 0; v < NV; ++v) {   // two independent recurrences, alternating
                if (v & 1) { const double t = __builtin_fma(x, q1, -(0.25 * q2)); q2 = q1; q1 = t; }
                else { const double t = __builtin_fma(x, p1, -(0.25 * p2)); p2 = p1; p1 = t; }
            }
#pragma unroll
            for (int w = 0; w < NW; ++w) lds[((u * 4 + w) % 64) * 66 + lane] = p1;
#pragma unroll
            for (int r = 0; r < NR; ++r) sink += lds[((u * 4 + r) % 64) * 66 + ((lane + it) & 63)];
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = p1 + q1 + va + sink;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC, int NV, int NW, int NR, int RUN = 1>
void run(int blocks_per_cu) {
    const int ncu = 256, iters = 400;
    const int blocks = ncu * blocks_per_cu;
    double *d;
    unsigned long long *clk, h[2];
    hipMalloc(&d, sizeof(double) * blocks * 256);
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) k<NACC, NV, NW, NR, RUN><<<blocks, 256>>>(d, 1.0000001, 0.5, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, NV, NW, NR, RUN><<<blocks, 256>>>(d, 1.0000001, 0.5, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk + 2 * (blocks / 2), sizeof(h), hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / (double)h[1] * 0.1;          // shader cycles per 100 MHz tick
    const double per_it = NACC * RUN;
    const double mfma_per_simd = (double)blocks_per_cu * iters * per_it;   // every wave of a SIMD issues iters * per_it
    const double cyc = (double)h[0] / (iters * per_it) / blocks_per_cu;   // shader cycles per MFMA of the SIMD (in-kernel clock)
    printf("acc=%d run=%d valu-pairs=%d lds-st=%d lds-ld=%d waves/SIMD=%d: %.3f ms  clock %.2f GHz  %.1f cycles per MFMA per SIMD  %.1f TFLOP/s\n",
           NACC, RUN, NV, NW, NR, blocks_per_cu, ms, ghz, cyc, mfma_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12);
    hipFree(d);
    hipFree(clk);
}

int main() {
    for (int w : {1, 2}) {
        run<4, 0, 0, 0, 4>(w);
        run<1, 0, 0, 0, 16>(w);
        run<11, 0, 0, 0, 1>(w);
        run<11, 0, 0, 0, 2>(w);
        run<11, 0, 0, 0, 4>(w);
        run<11, 0, 0, 0, 8>(w);
        run<11, 0, 0, 1, 4>(w);
        run<11, 0, 0, 2, 4>(w);
        run<4, 1, 0, 0, 4>(w);
        run<4, 2, 0, 0, 4>(w);
        run<4, 4, 0, 0, 4>(w);
        run<4, 0, 2, 0, 4>(w);
        run<4, 0, 0, 1, 4>(w);
        run<4, 0, 0, 2, 4>(w);
    }
    return 0;
}
