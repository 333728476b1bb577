// Micro-benchmark: sustained fp64 VALU issue rate on gfx950 (fma / mul / add mixes, N independent chains,
// waves per SIMD swept through the grid size).  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CH, int MODE>
__global__ __launch_bounds__(256) void k(double *out, double a, double b, int iters) {
    double x[CH], y[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) { x[c] = a + c + threadIdx.x; y[c] = b + c; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (MODE == 0) x[c] = __builtin_fma(x[c], a, b);                 // pure fma chain
                if (MODE == 1) { y[c] = y[c] * a; x[c] = __builtin_fma(x[c], b, y[c]); }   // mul + fma
                if (MODE == 2) { double t = a * y[c]; t = __builtin_fma(b, x[c], -t); y[c] = x[c]; x[c] = t; }  // recurrence
                if (MODE == 3) { x[c] = x[c] + a; y[c] = y[c] * b; }             // add + mul
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += x[c] + y[c];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH, int MODE>
void run(const char *name, int blocks_per_cu, int instr_per_inner) {
    int ncu = 256, iters = 2000;
    int blocks = ncu * blocks_per_cu;
    double *d;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH, MODE><<<blocks, 256>>>(d, 1.0000001, 0.5, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CH, MODE><<<blocks, 256>>>(d, 1.0000001, 0.5, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 8 * CH * instr_per_inner;   // wave-instructions
    double per_simd_cycles = ms * 1e-3 * 2.4e9;                                // cycles at 2.4 GHz
    double cyc_per_instr = per_simd_cycles / (winstr / (ncu * 4));
    printf("%-28s CH=%d waves/SIMD=%d  %.3f ms  %.2f cycles(2.4GHz)/wave-instr/SIMD  %.1f Ginstr-lanes/s\n", name, CH, blocks_per_cu, ms,
           cyc_per_instr, winstr * 64 / (ms * 1e-3) / 1e9);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<1, 0>("fma chain", w, 1);
        run<2, 0>("fma chain", w, 1);
        run<4, 0>("fma chain", w, 1);
        run<8, 0>("fma chain", w, 1);
        run<4, 1>("mul+fma", w, 2);
        run<4, 2>("recurrence(mul,fma)", w, 2);
        run<8, 2>("recurrence(mul,fma)", w, 2);
        run<8, 3>("add+mul", w, 2);
    }
    return 0;
}
