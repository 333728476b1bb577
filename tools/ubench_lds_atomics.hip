// Micro-benchmark: LDS atomic add throughput on gfx950 -- ds_add_f64 vs ds_add_u64 vs ds_add_u32 / f32, 64 lanes adding to
// random entries of a per-wave table (the access pattern of the sparse spline-moment accumulation).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_lds_atomics.hip -o tools/ubench_lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ __launch_bounds__(256) void k(T *out, int iters, unsigned seed, int spread) {
    __shared__ T tab[4][136];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 136; i += 256) (&tab[0][0])[i] = T(0);
    __syncthreads();
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    T v = T(1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            const int idx = (s >> 16) % spread;
            atomicAdd(&tab[wave][idx], v);
        }
    }
    __syncthreads();
    if (threadIdx.x < 128) out[blockIdx.x * 128 + threadIdx.x] = tab[0][threadIdx.x] + tab[1][threadIdx.x] + tab[2][threadIdx.x] + tab[3][threadIdx.x];
}

template <typename T>
void run(const char *name, int blocks_per_cu, int spread) {
    const int ncu = 256, iters = 2000, blocks = ncu * blocks_per_cu;
    T *d;
    (void)hipMalloc(&d, sizeof(T) * blocks * 128);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<T><<<blocks, 256>>>(d, 10, 1u, spread);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<T><<<blocks, 256>>>(d, iters, 7u, spread);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 256 * iters * 8;
    printf("%-4s spread %3d  %d workgroups/CU: %.3f ms  %.2f lane-atomics per cycle (2.4 GHz) per CU\n", name, spread, blocks_per_cu, ms,
           lane_ops / ncu / (ms * 1e-3 * 2.4e9));
    (void)hipFree(d);
}

int main() {
    for (int w : {1, 2, 4})
        for (int spread : {128, 16}) {
            run<double>("f64", w, spread);
            run<unsigned long long>("u64", w, spread);
            run<unsigned int>("u32", w, spread);
            run<float>("f32", w, spread);
        }
    return 0;
}
