// Diagnostic (not part of the library): per-wave shader cycles of the phases of the covariance kernels (R <= 16, <= 32, 64; pair
// levels and level 0; with variances and mean-only): phase 1 (evaluation), wait at the barrier behind it, phase 2 (MFMA), wait at
// the barrier behind it -- averaged, per wave index, with the SIMD each wave ran on; wall time of the workgroups by wave slot,
// by rank on the CU (blockIdx / 256) and by XCD with its clock.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DMLMC_PROF_COV -I include -I mlmc_amd/csrc
//        tools/prof_cov.hip mlmc_amd/csrc/api.hip mlmc_amd/csrc/moments.hip mlmc_amd/csrc/maxent.hip mlmc_amd/csrc/select.hip
//        mlmc_amd/csrc/expr.hip mlmc_amd/csrc/expr_jit.hip mlmc_amd/csrc/synth.hip -ldl -o tools/prof_cov
#include "../mlmc_amd/csrc/cov.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

using namespace mlmc;

template <int T, int MODE, bool PAIR = true>
static void run(const char *name, const BasisParams &bp, const double *f, const double *c, int64_t n, int blocks) {
    constexpr int NT = 16 * T;
    double *partials; unsigned long long *prof;
    (void)hipMalloc(&partials, sizeof(double) * (size_t)blocks * 4 * 3 * NT * NT);
    (void)hipMalloc(&prof, sizeof(unsigned long long) * (size_t)blocks * 4 * 6);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_cov), &prof, sizeof(prof));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 60; ++it) {
        (void)hipEventRecord(e0);
        if constexpr (T == 4)
            hipLaunchKernelGGL((k_cov_accum_t4<MLMC_LEGENDRE, PAIR, MODE, 0>), dim3(blocks), dim3(256), 0, 0, bp, f, PAIR ? c : nullptr, nullptr, n, bp.size, partials, nullptr);
        else
            hipLaunchKernelGGL((k_cov_accum<MLMC_LEGENDRE, T, PAIR, MODE>), dim3(blocks), dim3(256), 0, 0, bp, f, PAIR ? c : nullptr, nullptr, n, bp.size, partials, nullptr, 0, 0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> p((size_t)blocks * 4 * 6);
    (void)hipMemcpy(p.data(), prof, sizeof(unsigned long long) * p.size(), hipMemcpyDeviceToHost);
    double sum[5] = {0, 0, 0, 0, 0};
    for (size_t w = 0; w < (size_t)blocks * 4; ++w)
        for (int k = 0; k < 5; ++k) sum[k] += (double)p[w * 6 + k];
    const double nw = (double)blocks * 4;
    const int64_t bsz = T == 4 ? (PAIR ? COV_T4_BATCH : 2 * COV_T4_BATCH) : cov_batch(T, false, false, PAIR);
    const double batches_per_wg = (double)((n + bsz - 1) / bsz) / blocks;
    printf("%s: kernel %.3f ms; per wave and batch (cycles): phase1 %.0f  barrier1 %.0f  phase2 %.0f  barrier2 %.0f  total %.0f  (%.1f batches per workgroup)\n",
           name, ms, sum[0] / nw / batches_per_wg, sum[1] / nw / batches_per_wg, sum[2] / nw / batches_per_wg,
           sum[3] / nw / batches_per_wg, sum[4] / nw / batches_per_wg, batches_per_wg);
    {   // spread of the waves' total cycles: workgroups that finish early leave their CU partner running alone
        unsigned long long lo = ~0ull, hi = 0;
        for (size_t w = 0; w < (size_t)blocks * 4; ++w) { lo = std::min(lo, p[w * 6 + 4]); hi = std::max(hi, p[w * 6 + 4]); }
        printf("    cycles per wave: min %.3f M  max %.3f M  -> shader clock >= %.2f GHz over the kernel's %.3f ms\n", lo * 1e-6, hi * 1e-6,
               hi * 1e-6 / ms, ms);
        {   // by wave slot (HW_ID.wave_id): how many waves, their mean wall time
            double sum[16] = {0}; int cnt[16] = {0};
            for (size_t w = 0; w < (size_t)blocks * 4; ++w) { const int id = (int)(p[w * 6 + 5] & 15); sum[id] += (double)(p[w * 6 + 5] >> 24) * 0.01; ++cnt[id]; }
            printf("      wave slot: waves, mean wall us:");
            for (int i = 0; i < 16; ++i) if (cnt[i]) printf("  [%d] %d %.0f", i, cnt[i], sum[i] / cnt[i]);
            printf("\n");
        }
        {   // by rank on the CU (blockIdx / 256 for a full grid): mean and max wall time of the workgroups
            const int per = blocks / 256 > 0 ? blocks / 256 : 1;
            printf("      rank on the CU: mean / max wall us:");
            for (int r = 0; r < per; ++r) {
                double sum = 0, mx = 0; int cnt = 0;
                for (int b2 = r * 256; b2 < (r + 1) * 256 && b2 < blocks; ++b2)
                    for (int w = 0; w < 4; ++w) { const double us = (double)(p[((size_t)b2 * 4 + w) * 6 + 5] >> 24) * 0.01; sum += us; mx = std::max(mx, us); ++cnt; }
                if (cnt) printf("  [%d] %.0f / %.0f", r, sum / cnt, mx);
            }
            printf("\n");
        }
        {   // do the workgroups that share a CU have distinct blockIdx / (grid / workgroups per CU)?  (HW_ID: cu 11:8, sh 12, se 15:13)
            std::vector<std::vector<int>> on_cu(8 * 256);
            for (int b2 = 0; b2 < blocks; ++b2) {
                const unsigned long long h = p[((size_t)b2 * 4) * 6 + 5];
                const int cu = (int)((h >> 8) & 0xff), xcc = (int)((h >> 16) & 15);
                on_cu[xcc * 256 + cu].push_back(b2);
            }
            int groups = 0, distinct = 0; const int per = blocks / 256 > 0 ? blocks / 256 : 1;
            for (auto &v : on_cu) {
                if (v.empty()) continue;
                ++groups;
                unsigned seen = 0; bool ok = (int)v.size() == per;
                for (int b2 : v) { const unsigned bit = 1u << ((b2 / 256) % 8); if (seen & bit) ok = false; seen |= bit; }
                distinct += ok;
            }
            printf("      CUs with workgroups: %d, of them with %d workgroups of distinct blockIdx / 256: %d; first CU:", groups, per, distinct);
            for (auto &v : on_cu) if (!v.empty()) { for (int b2 : v) printf(" %d", b2); break; }
            printf("\n");
        }
        // per XCD: wall time of its waves (100 MHz ticks) and their clock
        for (int x = 0; x < 8; ++x) {
            double tl = 1e30, th = 0, cl = 1e30, ch = 0; int cnt = 0;
            for (size_t w = 0; w < (size_t)blocks * 4; ++w) {
                if ((int)((p[w * 6 + 5] >> 16) & 15) != x) continue;
                const double us = (double)(p[w * 6 + 5] >> 24) * 0.01, ghz = (double)p[w * 6 + 4] / (us * 1e3);
                tl = std::min(tl, us); th = std::max(th, us); cl = std::min(cl, ghz); ch = std::max(ch, ghz); ++cnt;
            }
            if (cnt) printf("      XCD %d: %4d waves, wall %.0f .. %.0f us, clock %.3f .. %.3f GHz\n", x, cnt, tl, th, cl, ch);
        }
    }
    for (int W = 0; W < 4; ++W) {   // per wave index: the same phases, and the SIMD the wave ran on
        double sw[5] = {0, 0, 0, 0, 0};
        int simd[4] = {0, 0, 0, 0};
        for (int b = 0; b < blocks; ++b) {
            for (int k = 0; k < 5; ++k) sw[k] += (double)p[((size_t)b * 4 + W) * 6 + k];
            ++simd[(p[((size_t)b * 4 + W) * 6 + 5] >> 4) & 3];
        }
        printf("    wave %d: phase1 %.0f  barrier1 %.0f  phase2 %.0f  barrier2 %.0f   SIMD histogram %d %d %d %d\n", W,
               sw[0] / blocks / batches_per_wg, sw[1] / blocks / batches_per_wg, sw[2] / blocks / batches_per_wg,
               sw[3] / blocks / batches_per_wg, simd[0], simd[1], simd[2], simd[3]);
    }
    (void)hipFree(partials);
    (void)hipFree(prof);
}

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    std::vector<double> h(n), hc(n);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    for (int64_t i = 0; i < n; ++i) { h[i] = nd(g); hc[i] = h[i] + 0.05 * nd(g); }
    double *f, *c;
    (void)hipMalloc(&f, sizeof(double) * n); (void)hipMalloc(&c, sizeof(double) * n);
    (void)hipMemcpy(f, h.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    (void)hipMemcpy(c, hc.data(), sizeof(double) * n, hipMemcpyHostToDevice);
    BasisParams bp{};
    bp.kind = MLMC_LEGENDRE; bp.size = 32; bp.shift = -3.7190164854556804; bp.scale = 2.0 / (2 * 3.7190164854556804);
    bp.ref0 = -1; bp.ref1 = 1; bp.is_log = 0; bp.is_clip = 1;
    run<2, 0>("T=2 pair, mean+var", bp, f, c, n, 512);
    run<2, 0>("T=2 pair, mean+var", bp, f, c, n, 512);
    run<2, 2>("T=2 pair, mean only", bp, f, c, n, 512);
    run<2, 0, false>("T=2 level 0, mean+var", bp, f, c, n, 512);
    run<2, 2, false>("T=2 level 0, mean only", bp, f, c, n, 512);
    bp.size = 64;
    run<4, 0>("T=4 pair, mean+var", bp, f, c, n, 256 * COV_T4_WGS);
    run<4, 3>("T=4 pair, variance only", bp, f, c, n, 256 * COV_T4_WGS);
    run<4, 3, false>("T=4 level 0, variance only", bp, f, c, n, 256 * COV_T4_WGS);
    run<4, 2>("T=4 pair, mean only", bp, f, c, n, 256 * COV_T4_WGS);
    run<4, 0, false>("T=4 level 0, mean+var", bp, f, c, n, 256 * COV_T4_WGS);
    run<4, 2, false>("T=4 level 0, mean only", bp, f, c, n, 256 * COV_T4_WGS);
    bp.size = 16;
    run<1, 0>("T=1 pair, mean+var (4 workgroups per CU)", bp, f, c, n, 1024);
    run<1, 2>("T=1 pair, mean only (4 workgroups per CU)", bp, f, c, n, 1024);
    return 0;
}
