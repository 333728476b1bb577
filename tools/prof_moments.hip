// Diagnostic (not part of the library): per-wave time stamps of k_moments_accum<LEGENDRE, 32> on the configs[1] shape.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DMLMC_PROF -I include -I mlmc_amd/csrc
//        tools/prof_moments.hip mlmc_amd/csrc/api.hip mlmc_amd/csrc/cov.hip mlmc_amd/csrc/maxent.hip mlmc_amd/csrc/select.hip
#include "../mlmc_amd/csrc/moments.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

using namespace mlmc;

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 10000000;
    const int nb0 = argc > 2 ? atoi(argv[2]) : 113, nb1 = argc > 3 ? atoi(argv[3]) : 199;
    std::vector<double> h(n);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    for (auto &v : h) v = nd(g);
    double *d[5];
    for (int i = 0; i < 5; ++i) { (void)hipMalloc(&d[i], sizeof(double) * n); (void)hipMemcpy(d[i], h.data(), sizeof(double) * n, hipMemcpyHostToDevice); }
    BasisParams bp{};
    bp.kind = MLMC_LEGENDRE; bp.size = 32; bp.shift = -3.7190164854556804; bp.scale = 2.0 / (2 * 3.7190164854556804);
    bp.ref0 = -1; bp.ref1 = 1; bp.is_log = 0; bp.is_clip = 1;
    SegTable tab{};
    tab.nseg = 3;
    tab.prio_mod = 2;
    tab.seg[0] = Seg{d[0], nullptr, nullptr, n, 0, nb0};
    tab.seg[1] = Seg{d[1], d[2], nullptr, n, nb0, nb1};
    tab.seg[2] = Seg{d[3], d[4], nullptr, n, nb0 + nb1, nb1};
    const int total = nb0 + 2 * nb1;
    double *partials; int64_t *pc; unsigned long long *prof;
    (void)hipMalloc(&partials, sizeof(double) * total * 64);
    (void)hipMalloc(&pc, sizeof(int64_t) * total * 2);
    (void)hipMalloc(&prof, sizeof(unsigned long long) * total * 4 * 5);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), &prof, sizeof(prof));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 300; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_moments_accum<MLMC_LEGENDRE, 32, 0, true>), dim3(total), dim3(ACC_THREADS), 0, 0, bp, tab, 0, partials, pc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it % 50 == 49) printf("iter %d  kernel %.1f us\n", it, ms * 1e3);
    }
    std::vector<unsigned long long> p((size_t)total * 4 * 5);
    (void)hipMemcpy(p.data(), prof, sizeof(unsigned long long) * p.size(), hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, tend = 0;
    for (size_t w = 0; w < (size_t)total * 4; ++w) { t0 = std::min(t0, p[w * 5]); tend = std::max(tend, p[w * 5 + 2]); }
    printf("last launch: event %.1f us, first wave start -> last wave end %.1f us\n", ms * 1e3, (tend - t0) / 100.0);
    for (int sgi = 0; sgi < 3; ++sgi) {
        const int b0 = tab.seg[sgi].block0, nb = tab.seg[sgi].nblocks;
        std::vector<double> st, le, en, loop_us, mhz, cyc;
        for (int b = b0; b < b0 + nb; ++b)
            for (int w = 0; w < 4; ++w) {
                const unsigned long long *q = &p[((size_t)b * 4 + w) * 5];
                st.push_back((q[0] - t0) / 100.0); le.push_back((q[1] - t0) / 100.0); en.push_back((q[2] - t0) / 100.0);
                loop_us.push_back((q[1] - q[0]) / 100.0); cyc.push_back((double)q[3]);
                mhz.push_back((double)q[3] / ((q[1] - q[0]) / 100.0));
            }
        auto stat = [](std::vector<double> v, const char *name) {
            std::sort(v.begin(), v.end());
            printf("   %-12s min %9.1f  p10 %9.1f  med %9.1f  p90 %9.1f  max %9.1f\n", name, v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
        };
        const double samples_per_thread = (double)n / (nb * 256.0);
        printf("segment %d (%s, %d blocks, %.1f samples/thread)\n", sgi, tab.seg[sgi].coarse ? "pair" : "single", nb, samples_per_thread);
        stat(st, "start us"); stat(le, "loop end us"); stat(en, "wave end us"); stat(loop_us, "loop us"); stat(mhz, "MHz"); stat(cyc, "loop cycles");
        std::sort(cyc.begin(), cyc.end());
        printf("   cycles per sample(-pair) per wave: %.1f (ideal issue, 2 waves/SIMD: %d)\n", cyc[cyc.size() / 2] / samples_per_thread,
               tab.seg[sgi].coarse ? 32 * 7 * 4 * 2 : 32 * 4 * 4 * 2);
    }
    // placement: waves per (XCC, SE, CU, SIMD)
    std::vector<int> hist(1 << 16, 0);
    int simds_used = 0, max_per_simd = 0;
    for (size_t w = 0; w < (size_t)total * 4; ++w) {
        const unsigned hw = (unsigned)p[w * 5 + 4];
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7, xcc = (hw >> 20) & 15;   // gfx9 HW_ID fields (xcc from XCC_ID not here)
        const unsigned key = (se << 7) | (sh << 6) | (cu << 2) | simd;
        (void)xcc;
        if (hist[key]++ == 0) ++simds_used;
        max_per_simd = std::max(max_per_simd, hist[key]);
    }
    {   // where do the early finishers live?  (loop end more than 12 % before the median)
        std::vector<double> le;
        for (size_t w = 0; w < (size_t)total * 4; ++w) le.push_back((p[w * 5 + 1] - t0) / 100.0);
        std::vector<double> srt = le;
        std::sort(srt.begin(), srt.end());
        const double med = srt[srt.size() / 2];
        int early_xcc[16] = {0}, all_xcc[16] = {0}, early_slot[16] = {0}, early_seg[3] = {0};
        int early_se[8] = {0};
        for (size_t w = 0; w < (size_t)total * 4; ++w) {
            const unsigned hw = (unsigned)p[w * 5 + 4], xcc = (unsigned)(p[w * 5 + 4] >> 32) & 15;
            all_xcc[xcc]++;
            if (le[w] < 0.88 * med) {
                early_xcc[xcc]++;
                early_slot[hw & 15]++;
                early_se[(hw >> 13) & 7]++;
                const int blk = (int)(w / 4);
                early_seg[blk < tab.seg[1].block0 ? 0 : (blk < tab.seg[2].block0 ? 1 : 2)]++;
            }
        }
        printf("median loop end %.1f us; early waves per XCC:", med);
        for (int i = 0; i < 8; ++i) printf(" %d/%d", early_xcc[i], all_xcc[i]);
        printf("\n  early by wave slot:");
        for (int i = 0; i < 4; ++i) printf(" [%d]=%d", i, early_slot[i]);
        printf("  by SE:");
        for (int i = 0; i < 8; ++i) printf(" %d", early_se[i]);
        printf("  by segment: %d %d %d\n", early_seg[0], early_seg[1], early_seg[2]);
    }
    int slot_hist[16] = {0};
    for (size_t w = 0; w < (size_t)total * 4; ++w) slot_hist[p[w * 5 + 4] & 15]++;
    printf("wave slots:");
    for (int i = 0; i < 16; ++i) if (slot_hist[i]) printf(" [%d]=%d", i, slot_hist[i]);
    printf("\n");
    printf("HW_ID: distinct (se,sh,cu,simd) keys %d (one XCC's view; x8 XCCs), max waves per key %d\n", simds_used, max_per_simd);
    return 0;
}
