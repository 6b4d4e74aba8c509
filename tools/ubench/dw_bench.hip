// Stand-alone timing of edge_dw_kernel<DROP> at N=30 on random data (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -DMPG_SINGLE_VARIANT=<0|2> [-DMPG_DW_EXP=n] tools/ubench/dw_bench.hip -o dw_bench
//   dw_bench [B=256] [ragged]
//   -DMPG_DW8: also compiles the uniform eight-wave form (every wave builds and multiplies; measured slower, see edge_dw.hip);
//   MPG_DW_FORM=uniform in the environment picks it -- the checksums of both forms must agree bit for bit
#include "../../mpgan_amd/csrc/edge_dw.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, N = 30, RB = 1, nblk = B * RB * N;
    std::vector<float> ha(B * N * 96), hc(B * N * 96), hm(B * N, 1.f), hd((size_t)B * N * 192);
    std::vector<uint16_t> hs((size_t)nblk * 10 * 64 * 8);
    std::vector<int> he(B * RB);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f); };
    for (auto& x : ha) x = rnd(); for (auto& x : hc) x = rnd(); for (auto& x : hd) x = rnd() * 1e-3f;
    for (auto& x : hs) x = (uint16_t)(0x3000 + (rand() & 0xfff) + ((rand() & 1) << 15));  // fp16 values of magnitude 0.1 .. 1
    for (auto& x : he) x = 12 + rand() % 3;
    if (argc > 2) for (int b = 0; b < B; ++b) { int n = 12 + rand() % 19; for (int j = n; j < N; ++j) hm[b * N + j] = 0.f; }
    float *a, *c, *m, *d, *part, *dW3, *dW2, *db3, *db2; void *sE, *sZ; unsigned int* sg; uint64_t* seed; int* gexp;
    const int nwg = nblk < 256 ? nblk : (256 > (nblk + 63) / 64 ? 256 : (nblk + 63) / 64);
    hipMalloc(&a, ha.size() * 4); hipMalloc(&c, hc.size() * 4); hipMalloc(&m, hm.size() * 4); hipMalloc(&d, hd.size() * 4);
    hipMalloc(&sE, hs.size() * 2); hipMalloc(&sZ, hs.size() * 2); hipMalloc(&sg, (size_t)nblk * 192 * 4); hipMalloc(&seed, 8); hipMalloc(&gexp, he.size() * 4);
    hipMalloc(&part, (size_t)nwg * 46432 * 4); hipMalloc(&dW3, 192 * 160 * 4); hipMalloc(&dW2, 160 * 96 * 4); hipMalloc(&db3, 192 * 4); hipMalloc(&db2, 160 * 4);
    hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(m, hm.data(), hm.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(sE, hs.data(), hs.size() * 2, hipMemcpyHostToDevice); hipMemcpy(sZ, hs.data(), hs.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(gexp, he.data(), he.size() * 4, hipMemcpyHostToDevice);
    hipMemset(sg, 0x5a, (size_t)nblk * 192 * 4); hipMemset(seed, 1, 8);
    MpgEdgeDw p = {};
    p.a = a; p.c = c; p.mask = m; p.dagg = d; p.ld_dagg = 192; p.sign3 = sg; p.stageE2 = sE; p.stageZ2 = sZ; p.part = part; p.nwg = nwg;
    p.dW3 = dW3; p.dW2 = dW2; p.db3 = db3; p.db2 = db2; p.B = B; p.N = N; p.alpha = 0.2f; p.agg_scale = 1.f; p.seed = seed; p.gexp = gexp;
    p.tag_base = 0; p.thr = MPG_SINGLE_VARIANT == 2 ? 128 : 0; p.dscale = MPG_SINGLE_VARIANT == 2 ? 2.f : 1.f; p.f16 = 1;
    for (int i = 0; i < 3; ++i) if (int e = mpg_edge_dw(&p, nullptr)) { printf("launch error %d\n", e); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("sync error\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 20;
    hipEventRecord(e0);
    for (int i = 0; i < R; ++i) mpg_edge_dw(&p, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(4); hipMemcpy(h.data(), dW3, 16, hipMemcpyDeviceToHost);
#ifdef MPG_DWSTAMP
    {
        std::vector<unsigned long long> st(64 * 4 * 8);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_dw_stamps), st.size() * 8);
        const bool uni = getenv("MPG_DW_FORM") && !strcmp(getenv("MPG_DW_FORM"), "uniform");
        const char* nmr[6] = {"setup(+barrier before)", "dZ3", "requests+E1", "requests+dZ2", "E2", "barrier wait"};
        const char* nmu[6] = {"build", "consume", "barrier wait", "-", "-", "-"};
        const char** nm = uni ? nmu : nmr;
        double tot[8] = {0};
        for (int w = 0; w < 64 * 4; ++w) for (int q = 0; q < 8; ++q) tot[q] += (double)st[w * 8 + q];
        const double nb = tot[6] > 0 ? tot[6] : 1;
        printf("builder clocks per block (s_memtime ticks = shader clocks, ~1.4 GHz in this kernel), mean over 256 builder waves, %.1f blocks each:\n", nb / 256);
        for (int q = 0; q < 6; ++q) printf("  %-24s %8.1f\n", nm[q], tot[q] / nb);
    }
#endif
    {   // the outputs and the per-workgroup partials as checksums: the two forms of the kernel (MPG_DW_FORM=roles | uniform) must agree bit for bit
        auto cks = [](const void* dev, size_t n) {
            std::vector<unsigned int> hh(n / 4); hipMemcpy(hh.data(), dev, n, hipMemcpyDeviceToHost);
            unsigned long long s_ = 0; for (size_t i = 0; i < hh.size(); ++i) s_ = s_ * 1099511628211ull + hh[i];
            return s_;
        };
        printf("  checksums dW3 %016llx dW2 %016llx db3 %016llx db2 %016llx\n", cks(dW3, 192 * 160 * 4), cks(dW2, 160 * 96 * 4), cks(db3, 192 * 4), cks(db2, 160 * 4));
    }
    printf("edge_dw<%d> (+reduce) %s B=%d N=%d%s nwg=%d: %.1f us/launch   dW3[0..3] = %g %g %g %g\n", MPG_SINGLE_VARIANT,
           getenv("MPG_DW_FORM") ? getenv("MPG_DW_FORM") : "roles", B, N,
           argc > 2 ? " ragged" : "", nwg, ms * 1e3 / R, h[0], h[1], h[2], h[3]);
    return 0;
}
