// Stand-alone timing of the data-gradient kernel edge_bwd_kernel (edge_bwd2_impl.h) on random data (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -DMPG_SINGLE_VARIANT=<0|1|2> [-DMPG_B2STAMP] [-DMPG_B2EXP=n]
//         tools/ubench/bwd2_bench.hip mpgan_amd/csrc/edge.hip -o bwd2_bench
//   bwd2_bench [B=256] [needw=1] [ragged=0|1|2 (2: scattered masks)] [N=30] [SC=1]
// Prints the launch time and a checksum of da / dc / the parked fragments (to see that an experiment changed nothing).
#include "../../mpgan_amd/csrc/edge_bwd2.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
extern "C" int mpg_pack_weights(const float* W, int ldw, int rows, int cols, int transpose, float scale, int f16, void* img, void* stream);
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256;
    const bool needw = argc > 2 ? atoi(argv[2]) != 0 : true;
    const int ragged = argc > 3 ? atoi(argv[3]) : 0;
    const int N = argc > 4 ? atoi(argv[4]) : 30, SC = argc > 5 ? atoi(argv[5]) : 1;
    const int RB = (N + 31) / 32, nblk = B * RB * N;
    std::vector<float> ha((size_t)B * N * 96), hc((size_t)B * N * 96), hm(B * N, 1.f), hd((size_t)B * N * 192), hb(160), hw2(160 * 96), hw3(192 * 160);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f); };
    for (auto& x : ha) x = rnd(); for (auto& x : hc) x = rnd(); for (auto& x : hd) x = rnd() * 1e-3f; for (auto& x : hb) x = rnd() * 0.1f;
    for (auto& x : hw2) x = rnd() * 0.2f; for (auto& x : hw3) x = rnd() * 0.15f;
    if (ragged == 1) for (int b = 0; b < B; ++b) { int n = N * 2 / 5 + rand() % (N * 3 / 5 + 1); for (int j = n; j < N; ++j) hm[b * N + j] = 0.f; }
    if (ragged == 2) for (int b = 0; b < B; ++b) { int n = N * 2 / 5 + rand() % (N * 3 / 5 + 1); for (int j = 0; j < N; ++j) hm[b * N + j] = 0.f; for (int k = 0; k < n; ) { int j = rand() % N; if (hm[b * N + j] == 0.f) { hm[b * N + j] = 1.f; ++k; } } }
    float *a, *c, *m, *d, *b2, *w2, *w3, *da, *dc; void *i2, *i3t, *i2t, *sE, *sZ; unsigned int* sg; uint64_t* seed; int* gexp;
    const size_t stb = (size_t)nblk * 10240, dab = (size_t)SC * B * N * 96 * 4, dcb = (size_t)RB * B * N * 96 * 4;
    hipMalloc(&a, ha.size() * 4); hipMalloc(&c, hc.size() * 4); hipMalloc(&m, hm.size() * 4); hipMalloc(&d, hd.size() * 4);
    hipMalloc(&b2, 160 * 4); hipMalloc(&w2, hw2.size() * 4); hipMalloc(&w3, hw3.size() * 4);
    hipMalloc(&da, dab); hipMalloc(&dc, dcb); hipMalloc(&sE, stb); hipMalloc(&sZ, stb); hipMalloc(&gexp, B * RB * 4);
    { std::vector<uint16_t> he(stb / 2); unsigned int st = 777u; for (auto& x : he) { st = st * 1664525u + 1013904223u; x = (uint16_t)(0x3000 + ((st >> 8) & 0xfff) + (((st >> 24) & 1) << 15)); } hipMemcpy(sE, he.data(), stb, hipMemcpyHostToDevice); }  // the forward's parked E2: fp16 values of either sign
    hipMemset(sZ, 0, stb); hipMemset(da, 0xff, dab); hipMemset(dc, 0xff, dcb);
    hipMalloc(&i2, 2 * 30 * 1024); hipMalloc(&i3t, 2 * 60 * 1024); hipMalloc(&i2t, 2 * 30 * 1024);
    hipMalloc(&sg, (size_t)nblk * 192 * 4); hipMalloc(&seed, 8);
    hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(m, hm.data(), hm.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b2, hb.data(), 160 * 4, hipMemcpyHostToDevice);
    hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w3, hw3.data(), hw3.size() * 4, hipMemcpyHostToDevice);
    { std::vector<unsigned int> hs((size_t)nblk * 192); unsigned int st = 12345u; for (auto& x : hs) { st = st * 1664525u + 1013904223u; x = st ^ (st >> 15); }   /* (own generator: rand() is shared with the HIP runtime's threads once it is up, and its sequence then differs from process to process) */ hipMemcpy(sg, hs.data(), hs.size() * 4, hipMemcpyHostToDevice); }
    hipMemset(seed, 1, 8);
    const float dscale = MPG_SINGLE_VARIANT == 2 ? 2.f : (MPG_SINGLE_VARIANT == 1 ? 256.f / 179.f : 1.f);
    mpg_pack_weights(w2, 96, 160, 96, 0, 16.f * dscale, 1, i2, nullptr);
    mpg_pack_weights(w3, 160, 160, 192, 1, 64.f * dscale, 1, i3t, nullptr);
    mpg_pack_weights(w2, 96, 96, 160, 1, 16.f * dscale, 1, i2t, nullptr);
    MpgEdgeBwd p = {};
    p.a = a; p.c = c; p.mask = m; p.dagg = d; p.ld_dagg = 192; p.sign3 = sg; p.W2img = i2; p.W3Timg = i3t; p.W2Timg = i2t; p.b2 = b2;
    p.B = B; p.N = N; p.SC = SC; p.gexp = gexp;
    p.alpha = 0.2f; p.agg_scale = 1.f; p.seed = seed; p.tag_base = 0; p.thr = MPG_SINGLE_VARIANT == 2 ? 128 : (MPG_SINGLE_VARIANT == 1 ? 77 : 0);
    p.dscale = dscale; p.f16 = 1;
    p.da = da; p.dc = dc; p.stageE2 = sE; p.stageZ2 = needw ? sZ : nullptr;
    for (int i = 0; i < 3; ++i) if (int e = mpg_edge_bwd(&p, nullptr)) { printf("launch error %d\n", e); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("sync error\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 20;
    hipEventRecord(e0);
    for (int i = 0; i < R; ++i) mpg_edge_bwd(&p, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    {   // run-to-run determinism: two more launches, outputs compared word for word
        auto grab = [&](void* dev, size_t n) { std::vector<unsigned int> h(n / 4); hipMemcpy(h.data(), dev, n, hipMemcpyDeviceToHost); return h; };
        mpg_edge_bwd(&p, nullptr); hipDeviceSynchronize();
        auto a0 = grab(da, dab), c0 = grab(dc, dcb), z0 = grab(sZ, needw ? stb : 4);
        mpg_edge_bwd(&p, nullptr); hipDeviceSynchronize();
        auto a1 = grab(da, dab), c1 = grab(dc, dcb), z1 = grab(sZ, needw ? stb : 4);
        size_t na = 0, nc = 0, nz = 0, fa = (size_t)-1, fc = (size_t)-1, fz = (size_t)-1;
        for (size_t i = 0; i < a0.size(); ++i) if (a0[i] != a1[i]) { if (!na) fa = i; ++na; }
        for (size_t i = 0; i < c0.size(); ++i) if (c0[i] != c1[i]) { if (!nc) fc = i; ++nc; }
        for (size_t i = 0; i < z0.size(); ++i) if (z0[i] != z1[i]) { if (!nz) fz = i; ++nz; }
        printf("  determinism: da %zu of %zu words differ (first %zu = jet %zu row %zu feat %zu), dc %zu (first %zu), dZ2 %zu (first word %zu = block %zu frag %zu lane %zu)\n",
               na, a0.size(), fa, fa / (N * 96), (fa / 96) % N, fa % 96, nc, fc, nz, fz, fz / 2560, (fz % 2560) / 256, (fz % 256) / 4);
    }
    auto checksum = [](void* dev, size_t n) {
        std::vector<unsigned int> h(n / 4); hipMemcpy(h.data(), dev, n, hipMemcpyDeviceToHost);
        unsigned long long s = 0; for (size_t i = 0; i < h.size(); ++i) s = s * 1099511628211ull + h[i];
        return s;
    };
#ifdef MPG_B2STAMP
    {
        std::vector<unsigned long long> st(64 * 4 * 8);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_b2_stamps), st.size() * 8);
        double avg[7] = {0}; int n = 0;
        for (int g = 0; g < 64 * 4; ++g) { if (st[g * 8 + 6] <= st[g * 8]) continue; ++n; for (int i = 1; i < 7; ++i) avg[i] += (double)(st[g * 8 + i] - st[g * 8 + i - 1]); }
        printf("  s_memtime ticks (100 MHz) per phase of a pair, avg over %d waves: setup %.0f | phase B %.0f | phase C %.0f | dZ1 epilogue %.0f\n",
               n, (avg[1] + avg[2] + avg[3]) / n, avg[4] / n, avg[5] / n, avg[6] / n);
    }
#endif
    printf("  inputs: a %016llx c %016llx dagg %016llx mask %016llx sign %016llx W3T %016llx W2T %016llx\n", checksum(a, ha.size() * 4), checksum(c, hc.size() * 4),
           checksum(d, hd.size() * 4), checksum(m, hm.size() * 4), checksum(sg, (size_t)nblk * 192 * 4), checksum(i3t, 2 * 60 * 1024), checksum(i2t, 2 * 30 * 1024));
#ifdef MPG_B1_STAMP
    {
        std::vector<unsigned long long> st((size_t)B * RB * SC * 8 * 10);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(b1_stamps), st.size() * 8);
        double acc[5] = {}, pro = 0, epi = 0, kclk = 0, ticks = 0; long n = 0, nw = 0;
        for (size_t wv = 0; wv < st.size() / 10; ++wv) {
            const unsigned long long* o = &st[wv * 10];
            if (o[5] < 3) continue;
            for (int q = 0; q < 5; ++q) acc[q] += (double)o[q];
            n += (long)o[5]; pro += (double)o[6]; epi += (double)o[7]; kclk += (double)o[8]; ticks += (double)o[9]; ++nw;
        }
        printf("  clk per sender (waves with >= 3 senders, %ld senders): setup %.0f  B %.0f  gate %.0f  C %.0f  dZ1 %.0f  = %.0f\n"
               "  per wave: prologue %.0f  epilogue %.0f  kernel %.0f clk = %.1f us at %.0f MHz\n",
               n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, (acc[0] + acc[1] + acc[2] + acc[3] + acc[4]) / n,
               pro / nw, epi / nw, kclk / nw, ticks / nw / 100.0, kclk / ticks * 100.0);
    }
#endif
    {   // da as numbers (its sum over senders is ordered differently by the four- and the eight-wave form: compare to ~1e-6)
        std::vector<float> h(dab / 4); hipMemcpy(h.data(), da, dab, hipMemcpyDeviceToHost);
        double s1 = 0, s2 = 0; for (float v : h) { s1 += v; s2 += fabs(v); }
        printf("  da sum %.9g |sum| %.9g\n", s1, s2);
    }
#ifdef MPG_BWD1
    printf("8 waves ");
#else
    printf("4 waves ");
#endif
    printf("DROP=%d B=%d N=%d SC=%d %s ragged=%d: %.1f us   checksums da %016llx dc %016llx E2 %016llx dZ2 %016llx\n",
           MPG_SINGLE_VARIANT, B, N, SC, needw ? "dW" : "data", ragged, ms * 1e3f / R, checksum(da, dab), checksum(dc, dcb),
           checksum(sE, stb), needw ? checksum(sZ, stb) : 0ull);
    return 0;
}
