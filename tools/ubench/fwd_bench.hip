// Stand-alone timing of edge_fwd_kernel<DROP, f16, sign> at B=256, N=30 on random data (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -fno-slp-vectorize -DMPG_SINGLE_VARIANT=<0|2> [-DMPG_EXP=n] tools/ubench/fwd_bench.hip -o fwd_bench
//   -DMPG_FWD1 [-DMPG_F1_STAGGER=n]: the eight-wave form (edge_fwd1_impl.h) in place of the four-wave one; the checksums of agg,
//   of the sign words and of the parked E2 fragments printed at the end must agree between the two (agg up to the order of its sums)
#include "../../mpgan_amd/csrc/edge.hip"
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, N = 30, RB = 1;
    std::vector<float> ha(B * N * 96), hc(B * N * 96), hm(B * N, 1.f), hb(160 + 192), hw2(160 * 96), hw3(192 * 160);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f); };
    for (auto& x : ha) x = rnd(); for (auto& x : hc) x = rnd(); for (auto& x : hb) x = rnd() * 0.1f;
    for (auto& x : hw2) x = rnd() * 0.2f; for (auto& x : hw3) x = rnd() * 0.15f;
    if (argc > 2) for (int b = 0; b < B; ++b) { int n = 12 + rand() % 19; for (int j = n; j < N; ++j) hm[b * N + j] = 0.f; }
    float *a, *c, *m, *b2, *w2, *w3, *agg; void *i2, *i3; unsigned int* sg; uint64_t* seed;
    hipMalloc(&a, ha.size() * 4); hipMalloc(&c, hc.size() * 4); hipMalloc(&m, hm.size() * 4); hipMalloc(&b2, hb.size() * 4);
    hipMalloc(&w2, hw2.size() * 4); hipMalloc(&w3, hw3.size() * 4); hipMalloc(&agg, (size_t)B * N * 192 * 4);
    hipMalloc(&i2, 2 * NF2 * 1024); hipMalloc(&i3, 2 * NF3 * 1024); hipMalloc(&sg, (size_t)B * RB * N * 192 * 4); hipMalloc(&seed, 8);
    void* stE2; hipMalloc(&stE2, (size_t)B * RB * N * 10240); hipMemset(stE2, 0, (size_t)B * RB * N * 10240); hipMemset(sg, 0, (size_t)B * RB * N * 192 * 4);
    hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(m, hm.data(), hm.size() * 4, hipMemcpyHostToDevice); hipMemcpy(b2, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w3, hw3.data(), hw3.size() * 4, hipMemcpyHostToDevice);
    hipMemset(seed, 1, 8);
    mpg_pack_weights(w2, 96, 160, 96, 0, 1.f, 1, i2, nullptr);
    mpg_pack_weights(w3, 160, 192, 160, 0, 1.f, 1, i3, nullptr);
    MpgEdgeFwd p = {};
    p.a = a; p.c = c; p.mask = m; p.W2img = i2; p.W3img = i3; p.b2 = b2; p.b3 = b2 + 160; p.agg = agg;
    p.B = B; p.N = N; p.SC = 1; p.alpha = 0.2f; p.agg_scale = 1.f; p.seed = seed; p.tag_base = 0;
    p.thr = MPG_SINGLE_VARIANT == 2 ? 128 : (MPG_SINGLE_VARIANT == 1 ? 77 : 0); p.dscale = 1.f; p.skip_masked = 1; p.f16 = 1; p.sign3 = sg; p.stageE2 = stE2;
    for (int i = 0; i < 3; ++i) if (int e = mpg_edge_fwd(&p, nullptr)) { printf("launch error %d\n", e); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 20;
    hipEventRecord(e0);
    for (int i = 0; i < R; ++i) mpg_edge_fwd(&p, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h((size_t)B * N * 192); hipMemcpy(h.data(), agg, h.size() * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned int> hs((size_t)B * RB * N * 192); hipMemcpy(hs.data(), sg, hs.size() * 4, hipMemcpyDeviceToHost);
    std::vector<unsigned short> he((size_t)B * RB * N * 5120); hipMemcpy(he.data(), stE2, he.size() * 2, hipMemcpyDeviceToHost);
    double sa = 0, saa = 0; for (float v : h) { sa += v; saa += fabs(v); }
    unsigned long long cs = 0, ce = 0; for (size_t i = 0; i < hs.size(); ++i) cs += hs[i] * (unsigned long long)(i % 1009 + 1);
    for (size_t i = 0; i < he.size(); ++i) ce += he[i] * (unsigned long long)(i % 1013 + 1);
#ifdef MPG_FWD1
    const char* form = "8 waves";
#else
    const char* form = "4 waves";
#endif
    printf("edge_fwd<%d,f16,sign> %s B=%d N=%d%s: %.1f us/launch   agg sum %.9g |sum| %.9g  signs %llx  E2 %llx\n", MPG_SINGLE_VARIANT, form, B, N,
           argc > 2 ? " ragged" : "", ms * 1e3 / R, sa, saa, cs, ce);
#ifdef MPG_F1_STAMP
    {   // per-section clk per sender, over the waves with >= 3 senders; prologue / loop / epilogue and the clock per wave
        std::vector<unsigned long long> st((size_t)B * 8 * 8);
        hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(f1_stamps), st.size() * 8);
        double acc[4] = {}, pro = 0, epi = 0, kclk = 0, ticks = 0; long n = 0, nw = 0;
        for (int wv = 0; wv < B * 8; ++wv) {
            const unsigned long long* o = &st[(size_t)wv * 8];
            if (o[5] < 3) continue;
            for (int q = 0; q < 4; ++q) acc[q] += (double)o[q];
            n += (long)o[5]; pro += (double)o[4]; epi += (double)o[6]; kclk += (double)(o[7] >> 20); ticks += (double)(o[7] & 0xfffff); ++nw;
        }
        printf("  clk per sender (waves with >= 3 senders, %ld senders): setup %.0f  layer2 %.0f  E2 %.0f  layer3 %.0f  = %.0f\n"
               "  per wave: prologue %.0f  epilogue %.0f  kernel %.0f clk = %.1f us at %.0f MHz\n",
               n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, (acc[0] + acc[1] + acc[2] + acc[3]) / n,
               pro / nw, epi / nw, kclk / nw, ticks / nw / 100.0, kclk / ticks * 100.0);
    }
#endif
    return 0;
}
