// Stand-alone timing of edge_fwd_kernel<DROP, f16, sign> at B=256, N=30 on random data (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -fno-slp-vectorize -DMPG_SINGLE_VARIANT=<0|2> [-DMPG_EXP=n] tools/ubench/fwd_bench.hip -o fwd_bench
#include "../../mpgan_amd/csrc/edge.hip"
#include <stdio.h>
#include <stdlib.h>
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
    hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(m, hm.data(), hm.size() * 4, hipMemcpyHostToDevice); hipMemcpy(b2, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w3, hw3.data(), hw3.size() * 4, hipMemcpyHostToDevice);
    hipMemset(seed, 1, 8);
    mpg_pack_weights(w2, 96, 160, 96, 0, 1.f, 1, i2, nullptr);
    mpg_pack_weights(w3, 160, 192, 160, 0, 1.f, 1, i3, nullptr);
    MpgEdgeFwd p = {};
    p.a = a; p.c = c; p.mask = m; p.W2img = i2; p.W3img = i3; p.b2 = b2; p.b3 = b2 + 160; p.agg = agg;
    p.B = B; p.N = N; p.SC = 1; p.alpha = 0.2f; p.agg_scale = 1.f; p.seed = seed; p.tag_base = 0;
    p.thr = MPG_SINGLE_VARIANT == 2 ? 128 : (MPG_SINGLE_VARIANT == 1 ? 77 : 0); p.dscale = 1.f; p.skip_masked = 1; p.f16 = 1; p.sign3 = sg;
    for (int i = 0; i < 3; ++i) if (int e = mpg_edge_fwd(&p, nullptr)) { printf("launch error %d\n", e); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 20;
    hipEventRecord(e0);
    for (int i = 0; i < R; ++i) mpg_edge_fwd(&p, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(8); hipMemcpy(h.data(), agg, 32, hipMemcpyDeviceToHost);
    printf("edge_fwd<%d,f16,sign> B=%d N=%d%s: %.1f us/launch   agg[0..3] = %g %g %g %g\n", MPG_SINGLE_VARIANT, B, N,
           argc > 2 ? " ragged" : "", ms * 1e3 / R, h[0], h[1], h[2], h[3]);
    return 0;
}
