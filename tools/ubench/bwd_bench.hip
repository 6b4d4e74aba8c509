// Stand-alone timing of edge_bwd_kernel<DROP, f16, needw> at B=256, N=30 on random data (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -DMPG_SINGLE_VARIANT=0 [-DMPG_PF=n] tools/ubench/bwd_bench.hip -o bwd_bench
#include "../../mpgan_amd/csrc/edge_bwd.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
extern "C" int mpg_pack_weights(const float* W, int ldw, int rows, int cols, int transpose, float scale, int f16, void* img, void* stream);
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 256, N = 30, RB = 1, nblk = B * RB * N;
    const bool needw = argc > 2 ? atoi(argv[2]) != 0 : true;
    std::vector<float> ha(B * N * 96), hc(B * N * 96), hm(B * N, 1.f), hd((size_t)B * N * 192), hb(160), hw2(160 * 96), hw3(192 * 160);
    srand(1);
    auto rnd = [] { return (rand() / (float)RAND_MAX - 0.5f); };
    for (auto& x : ha) x = rnd(); for (auto& x : hc) x = rnd(); for (auto& x : hd) x = rnd() * 1e-3f; for (auto& x : hb) x = rnd() * 0.1f;
    for (auto& x : hw2) x = rnd() * 0.2f; for (auto& x : hw3) x = rnd() * 0.15f;
    if (argc > 3) for (int b = 0; b < B; ++b) { int n = 12 + rand() % 19; for (int j = n; j < N; ++j) hm[b * N + j] = 0.f; }
    float *a, *c, *m, *d, *b2, *w2, *w3, *da, *dc; void *i2, *i3t, *i2t, *sE, *sZ; unsigned int* sg; uint64_t* seed;
    hipMalloc(&a, ha.size() * 4); hipMalloc(&c, hc.size() * 4); hipMalloc(&m, hm.size() * 4); hipMalloc(&d, hd.size() * 4);
    hipMalloc(&b2, 160 * 4); hipMalloc(&w2, hw2.size() * 4); hipMalloc(&w3, hw3.size() * 4);
    hipMalloc(&da, (size_t)B * N * 96 * 4); hipMalloc(&dc, (size_t)B * N * 96 * 4);
    hipMalloc(&i2, 2 * 30 * 1024); hipMalloc(&i3t, 2 * 60 * 1024); hipMalloc(&i2t, 2 * 30 * 1024);
    hipMalloc(&sE, (size_t)nblk * 20480); hipMalloc(&sZ, (size_t)nblk * 20480); hipMalloc(&sg, (size_t)nblk * 192 * 4); hipMalloc(&seed, 8);
    hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(m, hm.data(), hm.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(b2, hb.data(), 160 * 4, hipMemcpyHostToDevice);
    hipMemcpy(w2, hw2.data(), hw2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(w3, hw3.data(), hw3.size() * 4, hipMemcpyHostToDevice);
    hipMemset(sg, 0x5a, (size_t)nblk * 192 * 4); hipMemset(seed, 1, 8);
    mpg_pack_weights(w2, 96, 160, 96, 0, 1.f, 1, i2, nullptr);
    mpg_pack_weights(w3, 160, 160, 192, 1, 1.f, 0, i3t, nullptr);
    mpg_pack_weights(w2, 96, 96, 160, 1, 1.f, 0, i2t, nullptr);
    MpgEdgeBwd p = {};
    p.a = a; p.c = c; p.mask = m; p.dagg = d; p.ld_dagg = 192; p.sign3 = sg; p.W2img = i2; p.W3Timg = i3t; p.W2Timg = i2t; p.b2 = b2;
    p.da = da; p.dc = dc; p.stageE2 = needw ? sE : nullptr; p.stageZ2 = needw ? sZ : nullptr; p.B = B; p.N = N; p.SC = 1;
    p.alpha = 0.2f; p.agg_scale = 1.f; p.seed = seed; p.tag_base = 0; p.thr = MPG_SINGLE_VARIANT == 2 ? 128 : 0; p.dscale = 1.f; p.f16 = 1;
    for (int i = 0; i < 3; ++i) if (int e = mpg_edge_bwd(&p, nullptr)) { printf("launch error %d\n", e); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("sync error\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 20;
    hipEventRecord(e0);
    for (int i = 0; i < R; ++i) mpg_edge_bwd(&p, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(4); hipMemcpy(h.data(), da, 16, hipMemcpyDeviceToHost);
    printf("edge_bwd<%d,f16,%s> B=%d N=%d%s: %.1f us/launch   da[0..3] = %g %g %g %g\n", MPG_SINGLE_VARIANT, needw ? "dW" : "data", B, N,
           argc > 3 ? " ragged" : "", ms * 1e3 / R, h[0], h[1], h[2], h[3]);
    return 0;
}
