// Stand-alone timing of wgrad2_kernel on the job mix of one MPLayer backward (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include -I tools/ubench [-DMPG_W2_EXP=n] tools/ubench/w2_bench.hip -o w2_bench
//   w2_bench [rows=7680] [splitk=15] [jobs mask=63]
#include "wgrad2.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 7680, SK = argc > 2 ? atoi(argv[2]) : 15, mask = argc > 3 ? atoi(argv[3]) : 63;
    const int shapes[6][2] = {{256, 256}, {256, 192}, {256, 32}, {32, 256}, {96, 32}, {96, 32}};
    W2Group G = {};
    int n = 0;
    G.wg0[0] = 0;
    uint32_t lcg = 12345u;
    for (int q = 0; q < 6; ++q) {
        if (!((mask >> q) & 1)) continue;
        const int N = shapes[q][0], K = shapes[q][1];
        std::vector<float> hd((size_t)M * N), hx((size_t)M * K);
        for (auto& v : hd) { lcg = lcg * 1664525u + 1013904223u; v = ((lcg >> 8) / 16777216.f - 0.5f) * 1e-3f; }
        for (auto& v : hx) { lcg = lcg * 1664525u + 1013904223u; v = (lcg >> 8) / 16777216.f - 0.5f; }
        float *d, *x, *p;
        hipMalloc(&d, hd.size() * 4); hipMalloc(&x, hx.size() * 4); hipMalloc(&p, (size_t)SK * N * (K + 1) * 4);
        hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice); hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
        W2Job& j = G.j[n];
        j.dy = d; j.x = x; j.part = p; j.split_stride = (long long)N * (K + 1); j.ldy = N; j.ldx = K; j.ldp = K + 1; j.N = N; j.K = K; j.M = M;
        j.out_scale = 1.f; j.hb = 1; j.dy_vec = 1;
        G.splitk[n] = SK;
        G.wg0[n + 1] = G.wg0[n] + ((N + 127) / 128) * ((K + 127) / 128) * SK;
        ++n;
    }
    G.n = n;
    for (int i = 0; i < 3; ++i) if (int e = mpg_wgrad2_launch(&G, nullptr)) { printf("launch error %d\n", e); return 1; }
    if (hipDeviceSynchronize() != hipSuccess) { printf("sync error\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int R = 20;
    hipEventRecord(e0);
    for (int i = 0; i < R; ++i) mpg_wgrad2_launch(&G, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> h(4); hipMemcpy(h.data(), G.j[0].part, 16, hipMemcpyDeviceToHost);
    printf("wgrad2 EXP=%d rows=%d splitk=%d jobs=%d wgs=%d: %.1f us/launch   part[0..3] = %g %g %g %g\n", MPG_W2_EXP, M, SK, mask, G.wg0[n], ms * 1e3 / R,
           h[0], h[1], h[2], h[3]);
    return 0;
}
