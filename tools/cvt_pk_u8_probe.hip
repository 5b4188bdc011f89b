// Diagnostic: does v_cvt_pk_u8_f32 round to nearest even and saturate by itself?  (then quant4's rint + med3 are redundant)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/cvt_pk_u8_probe.hip -o /tmp/cvtprobe && /tmp/cvtprobe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *x, unsigned *a, unsigned *b, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(__builtin_rintf(x[i]), 0.f, 255.f), 0, 0u);
    b[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
}
int main()
{
    std::vector<float> h;
    for (int q = -8; q <= 262; ++q)
        for (float d : {-0.5f, -0.49999997f, -0.25f, 0.f, 0.25f, 0.49999997f, 0.5f, 0.50000006f})
            h.push_back((float)q + d);
    for (float v : {1e9f, -1e9f, 255.5f, 255.49f, 256.f, 1e-30f, -1e-30f, -0.f, INFINITY, -INFINITY}) h.push_back(v);
    for (int i = 0; i < 200000; ++i) h.push_back(-4.f + 264.f * (float)((i * 2654435761u) >> 8 & 0xffffff) / 16777216.f);
    const int n = (int)h.size();
    float *dx; unsigned *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, da, db, n);
    std::vector<unsigned> a(n), b(n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (a[i] != b[i] && bad++ < 20) printf("x=%.9g  rint+med3+cvt=%u  cvt alone=%u\n", h[i], a[i], b[i]);
    printf("%d values, %d differ\n", n, bad);
    return 0;
}
