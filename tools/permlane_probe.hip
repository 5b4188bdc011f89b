// permlane_probe.hip -- what v_permlane32_swap does on gfx950, and le_rows.hip's quads <-> chunks exchange built on it.
//   hipcc --offload-arch=gfx950 -O2 tools/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
// Expected: r[0] = {lanes 0-31: a[0..31], lanes 32-63: b[0..31]}, r[1] = {lanes 0-31: a[32..63], lanes 32-63: b[32..63]}.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned *in, unsigned *out)
{
    const unsigned a = in[threadIdx.x], b = in[64 + threadIdx.x];
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
}
int main()
{
    unsigned h[128], o[128], *d, *e;
    for (int i = 0; i < 64; ++i) { h[i] = 0xa000 + i; h[64 + i] = 0xb000 + i; }
    hipMalloc(&d, sizeof h); hipMalloc(&e, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const unsigned w0 = i < 32 ? 0xa000 + i : 0xb000 + (i - 32), w1 = i < 32 ? 0xa000 + 32 + i : 0xb000 + i;
        bad += o[i] != w0 || o[64 + i] != w1;
    }
    printf("lane  0: r0 %x r1 %x | lane 33: r0 %x r1 %x | mismatches vs the expected semantics: %d\n", o[0], o[64], o[33], o[97], bad);
    return bad != 0;
}
