// Do v_cvt_f16_f32 and v_cvt_pk_f16_f32 (gfx950) round every fp32 value to the same f16?  2^26 values: random bit patterns in
// [2^-20, 2^17), exact ties, f16 denormal range.  Build: hipcc --offload-arch=gfx950 -O2 tools/cvt_f16_probe.hip -o tools/build/cvt_f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(unsigned long long *count, unsigned *examples, unsigned n_per_thread)
{
    unsigned long long bad = 0;
    uint32_t s = 0x9e3779b9u * (blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (unsigned i = 0; i < n_per_thread; ++i) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        uint32_t bits = s;
        if ((i & 3) == 1) bits = (bits & 0xffffe000u) | 0x1000u;               // an exact tie between two f16 values
        if ((i & 3) == 2) bits = (bits & 0x807fffffu) | ((100u + (bits >> 23) % 14u) << 23);   // 2^-27 .. 2^-14: f16 denormals and below
        uint32_t e = (bits >> 23) & 0xff;
        if ((i & 3) != 2 && (e < 107 || e > 143)) bits = (bits & 0x807fffffu) | (120u << 23);
        float x = __uint_as_float(bits), y = __uint_as_float(bits ^ 0x00012345u);
        unsigned a, b, pk;
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(a) : "v"(x));
        asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(b) : "v"(y));
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(x), "v"(y));
        if ((a & 0xffff) != (pk & 0xffff) || (b & 0xffff) != (pk >> 16)) {
            if (bad == 0 && atomicAdd(&examples[0], 1u) < 8u) { unsigned k = atomicAdd(&examples[1], 4u); examples[2 + k] = bits; examples[3 + k] = a & 0xffff; examples[4 + k] = pk & 0xffff; examples[5 + k] = (i & 3); }
            ++bad;
        }
    }
    atomicAdd(count, bad);
}
int main()
{
    unsigned long long *count; unsigned *ex;
    hipMalloc(&count, 8); hipMalloc(&ex, 4 * 64); hipMemset(count, 0, 8); hipMemset(ex, 0, 4 * 64);
    probe<<<1024, 256>>>(count, ex, 256);
    unsigned long long h; unsigned he[64];
    hipMemcpy(&h, count, 8, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 256, hipMemcpyDeviceToHost);
    printf("values tested %llu, pairs where v_cvt_pk_f16_f32 != v_cvt_f16_f32: %llu\n", 1024ull * 256 * 256 * 2, h);
    for (unsigned k = 0; k + 4 <= he[1] && k < 32; k += 4) printf("  x bits %08x (%g): scalar %04x packed %04x class %u\n", he[2 + k], *(float *)&he[2 + k], he[3 + k], he[4 + k], he[5 + k]);
    return 0;
}
