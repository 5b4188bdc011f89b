// Sustained dense f16 MFMA rate of the chip with no memory traffic at all: every wave keeps its operands in registers and
// issues back-to-back independent MFMAs.  This is the ceiling any conv/GEMM kernel on this part can approach; DESIGN.md
// section 4 compares conv_pglds against it (the 2.5 PFLOP/s datasheet peak assumes 2.4 GHz with every MFMA slot filled,
// which the power envelope does not sustain).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_envelope.hip -o build/mfma_envelope && build/mfma_envelope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// 16 independent accumulators (the shape conv_pglds uses: acc[4][4]), 4 a-fragments x 4 b-fragments per k-step
__global__ __launch_bounds__(512, 1) void mfma16(const f16x8 *__restrict__ src, float *__restrict__ dst, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(lane + 64 * i) & 1023]; b[i] = src[(lane + 64 * (i + 4)) & 1023]; }
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    f32x4 s = {};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    if (s[0] == 123.456f) dst[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(512, 1) void mfma32(const f16x8 *__restrict__ src, float *__restrict__ dst, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a[2], b[2];
    for (int i = 0; i < 2; ++i) { a[i] = src[(lane + 64 * i) & 1023]; b[i] = src[(lane + 64 * (i + 2)) & 1023]; }
    f32x16 acc[2][2] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    f32x16 s = {};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) s += acc[i][j];
    if (s[0] == 123.456f) dst[blockIdx.x * blockDim.x + threadIdx.x] = s[0];
}

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

// int8: v_mfma_i32_16x16x64_i8, the same cycles as the f16 16x16x32 at twice the K
__global__ __launch_bounds__(512, 1) void mfma16_i8(const f16x8 *__restrict__ src, float *__restrict__ dst, int iters) {
    const int lane = threadIdx.x & 63;
    i32x4v a[4], b[4];
    const i32x4v *s4 = (const i32x4v *)src;
    for (int i = 0; i < 4; ++i) { a[i] = s4[(lane + 64 * i) & 1023]; b[i] = s4[(lane + 64 * (i + 4)) & 1023]; }
    i32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    i32x4 s = {};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    if (s[0] == 123456789) dst[blockIdx.x * blockDim.x + threadIdx.x] = (float)(s[0] + s[1] + s[2] + s[3]);
}

// LDS read rate alone: every wave streams conflict-free ds_read_b128 from a 16-KiB image (the conv kernels' fragment
// reads); calibrates SQ_LDS_IDX_ACTIVE and gives the ceiling the MFMA kernels' operand traffic is priced against
__global__ __launch_bounds__(512, 1) void lds_read(const f16x8 *__restrict__ src, float *__restrict__ dst, int iters) {
    __shared__ f16x8 img[1024];
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) img[e] = src[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc = {};
    const unsigned base = (unsigned)(size_t)(&img[0]) + (unsigned)lane * 16u;      // LDS byte address of this lane's slot
    for (int it = 0; it < iters; ++it) {
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)     // volatile asm: the compiler may neither hoist nor drop the reads
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(base + (unsigned)(((wave + k) & 15) * 1024)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += v[k];
    }
    if (acc[0] == 123.456f) dst[blockIdx.x * blockDim.x + threadIdx.x] = acc[1];
}

static double run(void (*k)(const f16x8 *, float *, int), int wg, int threads, const f16x8 *src, float *dst, int iters,
                  double flop_per_wave_iter, int reps, double *ms_out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(wg), dim3(threads), 0, 0, src, dst, iters);     // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k, dim3(wg), dim3(threads), 0, 0, src, dst, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / reps;
    const double waves = (double)wg * threads / 64;
    return waves * iters * flop_per_wave_iter * reps / (ms * 1e-3) / 1e12;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s  CUs %d  clock %d MHz\n", prop.gcnArchName, ncu, prop.clockRate / 1000);
    f16x8 *src; float *dst;
    CK(hipMalloc(&src, 1024 * sizeof(f16x8))); CK(hipMalloc(&dst, 1 << 24));
    std::vector<_Float16> h(8192);
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int wpc : {8, 16}) {
            const int wg = ncu * wpc / 8, iters = 20000;
            hipLaunchKernelGGL(lds_read, dim3(wg), dim3(512), 0, 0, src, dst, iters);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(lds_read, dim3(wg), dim3(512), 0, 0, src, dst, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double bytes = (double)wg * 8 * iters * 16 * 1024 * 5;
            printf("lds_read ds_read_b128 %2d waves/CU  %8.3f ms/launch  %7.1f TB/s  (%.0f B/clk/CU at 2.4 GHz)\n", wpc, ms / 5,
                   bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / ncu / 2.4e9);
        }
    }
    for (int mode = 0; mode < 3; ++mode) {           // operand data: zeros / small-range activations / full random bits
        srand(7);
        for (auto &v : h) {
            float r = (float)rand() / RAND_MAX;
            v = mode == 0 ? (_Float16)0.f : mode == 1 ? (_Float16)(r * 2.f - 0.6f) : (_Float16)((r - 0.5f) * 200.f);
        }
        CK(hipMemcpy(src, h.data(), 8192 * sizeof(_Float16), hipMemcpyHostToDevice));
        const char *mn = mode == 0 ? "zeros" : mode == 1 ? "activations" : "wide-range";
        for (int waves_per_cu : {4, 8, 16}) {
            const int threads = waves_per_cu >= 8 ? 512 : 256;
            const int wg = ncu * (waves_per_cu * 64 / threads);
            double ms;
            // 16 MFMAs of 16x16x32 per iteration: 2*16*16*32 flop each
            double tf16 = run(mfma16, wg, threads, src, dst, 20000, 16.0 * 2 * 16 * 16 * 32, 50, &ms);
            printf("%-12s 16x16x32  %2d waves/CU  %8.3f ms/launch  %7.1f TFLOP/s\n", mn, waves_per_cu, ms, tf16);
            double ti8 = run(mfma16_i8, wg, threads, src, dst, 20000, 16.0 * 2 * 16 * 16 * 64, 50, &ms);
            printf("%-12s i8 16x16x64 %2d waves/CU  %8.3f ms/launch  %7.1f TOP/s\n", mn, waves_per_cu, ms, ti8);
            double tf32 = run(mfma32, wg, threads, src, dst, 20000, 8.0 * 2 * 32 * 32 * 16, 50, &ms);
            printf("%-12s 32x32x16  %2d waves/CU  %8.3f ms/launch  %7.1f TFLOP/s\n", mn, waves_per_cu, ms, tf32);
        }
    }
    return 0;
}
