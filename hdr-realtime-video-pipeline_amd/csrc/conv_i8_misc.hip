// conv_i8_misc.hip -- the two small kernels beside conv3x3_pglds_i8.hip on the W8A8 HG path:
//   * quant_i8: f16 tensor -> int8 codes of a consumer's activation quantiser (the one fp16 -> int8 boundary, conv2's
//     output; W8A8Conv2d.forward, hdrtvnet_torch.py:351-356: round((x - x_zero) / x_scale).clamp(0, 255), stored - 128);
//   * conv1x1_i8: the decoder's 1x1 fuse convolutions conv6 / conv7 / conv8 (Hallucination_arch.py:118-131) on
//     v_mfma_i32_16x16x64_i8: K = the concatenation of two int8 tensors that share one quantiser, no activation, output
//     re-quantised to the next layer's (signed-range) codes.  HBM-bound: 256 pixels x 128 output channels per block,
//     128-channel K chunks staged by LDS-DMA into a 3-deep ring, one barrier per chunk.
#include "launchers.h"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void quant_i8_kernel(const f16 *__restrict__ src, int8_t *__restrict__ dst, size_t n8,
                                                        float inv_scale, float zero_code)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const f16x8 v = reinterpret_cast<const f16x8 *>(src)[i];
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float q = fminf(fmaxf(__builtin_rintf((float)v[k] * inv_scale + zero_code), -128.f), 127.f);
            const unsigned b = (unsigned)(int)q & 0xffu;
            if (k < 4) lo |= b << (8 * k); else hi |= b << (8 * (k - 4));
        }
        reinterpret_cast<uint2 *>(dst)[i] = make_uint2(lo, hi);
    }
}

constexpr int TP = 256, BN = 128, CT = 128;           // pixels x output channels per block, K chunk (bytes per row)
constexpr int A_BYTES = TP * CT, B_BYTES = BN * CT;   // 32 KiB + 16 KiB per stage
constexpr int STAGE = A_BYTES + B_BYTES, NSTAGE = 3;
constexpr int SMEM = NSTAGE * STAGE;                  // 144 KiB

__device__ __forceinline__ void glds16(const void *g, void *lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__global__ __launch_bounds__(512) void conv1x1_i8_kernel(ConvI8Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int ntn = p.Cout / BN;
    const int n0 = (blockIdx.x % ntn) * BN;           // channel tiles of one pixel tile are neighbours: its K rows stay in L2
    const size_t px0 = (size_t)(blockIdx.x / ntn) * TP;
    const size_t npx = (size_t)p.Hi * p.Wi;
    const int nchunk0 = p.c0 / CT, nchunk = (p.c0 + p.c1) / CT;

    const int l_row = lane >> 3, l_slot = lane & 7;
    auto issue = [&](int cc, int stage) {
        const int8_t *src;
        int cs, coff;
        if (cc < nchunk0) { src = p.src0; cs = p.c0; coff = cc * CT; }
        else { src = p.src1; cs = p.c1; coff = (cc - nchunk0) * CT; }
        char *sa = smem + stage * STAGE, *sb = sa + A_BYTES;
#pragma unroll
        for (int it = 0; it < 4; ++it) {              // 32 pieces of 8 pixel rows
            const int piece = wave * 4 + it;
            const int r = piece * 8 + l_row;
            const size_t px = px0 + r;
            const int8_t *g = px < npx ? src + px * cs + coff + ((l_slot ^ (r & 7)) << 4) : p.padline + (l_slot << 4);
            glds16(g, sa + piece * 1024);
        }
        const int8_t *wb = p.wpk + ((size_t)cc * p.Cout + n0) * CT;
#pragma unroll
        for (int it = 0; it < 2; ++it) {              // 16 pieces of 8 weight rows
            const int piece = wave * 2 + it;
            const int n = piece * 8 + l_row;
            glds16(wb + (size_t)n * CT + ((l_slot ^ (n & 7)) << 4), sb + piece * 1024);
        }
    };

    // wave tiling as in conv3x3_pglds_i8: 2 (channels) x 4 (pixel groups of 64), 4x4 tiles of 16x16 each
    const int wc = wave & 1, wp = wave >> 1;
    i32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
    const int kw = l15 & 7;
    const int b_lane = (wc * 64 + l15) * CT, a_lane = (wp * 64 + l15) * CT;

    issue(0, 0);
    if (nchunk > 1) issue(1, 1);
    for (int cc = 0; cc < nchunk; ++cc) {
        if (cc + 1 < nchunk) wait_vm<6>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();                 // chunk cc is in LDS for everyone; stage (cc+2)%3 is no longer read
        if (cc + 2 < nchunk) issue(cc + 2, (cc + 2) % NSTAGE);
        const char *a = smem + (cc % NSTAGE) * STAGE + a_lane, *b = smem + (cc % NSTAGE) * STAGE + A_BYTES + b_lane;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x4 wf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wf[i] = *reinterpret_cast<const i32x4 *>(b + i * 16 * CT + (((ks * 4 + kg) ^ kw) << 4));
                xf[i] = *reinterpret_cast<const i32x4 *>(a + i * 16 * CT + (((ks * 4 + kg) ^ kw) << 4));
            }
#pragma unroll
            for (int m = 0; m < 16; ++m)
                acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[m >> 2], xf[m & 3], acc[m >> 2][m & 3], 0, 0, 0);
        }
    }
    __syncthreads();                                   // LDS is free: wave-private strips for the 16-byte stores

    // lane: pixel wp*64 + j*16 + l15, channels wc*64 + i*16 + 4*kg + {0..3}
    constexpr int SP = 80;
    char *stg = smem + wave * 5120;
    const int cw = n0 + wc * 64 + 4 * kg;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 sc = *reinterpret_cast<const float4 *>(p.scale + cw + i * 16);
        const float4 sh = *reinterpret_cast<const float4 *>(p.shift + cw + i * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float q0 = fminf(fmaxf(__builtin_rintf((float)acc[i][j][0] * sc.x + sh.x), -128.f), 127.f);
            const float q1 = fminf(fmaxf(__builtin_rintf((float)acc[i][j][1] * sc.y + sh.y), -128.f), 127.f);
            const float q2 = fminf(fmaxf(__builtin_rintf((float)acc[i][j][2] * sc.z + sh.z), -128.f), 127.f);
            const float q3 = fminf(fmaxf(__builtin_rintf((float)acc[i][j][3] * sc.w + sh.w), -128.f), 127.f);
            const unsigned pk = ((unsigned)(int)q0 & 0xffu) | (((unsigned)(int)q1 & 0xffu) << 8) | (((unsigned)(int)q2 & 0xffu) << 16) |
                                ((unsigned)(int)q3 << 24);
            *reinterpret_cast<unsigned *>(stg + (j * 16 + l15) * SP + i * 16 + 4 * kg) = pk;
        }
    }
    const int s_px = lane >> 2, s_chunk = lane & 3;
    int8_t *dst = reinterpret_cast<int8_t *>(p.dst);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const i32x4 v = *reinterpret_cast<const i32x4 *>(stg + (rr * 16 + s_px) * SP + s_chunk * 16);
        const size_t px = px0 + wp * 64 + rr * 16 + s_px;
        if (px < npx) *reinterpret_cast<i32x4 *>(dst + px * p.dstC + n0 + wc * 64 + s_chunk * 16) = v;
    }
}

}  // namespace

hipError_t quant_i8_launch(const f16 *src, int8_t *dst, size_t n, float inv_scale, float zero_code, hipStream_t stream)
{
    if (n % 8) return hipErrorInvalidValue;
    const size_t n8 = n / 8;
    const int grid = (int)((n8 + 255) / 256 < 4096 ? (n8 + 255) / 256 : 4096);
    hipLaunchKernelGGL(quant_i8_kernel, dim3(grid ? grid : 1), dim3(256), 0, stream, src, dst, n8, inv_scale, zero_code);
    return hipGetLastError();
}

// 1x1 on int8 codes, NHWC, src0 [+ src1] channels multiples of 128, Cout multiple of 128, dst int8 [Hi*Wi][dstC].
hipError_t conv1x1_i8_launch(ConvI8Params p, hipStream_t stream)
{
    if ((p.c0 % CT) || (p.c1 % CT) || p.c0 + p.c1 < CT || (p.Cout % BN) || !p.padline || p.out_f16 || p.mode != ST_NHWC ||
        p.dstC < p.Cout)
        return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_i8_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const size_t npx = (size_t)p.Hi * p.Wi;
    const int grid = (int)((npx + TP - 1) / TP) * (p.Cout / BN);
    hipLaunchKernelGGL(conv1x1_i8_kernel, dim3(grid), dim3(512), SMEM, stream, p);
    return hipGetLastError();
}
