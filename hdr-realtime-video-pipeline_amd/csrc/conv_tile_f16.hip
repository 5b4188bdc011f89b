// conv_tile_f16.hip -- LE's three stride-2 down-convolutions in fp16 (gfx950).
//
// Reference: HDRUNet3T1.down_conv{1,2,3} (3x3, stride 2, pad 1, 32 -> 32, ReLU; HDRUNet3T1_arch.py:170-178).
// 80 B of HBM per input pixel against 288 MAC per output channel: bytes decide.  One workgroup = 4 waves = an 8 x 16
// output tile; its 17 x 33 halo patch (64 B per pixel) is staged once with 16-byte accesses, a wave owns two output rows
// (32 pixels = N of v_mfma_f32_32x32x16_f16), M = the 32 output channels, K = 16 input channels of one tap per MFMA, and
// every wave keeps the layer's whole 18 KiB of weights in 72 VGPRs (its A fragments are the same for all its pixels).
// 36 KiB of LDS: four workgroups per CU overlap each other's loads and MFMAs.  The int8 twin of this kernel (conv_q8.hip) ran these layers at 0.23 ms per frame at 3840x2160 when
// the generic implicit-GEMM kernel needed 0.39 ms, hence this one.
#include "launchers.h"

namespace {

constexpr int TT_TH = 8, TT_TW = 16, TT_CIN = 32, TT_NCH = 4;     // 4 x 16-byte chunks per pixel / weight row
__device__ __forceinline__ int tsw(int r) { return (r >> 2) & 3; }   // rows 256 B apart use different chunk positions

template <int S>
__global__ __launch_bounds__(256, 4) void conv_t16_kernel(ConvParams p)
{
    constexpr int HH = (TT_TH - 1) * S + 3, HWD = (TT_TW - 1) * S + 3, NPX = HH * HWD, ROWB = TT_CIN * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sX = smem;                                           // [NPX][32] f16, chunk-swizzled
    constexpr int NPIECE = (NPX * TT_NCH + 63) / 64;           // the patch as whole 1-KiB LDS-DMA pieces (the last one part-filled)
    float *sS = reinterpret_cast<float *>(smem + NPIECE * 1024); // scale[32], shift[32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tiles_x = (p.Wo + TT_TW - 1) / TT_TW, ntiles = tiles_x * ((p.Ho + TT_TH - 1) / TT_TH);
    // 16 of the 18 A fragments live in registers; the last tap's two sit in LDS (the same for every wave) and are read per tile:
    // with all 72 weight registers live the kernel spills at the 128 VGPRs its four workgroups per CU allow
    f16x8 wv[16];
#pragma unroll
    for (int st = 0; st < 16; ++st) wv[st] = *reinterpret_cast<const f16x8 *>(p.wpk + (size_t)((st >> 1) * 32 + l31) * TT_CIN + ((st & 1) * 2 + lh) * 8);
    f16x8 *sW8 = reinterpret_cast<f16x8 *>(reinterpret_cast<char *>(sS) + 256);
    if (tid < 128) sW8[tid] = *reinterpret_cast<const f16x8 *>(p.wpk + (size_t)(8 * 32 + (tid & 31)) * TT_CIN + ((tid >> 6) * 2 + ((tid >> 5) & 1)) * 8);
    if (tid < 64) sS[tid] = tid < 32 ? p.scale[tid] : p.shift[tid - 32];
    // persistent over tiles: the 72 weight registers are loaded once per workgroup, not once per tile (18 KiB per wave against a
    // 36-KiB halo patch: per-tile workgroups moved twice the payload in weights)
    // The halo patch is staged by LDS-DMA (buffer_load ... lds: no VGPR staging, no address arithmetic per tile beyond one
    // base): piece = 64 consecutive 16-byte LDS slots, slot q holds chunk (q & 3) ^ tsw(q >> 2) of halo pixel q >> 2.  The
    // Out-of-image lanes (zero padding) get an offset beyond the resource and the DMA writes zeros for them.
    constexpr int PPW = (NPIECE + 3) / 4;
    const dma_rsrc_t rsrc = dma_rsrc(p.src0, (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)p.s0_stride * 2u);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * TT_TH, ox0 = tx * TT_TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    if (t != (int)blockIdx.x) __syncthreads();                 // the previous tile's fragment reads are done
    {
        const unsigned base = (unsigned)((iy0 * p.Wi + ix0) * p.s0_stride) * 2u;     // may wrap below zero; in-image lanes land >= 0
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + HH <= p.Hi && ix0 + HWD <= p.Wi;     // workgroup-uniform
        int ln = lane;
        asm volatile("" : "+v"(ln));                           // opaque copy: keeps the nine pieces' lane arithmetic inside the tile loop
#pragma unroll 1                                               // (hoisted or interleaved, its results do not fit beside the 64 weight registers at 128 VGPRs)
        for (int it = 0; it < PPW; ++it) {
            if (wave + 4 * it < NPIECE) {                      // wave-uniform
                const int q = (wave + 4 * it) * 64 + ln, hp = q >> 2;
                const int hy = hp / HWD, hx = hp - hy * HWD;
                const bool ok = hp < NPX && (interior || ((unsigned)(iy0 + hy) < (unsigned)p.Hi && (unsigned)(ix0 + hx) < (unsigned)p.Wi));
                const unsigned off = (unsigned)((hy * p.Wi + hx) * p.s0_stride + (((q & 3) ^ tsw(hp)) << 3)) * 2u;
                dma16(rsrc, sX + (wave + 4 * it) * 1024, ok ? base + off : DMA_OOB);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);                    // vmcnt(0): this wave's pieces have landed
    }
    __syncthreads();
    const int qy = 2 * wave + (l31 >> 4), qx = l31 & 15;
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int hp = (qy * S + tap / 3) * HWD + qx * S + tap % 3;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ch = ks * 2 + lh;
            const f16x8 xv = *reinterpret_cast<const f16x8 *>(sX + hp * ROWB + ((ch ^ tsw(hp)) << 4));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(tap < 8 ? wv[(tap < 8 ? tap : 0) * 2 + ks] : sW8[ks * 64 + lane], xv, acc, 0, 0, 0);
        }
        if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);   // fragment reads at most a kernel row ahead: 72 weight registers are live
    }
    const int oy = oy0 + qy, ox = ox0 + qx;
    if (oy < p.Ho && ox < p.Wo) {
        const float aslope = act_slope(p.act);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = 8 * g + 4 * lh;
            const float4 sc = *reinterpret_cast<const float4 *>(sS + n), sh = *reinterpret_cast<const float4 *>(sS + 32 + n);
            f16x4 o;
            o[0] = (f16)act_fast(acc[4 * g + 0] * sc.x + sh.x, aslope); o[1] = (f16)act_fast(acc[4 * g + 1] * sc.y + sh.y, aslope);
            o[2] = (f16)act_fast(acc[4 * g + 2] * sc.z + sh.z, aslope); o[3] = (f16)act_fast(acc[4 * g + 3] * sc.w + sh.w, aslope);
            *reinterpret_cast<f16x4 *>(p.dst + ((size_t)oy * p.Wo + ox) * p.dstC + n) = o;
        }
    }
    }   // tile loop
}

}  // namespace

// 3x3 / stride 2 / pad 1, 32 -> 32 (CoutPad == 32), weights [9][32][32] f16 (pack_conv with cin_t = 32), NHWC in and out
hipError_t conv_t16_launch(ConvParams p, hipStream_t s, int n_cu)
{
    if (p.c0 != 32 || p.c1 != 0 || p.CoutPad != 32 || (p.s0_stride % 8) || (p.dstC % 4) || p.mode != ST_NHWC || p.res1 || p.res2 ||
        p.Ho != (p.Hi - 1) / 2 + 1 || p.Wo != (p.Wi - 1) / 2 + 1)
        return hipErrorInvalidValue;
    constexpr int NPX = 17 * 33;
    const int smem = ((NPX * 4 + 63) / 64) * 1024 + 256 + 2048;    // halo patch (whole DMA pieces), scale / shift, the last tap's two A fragments
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_t16_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const int ntiles = ((p.Wo + TT_TW - 1) / TT_TW) * ((p.Ho + TT_TH - 1) / TT_TH);
    const int grid = ntiles < 4 * n_cu ? ntiles : 4 * n_cu;       // four workgroups per CU are resident (LDS 36 KiB, <= 128 VGPRs)
    hipLaunchKernelGGL(conv_t16_kernel<2>, dim3(grid), dim3(256), smem, s, p);
    return hipGetLastError();
}
