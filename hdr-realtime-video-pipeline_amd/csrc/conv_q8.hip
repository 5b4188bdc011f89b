// conv_q8.hip -- W8A8 convolution of the LE condition nets and down-convs on v_mfma_i32_32x32x32_i8 (gfx950).
//
// Reference: W8A8Conv2d.forward (hdrtvnet_torch.py:351-364) for the layers of HDRUNet3T1 that are not 3x3 / stride 1 /
// 32 channels (those run in conv32p.hip): down_conv{1,2,3} (3x3, stride 2, 32 -> 32), CondNet3.0 / 4.0 / 3.2 / 4.2
// (3x3, stride 2, 64 -> 64), CondNet4.4 (3x3, stride 2, 64 -> 16) and CondNet3.4 (1x1, 64 -> 16);
// HDRUNet3T1_arch.py:47-55, 170-178.
//
// The layer's input is either the f16 NHWC tensor its producer wrote -- it is quantised while the halo tile is staged,
// q = clamp(rint((x - x_zero) / x_scale), 0, 255), LDS holds the codes c = q - 128 -- or already the int8 codes of this
// layer's quantiser (a W8A8 layer whose only reader is another W8A8 layer stores them directly: half the bytes).
// The reference pads with zeros AFTER dequantisation and x_zero is a float, so there is no code for "0.0": out-of-image
// halo pixels get code 0, which adds nothing to the integer sum, and the epilogue adds the exact constant
//     w_scale[n] * (128 * x_scale + x_zero) * sum(w_int8[n] over the taps that are INSIDE the image)
// from a table of the 16 border classes (first / last row x first / last column; hdrtv_api.hip pack_conv_q8).
// One workgroup = 4 waves = an 8 x 16 output tile; a wave owns two rows (32 pixels = the N of a 32x32x32 MFMA) and
// walks the output channels in blocks of 32 (M); K = one 32-channel slice of one tap per MFMA.
#include "launchers.h"

namespace {


constexpr int Q8_TH = 8, Q8_TW = 16;

// chunk swizzle of a row of NCH 16-byte chunks: rows 256 B apart in LDS use different chunk positions
template <int NCH> __device__ __forceinline__ int rsw(int r) { return (r / (16 / NCH)) & (NCH - 1); }

template <int CIN, int KS, int S>
__global__ __launch_bounds__(256) void conv_q8_kernel(ConvQ8Params p)
{
    constexpr int NCH = CIN / 16;                              // 16-byte chunks per pixel / weight row
    constexpr int HH = (Q8_TH - 1) * S + KS, HWD = (Q8_TW - 1) * S + KS, NPX = HH * HWD;
    constexpr int PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sX = smem;                                           // [NPX][CIN] int8 codes, chunk-swizzled
    char *sW = smem + ((NPX * CIN + 255) & ~255);              // [KS*KS][CoutPad][CIN]
    float *sS = reinterpret_cast<float *>(sW + KS * KS * p.CoutPad * CIN);   // scale[CoutPad], shift[16][CoutPad]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tiles_x = (p.Wo + Q8_TW - 1) / Q8_TW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int oy0 = ty * Q8_TH, ox0 = tx * Q8_TW;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;

    // ---- weights, scale and the border-class shifts
    const int wrows = KS * KS * p.CoutPad;
    for (int e = tid; e < wrows * NCH; e += 256) {
        const int r = e / NCH, ch = e - r * NCH;
        *reinterpret_cast<i32x4 *>(sW + r * CIN + ((ch ^ rsw<NCH>(r)) << 4)) =
            *reinterpret_cast<const i32x4 *>(p.wpk8 + (size_t)r * CIN + ch * 16);
    }
    for (int e = tid; e < 17 * p.CoutPad; e += 256) sS[e] = e < p.CoutPad ? p.scale[e] : p.shift[e - p.CoutPad];

    // ---- halo tile: quantise on load (f16 source) or copy (int8 source); out-of-image pixels are code 0
    for (int e = tid; e < NPX * NCH; e += 256) {
        const int hp = e / NCH, ch = e - hp * NCH;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int iy = iy0 + hy, ix = ix0 + hx;
        i32x4 v = {0, 0, 0, 0};
        if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) {
            const size_t pix = (size_t)iy * p.Wi + ix;
            if (p.src_i8) {
                v = *reinterpret_cast<const i32x4 *>(reinterpret_cast<const int8_t *>(p.src) + pix * p.src_stride + ch * 16);
            } else {
                const f16 *g = reinterpret_cast<const f16 *>(p.src) + pix * p.src_stride + ch * 16;
                const f16x8 a = *reinterpret_cast<const f16x8 *>(g), b = *reinterpret_cast<const f16x8 *>(g + 8);
                v[0] = (int)quant4((float)a[0], (float)a[1], (float)a[2], (float)a[3], p.q_inv, p.q_zoff);
                v[1] = (int)quant4((float)a[4], (float)a[5], (float)a[6], (float)a[7], p.q_inv, p.q_zoff);
                v[2] = (int)quant4((float)b[0], (float)b[1], (float)b[2], (float)b[3], p.q_inv, p.q_zoff);
                v[3] = (int)quant4((float)b[4], (float)b[5], (float)b[6], (float)b[7], p.q_inv, p.q_zoff);
            }
        }
        *reinterpret_cast<i32x4 *>(sX + hp * CIN + ((ch ^ rsw<NCH>(hp)) << 4)) = v;
    }
    __syncthreads();

    // ---- this lane's output pixel and its border class
    const int qy = 2 * wave + (l31 >> 4), qx = l31 & 15;
    const int oy = oy0 + qy, ox = ox0 + qx;
    int bcls = 0;
    if (KS == 3) {
        const int ty0 = oy * S - PAD, tx0 = ox * S - PAD;     // first tap's input row / column
        bcls = ((((ty0 < 0) | ((ty0 + 2 >= p.Hi) << 1)) << 2) | ((tx0 < 0) | ((tx0 + 2 >= p.Wi) << 1))) & 15;
    }
    const bool live = oy < p.Ho && ox < p.Wo;
    const float aslope = act_slope(p.act);

    for (int pass = 0; pass < p.CoutPad / 32; ++pass) {
        i32x16 acc;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0;
#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int hp = (qy * S + tap / KS) * HWD + qx * S + tap % KS;
            const int wr = tap * p.CoutPad + pass * 32 + l31;
#pragma unroll
            for (int kc = 0; kc < CIN / 32; ++kc) {
                const int ch = kc * 2 + lh;
                const i32x4 wv = *reinterpret_cast<const i32x4 *>(sW + wr * CIN + ((ch ^ rsw<NCH>(wr)) << 4));
                const i32x4 xv = *reinterpret_cast<const i32x4 *>(sX + hp * CIN + ((ch ^ rsw<NCH>(hp)) << 4));
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wv, xv, acc, 0, 0, 0);
            }
        }
        // ---- epilogue: accumulator rows 8g + 4lh + k of this 32-channel block, column = this lane's pixel
        if (live) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = pass * 32 + 8 * g + 4 * lh;
                if (n < p.Cout) {
                    const float4 sc = *reinterpret_cast<const float4 *>(sS + n);
                    const float4 sh = *reinterpret_cast<const float4 *>(sS + p.CoutPad + bcls * p.CoutPad + n);
                    const float v0 = act_fast((float)acc[4 * g + 0] * sc.x + sh.x, aslope), v1 = act_fast((float)acc[4 * g + 1] * sc.y + sh.y, aslope),
                                v2 = act_fast((float)acc[4 * g + 2] * sc.z + sh.z, aslope), v3 = act_fast((float)acc[4 * g + 3] * sc.w + sh.w, aslope);
                    const size_t o = ((size_t)oy * p.Wo + ox) * p.dstC + n;
                    if (p.dst_i8) {
                        // the reader's quantiser sees the f16 tensor the reference's fp16 graph would hold
                        *reinterpret_cast<unsigned *>(reinterpret_cast<int8_t *>(p.dst) + o) =
                            quant4((float)(f16)v0, (float)(f16)v1, (float)(f16)v2, (float)(f16)v3, p.oq_inv, p.oq_zoff);
                    } else {
                        f16x4 ov;
                        ov[0] = (f16)v0; ov[1] = (f16)v1; ov[2] = (f16)v2; ov[3] = (f16)v3;
                        *reinterpret_cast<f16x4 *>(reinterpret_cast<f16 *>(p.dst) + o) = ov;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Up to three 3x3 / stride-2 / 64 -> 64 W8A8 layers that read ONE f16 tensor through their own quantisers (CondNet2.0,
// CondNet3.0, CondNet4.0 all read the full-resolution condition map, HDRUNet3T1_arch.py:47-55): the 17 x 33 halo patch is
// fetched from HBM once, quantised in registers with each layer's (x_scale, x_zero) into that layer's int8 tile, and the
// layers' convolutions run back to back from LDS.  8 waves: wave = (output-channel half, 32-pixel group) of an 8 x 16 tile.
template <int NG>
__global__ __launch_bounds__(512) void conv_q8_multi_kernel(ConvQ8MultiParams p)
{
    constexpr int CIN = 64, NCH = 4, S = 2, KS = 3;
    constexpr int HH = (Q8_TH - 1) * S + KS, HWD = (Q8_TW - 1) * S + KS, NPX = HH * HWD;     // 17 x 33
    constexpr int XB = (NPX * CIN + 255) & ~255, WB = KS * KS * 64 * CIN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sX = smem;                           // [NG][NPX][64] codes
    char *sW = smem + NG * XB;                 // [9][64][64], one layer at a time
    float *sS = reinterpret_cast<float *>(sW + WB);   // scale[64] + shift[16][64] of the current layer

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int pg = wave & 3, ch_half = wave >> 2;
    const int tiles_x = (p.Wo + Q8_TW - 1) / Q8_TW, ntiles = tiles_x * ((p.Ho + Q8_TH - 1) / Q8_TH);
    const int qy = 2 * pg + (l31 >> 4), qx = l31 & 15;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int oy0 = ty * Q8_TH, ox0 = tx * Q8_TW;
        const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        __syncthreads();                       // the previous tile's last layer is done with sX
        for (int e = tid; e < NPX * NCH; e += 512) {
            const int hp = e / NCH, ch = e - hp * NCH;
            const int hy = hp / HWD, hx = hp - hy * HWD;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool in = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            f16x8 a, b;
            if (in) {
                const f16 *g = p.src + ((size_t)iy * p.Wi + ix) * p.src_stride + ch * 16;
                a = *reinterpret_cast<const f16x8 *>(g);
                b = *reinterpret_cast<const f16x8 *>(g + 8);
            }
            const int off = hp * CIN + ((ch ^ rsw<NCH>(hp)) << 4);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                i32x4 v = {0, 0, 0, 0};
                if (in) {
                    const float inv = p.g[g].q_inv, zo = p.g[g].q_zoff;
                    v[0] = (int)quant4((float)a[0], (float)a[1], (float)a[2], (float)a[3], inv, zo);
                    v[1] = (int)quant4((float)a[4], (float)a[5], (float)a[6], (float)a[7], inv, zo);
                    v[2] = (int)quant4((float)b[0], (float)b[1], (float)b[2], (float)b[3], inv, zo);
                    v[3] = (int)quant4((float)b[4], (float)b[5], (float)b[6], (float)b[7], inv, zo);
                }
                *reinterpret_cast<i32x4 *>(sX + g * XB + off) = v;
            }
        }
        const int oy = oy0 + qy, ox = ox0 + qx;
        const int ty0 = oy * S - 1, tx0 = ox * S - 1;
        const int bcls = ((((ty0 < 0) | ((ty0 + 2 >= p.Hi) << 1)) << 2) | ((tx0 < 0) | ((tx0 + 2 >= p.Wi) << 1))) & 15;
        const bool live = oy < p.Ho && ox < p.Wo;
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const ConvQ8Group &G = p.g[g];
            __syncthreads();                   // tile staged (g = 0) / previous layer done with sW and sS
            for (int e = tid; e < KS * KS * 64 * NCH; e += 512) {
                const int r = e / NCH, ch = e - r * NCH;
                *reinterpret_cast<i32x4 *>(sW + r * CIN + ((ch ^ rsw<NCH>(r)) << 4)) =
                    *reinterpret_cast<const i32x4 *>(G.wpk8 + (size_t)r * CIN + ch * 16);
            }
            for (int e = tid; e < 17 * 64; e += 512) sS[e] = e < 64 ? G.scale[e] : G.shift[e - 64];
            __syncthreads();
            const char *x = sX + g * XB;
            i32x16 acc;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hp = (qy * S + tap / 3) * HWD + qx * S + tap % 3;
                const int wr = tap * 64 + ch_half * 32 + l31;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const int ch = kc * 2 + lh;
                    const i32x4 wv = *reinterpret_cast<const i32x4 *>(sW + wr * CIN + ((ch ^ rsw<NCH>(wr)) << 4));
                    const i32x4 xv = *reinterpret_cast<const i32x4 *>(x + hp * CIN + ((ch ^ rsw<NCH>(hp)) << 4));
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wv, xv, acc, 0, 0, 0);
                }
            }
            if (live) {
                const float aslope = act_slope(G.act);
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    const int n = ch_half * 32 + 8 * gg + 4 * lh;
                    const float4 sc = *reinterpret_cast<const float4 *>(sS + n);
                    const float4 sh = *reinterpret_cast<const float4 *>(sS + 64 + bcls * 64 + n);
                    const float v0 = act_fast((float)acc[4 * gg + 0] * sc.x + sh.x, aslope), v1 = act_fast((float)acc[4 * gg + 1] * sc.y + sh.y, aslope),
                                v2 = act_fast((float)acc[4 * gg + 2] * sc.z + sh.z, aslope), v3 = act_fast((float)acc[4 * gg + 3] * sc.w + sh.w, aslope);
                    const size_t o = ((size_t)oy * p.Wo + ox) * 64 + n;
                    if (G.dst_i8) {
                        *reinterpret_cast<unsigned *>(reinterpret_cast<int8_t *>(G.dst) + o) =
                            quant4((float)(f16)v0, (float)(f16)v1, (float)(f16)v2, (float)(f16)v3, G.oq_inv, G.oq_zoff);
                    } else {
                        f16x4 ov;
                        ov[0] = (f16)v0; ov[1] = (f16)v1; ov[2] = (f16)v2; ov[3] = (f16)v3;
                        *reinterpret_cast<f16x4 *>(reinterpret_cast<f16 *>(G.dst) + o) = ov;
                    }
                }
            }
        }
    }
}

template <int CIN, int KS, int S>
hipError_t launch_q8(const ConvQ8Params &p, hipStream_t s)
{
    constexpr int HH = (Q8_TH - 1) * S + KS, HWD = (Q8_TW - 1) * S + KS, NPX = HH * HWD;
    const int smem = ((NPX * CIN + 255) & ~255) + KS * KS * p.CoutPad * CIN + 17 * p.CoutPad * 4;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv_q8_kernel<CIN, KS, S>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const int grid = ((p.Wo + Q8_TW - 1) / Q8_TW) * ((p.Ho + Q8_TH - 1) / Q8_TH);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, p);
    return hipGetLastError();
}

}  // namespace

template <int NG>
hipError_t launch_multi(const ConvQ8MultiParams &p, int n_cu, hipStream_t s)
{
    constexpr int NPX = 17 * 33, XB = (NPX * 64 + 255) & ~255;
    constexpr int smem = NG * XB + 9 * 64 * 64 + 17 * 64 * 4;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv_q8_multi_kernel<NG>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const int ntiles = ((p.Wo + Q8_TW - 1) / Q8_TW) * ((p.Ho + Q8_TH - 1) / Q8_TH);
    const int per_cu = (160 * 1024) / smem;
    const int cap = n_cu * (per_cu < 1 ? 1 : per_cu);
    hipLaunchKernelGGL(kern, dim3(ntiles < cap ? ntiles : cap), dim3(512), smem, s, p);
    return hipGetLastError();
}

// 1..3 layers (3x3, stride 2, 64 -> 64) over one f16 NHWC source, each with its own quantiser, weights and destination
hipError_t conv_q8_multi_launch(ConvQ8MultiParams p, int n_cu, hipStream_t s)
{
    if (p.ngroups < 1 || p.ngroups > 3 || (p.src_stride % 16) || p.Ho != (p.Hi - 1) / 2 + 1 || p.Wo != (p.Wi - 1) / 2 + 1)
        return hipErrorInvalidValue;
    for (int g = 0; g < p.ngroups; ++g)
        if (!p.g[g].wpk8 || !p.g[g].scale || !p.g[g].shift || !p.g[g].dst) return hipErrorInvalidValue;
    return p.ngroups == 1 ? launch_multi<1>(p, n_cu, s) : (p.ngroups == 2 ? launch_multi<2>(p, n_cu, s) : launch_multi<3>(p, n_cu, s));
}

// Cin in {32, 64}; (ks, stride) in {(3,2), (3,1), (1,1)}; CoutPad a multiple of 32, Cout a multiple of 4; src_stride and dstC
// multiples of 16 / 4 elements so that every access is aligned.  hipErrorInvalidValue otherwise.
hipError_t conv_q8_launch(ConvQ8Params p, hipStream_t s)
{
    if ((p.CoutPad % 32) || (p.Cout % 4) || p.Cout > p.CoutPad || (p.src_stride % 16) || (p.dstC % 4) || !p.wpk8 || !p.scale || !p.shift)
        return hipErrorInvalidValue;
    const int pad = p.ks / 2;
    if (p.Ho != (p.Hi + 2 * pad - p.ks) / p.stride + 1 || p.Wo != (p.Wi + 2 * pad - p.ks) / p.stride + 1) return hipErrorInvalidValue;
    if (p.Cin == 32 && p.ks == 3 && p.stride == 2) return launch_q8<32, 3, 2>(p, s);
    if (p.Cin == 64 && p.ks == 3 && p.stride == 2) return launch_q8<64, 3, 2>(p, s);
    if (p.Cin == 32 && p.ks == 3 && p.stride == 1) return launch_q8<32, 3, 1>(p, s);
    if (p.Cin == 64 && p.ks == 3 && p.stride == 1) return launch_q8<64, 3, 1>(p, s);
    if (p.Cin == 64 && p.ks == 1 && p.stride == 1) return launch_q8<64, 1, 1>(p, s);
    if (p.Cin == 32 && p.ks == 1 && p.stride == 1) return launch_q8<32, 1, 1>(p, s);
    return hipErrorInvalidValue;
}
