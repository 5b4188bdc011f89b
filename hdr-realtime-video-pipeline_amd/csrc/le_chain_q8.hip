// le_chain_q8.hip -- the per-pixel layer chains of the fully quantised (W8A8) recipe on v_mfma_i32_32x32x32_i8 (gfx950).
//
// Reference: the same layers as le_fused.hip and agcm.hip's agcm_mlp -- HDRUNet3T1.cond_first + CondNet1
// (HDRUNet3T1_arch.py:41-46), CondNet2's tail (47-49) and ConditionNet's 3 -> 64 -> 64 -> 3 GFM MLP
// (Condition_arch.py:571-583) -- for checkpoints in which every one of them is a W8A8Conv2d
// (hdrtvnet_torch.py:296-364; HR_original_int8_full*.pt: 128 of 128 layers).
//
// A chain stays in registers exactly as in the fp16 kernels, but what travels from layer to layer is the int8 code the
// NEXT layer's quantiser assigns: a 32x32 int32 accumulator tile (rows = this layer's output channels 32mt + 8g + 4lh + k
// in register 4g + k, column = the lane's pixel) is dequantised, activated, re-quantised
//     code = clamp(rint(act(acc * s_x * s_w[row] + shift[row]) / s_x' - z_x' / s_x'), 0, 255) - 128
// with the constants folded on the host (or per frame by agcm_fold for the GFM-modulated AGCM layers) into one
// multiply-add per value, and the 16 codes a lane holds of M-tile mt ARE its 16 bytes of the next layer's K-step mt
// (weights packed in that byte order: hdrtv_api.hip pack_chain_frag).  Zero padding after dequantisation (the 3x3
// first layer of the trunk) is handled as everywhere else: out-of-image pixels are code 0 and the shift constant
// comes from a 16-entry border-class table.
#include "launchers.h"

namespace {

constexpr int T_TH = 8, T_TW = 32;                 // pixels per workgroup of the trunk: 8 rows x 32 columns
constexpr int T_HH = T_TH + 2, T_HW = T_TW + 2, T_PITCH = 40;
constexpr int STG_ROWB = 128 + 16;

__device__ __forceinline__ i32x16 zero16()
{
    i32x16 z;
#pragma unroll
    for (int k = 0; k < 16; ++k) z[k] = 0;
    return z;
}

// acc -> 16 codes (the next layer's K-step bytes): u = acc * A + B, t = max(u, slope * u) + zoff, code = q(t).
// A / B: 16 floats each for this lane half and M-tile.
__device__ __forceinline__ i32x4 requant16(const i32x16 &acc, const float *A, const float *B, float slope, float zoff)
{
    i32x4 o;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 ka = *reinterpret_cast<const float4 *>(A + 4 * g), kb = *reinterpret_cast<const float4 *>(B + 4 * g);
        const float u0 = (float)acc[4 * g + 0] * ka.x + kb.x, u1 = (float)acc[4 * g + 1] * ka.y + kb.y,
                    u2 = (float)acc[4 * g + 2] * ka.z + kb.z, u3 = (float)acc[4 * g + 3] * ka.w + kb.w;
        o[g] = (int)quant4u(fmaxf(u0, slope * u0) + zoff, fmaxf(u1, slope * u1) + zoff, fmaxf(u2, slope * u2) + zoff,
                           fmaxf(u3, slope * u3) + zoff);
    }
    return o;
}
// acc -> real values (no re-quantisation): v = act(acc * A + B)
__device__ __forceinline__ void dequant16(const i32x16 &acc, const float *A, const float *B, float slope, float *v)
{
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 ka = *reinterpret_cast<const float4 *>(A + 4 * g), kb = *reinterpret_cast<const float4 *>(B + 4 * g);
        const float u0 = (float)acc[4 * g + 0] * ka.x + kb.x, u1 = (float)acc[4 * g + 1] * ka.y + kb.y,
                    u2 = (float)acc[4 * g + 2] * ka.z + kb.z, u3 = (float)acc[4 * g + 3] * ka.w + kb.w;
        v[4 * g + 0] = fmaxf(u0, slope * u0); v[4 * g + 1] = fmaxf(u1, slope * u1);
        v[4 * g + 2] = fmaxf(u2, slope * u2); v[4 * g + 3] = fmaxf(u3, slope * u3);
    }
}

// =============================================================================== condition trunk
// constants (floats): [0,64) L1 A [mt][lh][16]; [64,1088) L1 B [cls][mt][lh][16]; then L2..L5 [mt][lh][A16|B16] = 128 each;
// then L6 [lh][A16|B16].  Fragments: L1 mt (2), L2..L5 mt*2+kb (4 each), L6 kb (2) = 20.
constexpr int TQ_NFRAG = 20, TQ_NCONST = 64 + 1024 + 4 * 128 + 64;
struct TrunkQ8Params {
    const f16 *img;
    int H, W;
    const i32x4 *wfrag;
    const float *consts;
    float q1_inv, q1_zoff;         // cond_first.0's quantiser of the image
    float zoff[5];                 // zoff of the quantisers of layers 2..6 (their 1 / x_scale is folded into the constants)
    float q4_inv;                  // CondNet1.0 reads the stored f16 `cond`
    f16 *cond, *cond1;
};

#ifndef TRUNKQ_SUBS
#define TRUNKQ_SUBS 4          // 4-wave groups per workgroup sharing one copy of the fragments and constants (1: TRUNKQ_PER_CU workgroups per CU)
#endif
#ifndef TRUNKQ_PER_CU
#define TRUNKQ_PER_CU 3        // SUBS = 1: resident workgroups per CU (46 KiB of LDS and 120 VGPRs each)
#endif
constexpr int TQ_SUBS = TRUNKQ_SUBS, TQ_NT = 256 * TQ_SUBS;
constexpr int TQ_SIN_B = ((3 * T_HH * T_PITCH + 15) / 16) * 16, TQ_SUB_B = TQ_SIN_B + 4 * 32 * STG_ROWB;
__global__ __launch_bounds__(TQ_NT, TQ_SUBS == 1 ? TRUNKQ_PER_CU : 1) void le_cond_trunk_q8_kernel(TrunkQ8Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    i32x4 *s_w = reinterpret_cast<i32x4 *>(smem);                                        // [TQ_NFRAG][64]
    float *s_c = reinterpret_cast<float *>(smem + TQ_NFRAG * 1024);                       // [TQ_NCONST]
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);               // this thread's 4-wave group: own tile, patch, staging
    unsigned char *s_in = reinterpret_cast<unsigned char *>(s_c + TQ_NCONST) + sub * TQ_SUB_B;   // [3][T_HH][T_PITCH] codes
    char *s_stg = reinterpret_cast<char *>(s_in) + TQ_SIN_B;                              // [4 waves][32][STG_ROWB]

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int H = p.H, W = p.W;
    const int tiles_x = (W + T_TW - 1) / T_TW, ntiles = tiles_x * ((H + T_TH - 1) / T_TH);
    for (int e = threadIdx.x; e < TQ_NFRAG * 64; e += TQ_NT) s_w[e] = p.wfrag[e];
    for (int e = threadIdx.x; e < TQ_NCONST; e += TQ_NT) s_c[e] = p.consts[e];

    constexpr int NE = (3 * T_HH * T_HW + 255) / 256;
    f16 pre[NE];
    bool pre_ok[NE];
    auto fetch = [&](int t) {
        const int ox0 = (t % tiles_x) * T_TW, oy0 = (t / tiles_x) * T_TH;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 256 * i;
            const int c = e / (T_HH * T_HW), r = (e / T_HW) % T_HH, q = e % T_HW;
            const int iy = oy0 - 1 + r, ix = ox0 - 1 + q;
            pre_ok[i] = t < ntiles && e < 3 * T_HH * T_HW && iy >= 0 && iy < H && ix >= 0 && ix < W;
            pre[i] = p.img[pre_ok[i] ? ((size_t)c * H + iy) * W + ix : 0];
        }
    };
    // the groups of a workgroup walk tiles tb + sub in lockstep (shared barriers); a group past the last tile idles through them
    const int tstep = (int)gridDim.x * TQ_SUBS;
    int tb = blockIdx.x * TQ_SUBS;
    if (tb < ntiles) fetch(tb + sub);
    for (; tb < ntiles; tb += tstep) {
        const int t = tb + sub;
        const int ox0 = (t % tiles_x) * T_TW, oy0 = (t / tiles_x) * T_TH;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 256 * i;
            const int c = e / (T_HH * T_HW), r = (e / T_HW) % T_HH, q = e % T_HW;
            if (e < 3 * T_HH * T_HW) {
                const unsigned code = (__builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)pre[i], p.q1_inv, p.q1_zoff), 0, 0u) ^ 0x80u) & 255u;
                s_in[(c * T_HH + r) * T_PITCH + q] = pre_ok[i] ? (unsigned char)code : (unsigned char)0;   // out-of-image: code 0
            }
        }
        __syncthreads();
        if (tb + tstep < ntiles) fetch(tb + tstep + sub);
        if (t >= ntiles) continue;
        char *stg = s_stg + wave * 32 * STG_ROWB;
#pragma unroll 1
        for (int j = 0; j < 2; ++j) {
            const int row = 2 * wave + j, oy = oy0 + row, ox = ox0 + l31;
            const int cls = ((((oy == 0) | ((oy == H - 1) << 1)) << 2) | ((ox == 0) | ((ox == W - 1) << 1))) & 15;
            // ---- layer 1: 3x3 from the 3 code planes, k = (ky * 3 + kx) * 3 + c, 27 of 32 used; this lane: k = 16 lh + e
            i32x4 b1 = {0, 0, 0, 0};
            {
                const int base = row * T_PITCH + l31;
                unsigned w[4] = {0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k0 = e, k1 = 16 + e;      // lane half 0 / 1
                    const int o0 = ((k0 % 3) * T_HH + (k0 / 3) / 3) * T_PITCH + (k0 / 3) % 3;
                    const int o1 = k1 < 27 ? ((k1 % 3) * T_HH + (k1 / 3) / 3) * T_PITCH + (k1 / 3) % 3 : 0;
                    unsigned v = s_in[base + (lh ? o1 : o0)];
                    if (lh && k1 >= 27) v = 0;
                    w[e >> 2] |= v << (8 * (e & 3));
                }
                b1[0] = (int)w[0]; b1[1] = (int)w[1]; b1[2] = (int)w[2]; b1[3] = (int)w[3];
            }
            i32x4 b[2];
            {
                const float *A = s_c + lh * 16, *B = s_c + 64 + cls * 64 + lh * 16;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const i32x16 a = __builtin_amdgcn_mfma_i32_32x32x32_i8(s_w[mt * 64 + lane], b1, zero16(), 0, 0, 0);
                    b[mt] = requant16(a, A + mt * 32, B + mt * 32, 0.1f, p.zoff[0]);
                }
            }
            // ---- layers 2..5: 64 -> 64, LeakyReLU(0.1); layer 3's output is `cond` (stored f16, re-read by four layers)
#pragma unroll
            for (int layer = 2; layer <= 5; ++layer) {
                const i32x4 *wf = s_w + (2 + (layer - 2) * 4) * 64;
                const float *K = s_c + 1088 + (layer - 2) * 128 + lh * 32;
                i32x4 nb[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    i32x16 a = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[(mt * 2 + 0) * 64 + lane], b[0], zero16(), 0, 0, 0);
                    a = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[(mt * 2 + 1) * 64 + lane], b[1], a, 0, 0, 0);
                    if (layer == 3) {
                        float v[16];
                        dequant16(a, K + mt * 64, K + mt * 64 + 16, 0.1f, v);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            f16x4 o;
                            o[0] = (f16)v[4 * g + 0]; o[1] = (f16)v[4 * g + 1]; o[2] = (f16)v[4 * g + 2]; o[3] = (f16)v[4 * g + 3];
                            *reinterpret_cast<f16x4 *>(stg + l31 * STG_ROWB + (32 * mt + 8 * g + 4 * lh) * 2) = o;
                            nb[mt][g] = (int)quant4((float)o[0], (float)o[1], (float)o[2], (float)o[3], p.q4_inv, p.zoff[2]);
                        }
                    } else {
                        nb[mt] = requant16(a, K + mt * 64, K + mt * 64 + 16, 0.1f, p.zoff[layer - 1]);
                    }
                }
                b[0] = nb[0]; b[1] = nb[1];
                if (layer == 3) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    if (oy < H) {
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int e = lane + 64 * it, px = e >> 3, c8 = e & 7;
                            if (ox0 + px < W)
                                *reinterpret_cast<f16x8 *>(p.cond + ((size_t)oy * W + ox0 + px) * 64 + c8 * 8) =
                                    *reinterpret_cast<const f16x8 *>(stg + px * STG_ROWB + c8 * 16);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            // ---- layer 6: 64 -> 16, no activation -> cond1 (NHWC 16)
            {
                const float *K = s_c + 1600 + lh * 32;
                i32x16 a = __builtin_amdgcn_mfma_i32_32x32x32_i8(s_w[18 * 64 + lane], b[0], zero16(), 0, 0, 0);
                a = __builtin_amdgcn_mfma_i32_32x32x32_i8(s_w[19 * 64 + lane], b[1], a, 0, 0, 0);
                float v[16];
                dequant16(a, K, K + 16, 1.f, v);
                f16x4 lo, hi;
#pragma unroll
                for (int k = 0; k < 4; ++k) { lo[k] = (f16)v[k]; hi[k] = (f16)v[4 + k]; }
                *reinterpret_cast<f16x4 *>(stg + l31 * STG_ROWB + (4 * lh) * 2) = lo;
                *reinterpret_cast<f16x4 *>(stg + l31 * STG_ROWB + (8 + 4 * lh) * 2) = hi;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const int px = lane >> 1, c8 = lane & 1;
                if (oy < H && ox0 + px < W)
                    *reinterpret_cast<f16x8 *>(p.cond1 + ((size_t)oy * W + ox0 + px) * 16 + c8 * 8) =
                        *reinterpret_cast<const f16x8 *>(stg + px * STG_ROWB + c8 * 16);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
}

constexpr int TRUNK_Q8_SMEM = TQ_NFRAG * 1024 + TQ_NCONST * 4 + TQ_SUBS * TQ_SUB_B;

// ================================================================================== CondNet2 tail
// x: int8 codes of CondNet2.2's quantiser, NHWC 64.  Fragments: layer 1 mt*2+kb (4, natural byte order), layer 2 kb (2).
// constants: layer 1 [mt][lh][A16|B16] (128), layer 2 [lh][A16|B16] (64).
struct TailQ8Params {
    const int8_t *x;
    size_t npx;
    const i32x4 *wfrag;
    const float *consts;
    float zoff2;
    f16 *out;
};

__global__ __launch_bounds__(256) void cond_tail_q8_kernel(TailQ8Params p)
{
    __shared__ __attribute__((aligned(16))) float s_c[192];
    for (int e = threadIdx.x; e < 192; e += 256) s_c[e] = p.consts[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    i32x4 w1[4], w2[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) w1[i] = p.wfrag[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 2; ++i) w2[i] = p.wfrag[(4 + i) * 64 + lane];
    const size_t ngroups = (p.npx + 31) / 32, gstep = (size_t)gridDim.x * 4;
    for (size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); g < ngroups; g += gstep) {
        size_t px = g * 32 + l31;
        const bool ok = px < p.npx;
        if (!ok) px = p.npx - 1;
        const i32x4 b0 = *reinterpret_cast<const i32x4 *>(p.x + px * 64 + 16 * lh);
        const i32x4 b1 = *reinterpret_cast<const i32x4 *>(p.x + px * 64 + 32 + 16 * lh);
        i32x4 nb[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            i32x16 a = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1[mt * 2 + 0], b0, zero16(), 0, 0, 0);
            a = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1[mt * 2 + 1], b1, a, 0, 0, 0);
            nb[mt] = requant16(a, s_c + mt * 64 + lh * 32, s_c + mt * 64 + lh * 32 + 16, 0.1f, p.zoff2);
        }
        i32x16 a = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2[0], nb[0], zero16(), 0, 0, 0);
        a = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2[1], nb[1], a, 0, 0, 0);
        float v[16];
        dequant16(a, s_c + 128 + lh * 32, s_c + 128 + lh * 32 + 16, 1.f, v);
        if (ok) {
            f16x4 lo, hi;
#pragma unroll
            for (int k = 0; k < 4; ++k) { lo[k] = (f16)v[k]; hi[k] = (f16)v[4 + k]; }
            *reinterpret_cast<f16x4 *>(p.out + px * 16 + 4 * lh) = lo;
            *reinterpret_cast<f16x4 *>(p.out + px * 16 + 8 + 4 * lh) = hi;
        }
    }
}

// ====================================================================================== AGCM MLP
// Fragments (static): layer 1 mt (2; byte c of lane half 0 = colour channel c), layer 2 mt*2+kb (4), layer 3 kb (2).
// Per-frame constants (agcm_fold_q8): layer 1 [mt][lh][A16|B16] (128), layer 2 (128), layer 3 [lh][A16|B16] (64).
struct AgcmQ8Params {
    const f16 *in;
    f16 *out;
    size_t npix;
    const i32x4 *wfrag;
    const float *consts;
    float q1_inv, q1_zoff, zoff2, zoff3;
};

__global__ __launch_bounds__(256) void agcm_mlp_q8_kernel(AgcmQ8Params p)
{
    __shared__ __attribute__((aligned(16))) float s_c[320];
    for (int e = threadIdx.x; e < 320; e += 256) s_c[e] = p.consts[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    i32x4 w1[2], w2[4], w3[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) w1[i] = p.wfrag[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) w2[i] = p.wfrag[(2 + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 2; ++i) w3[i] = p.wfrag[(6 + i) * 64 + lane];
    const size_t ngrp = (p.npix + 31) / 32;
    const size_t wave_id = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
    // a group's three input values are fetched one trip ahead (as agcm_mlp_kernel: nothing else hides the load latency in front
    // of the dependent MFMA chain)
    f16 nx[3] = {(f16)0.f, (f16)0.f, (f16)0.f};
    auto fetch = [&](size_t g) {
        const size_t pix = g * 32 + l31;
        const size_t o = (g < ngrp && pix < p.npix && lh == 0) ? pix : 0;     // masked lanes read pixel 0 and drop it
        nx[0] = p.in[o]; nx[1] = p.in[p.npix + o]; nx[2] = p.in[2 * p.npix + o];
    };
    fetch(wave_id);
    for (size_t g = wave_id; g < ngrp; g += nwave) {
        const size_t pix = g * 32 + l31;
        const bool ok = pix < p.npix;
        i32x4 x = {0, 0, 0, 0};
        if (ok && lh == 0)
            x[0] = (int)(quant4((float)nx[0], (float)nx[1], (float)nx[2], 0.f, p.q1_inv, p.q1_zoff) & 0x00ffffffu);
        fetch(g + nwave);
        i32x4 b[2], c[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const i32x16 a = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1[mt], x, zero16(), 0, 0, 0);
            b[mt] = requant16(a, s_c + mt * 64 + lh * 32, s_c + mt * 64 + lh * 32 + 16, 0.f, p.zoff2);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            i32x16 a = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2[mt * 2 + 0], b[0], zero16(), 0, 0, 0);
            a = __builtin_amdgcn_mfma_i32_32x32x32_i8(w2[mt * 2 + 1], b[1], a, 0, 0, 0);
            c[mt] = requant16(a, s_c + 128 + mt * 64 + lh * 32, s_c + 128 + mt * 64 + lh * 32 + 16, 0.f, p.zoff3);
        }
        i32x16 a = __builtin_amdgcn_mfma_i32_32x32x32_i8(w3[0], c[0], zero16(), 0, 0, 0);
        a = __builtin_amdgcn_mfma_i32_32x32x32_i8(w3[1], c[1], a, 0, 0, 0);
        if (ok && lh == 0) {
            const float *K = s_c + 256;      // lane half 0: rows 0..3 in registers 0..3
            p.out[pix] = (f16)((float)a[0] * K[0] + K[16]);
            p.out[p.npix + pix] = (f16)((float)a[1] * K[1] + K[17]);
            p.out[2 * p.npix + pix] = (f16)((float)a[2] * K[2] + K[18]);
        }
    }
}

// planar f16 [3][H][W] -> NHWC int8 codes [H][W][32] (3 real + 29 zero bytes): input of the 3x3 conv_first as a conv_q8 layer
__global__ __launch_bounds__(256) void planar3_to_q8_kernel(const f16 *__restrict__ in, size_t npix, float inv, float zoff,
                                                            int8_t *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) {
        i32x4 v = {0, 0, 0, 0};
        v[0] = (int)(quant4((float)in[i], (float)in[npix + i], (float)in[2 * npix + i], 0.f, inv, zoff) & 0x00ffffffu);
        *reinterpret_cast<i32x4 *>(out + i * 32) = v;
        *reinterpret_cast<i32x4 *>(out + i * 32 + 16) = i32x4{0, 0, 0, 0};
    }
}

}  // namespace

hipError_t le_cond_trunk_q8_launch(const f16 *img, int H, int W, const TrunkQ8Args &a, f16 *cond, f16 *cond1, int n_cu, hipStream_t s)
{
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(le_cond_trunk_q8_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, TRUNK_Q8_SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    TrunkQ8Params p;
    p.img = img; p.H = H; p.W = W; p.wfrag = reinterpret_cast<const i32x4 *>(a.wfrag); p.consts = a.consts;
    p.q1_inv = a.q1_inv; p.q1_zoff = a.q1_zoff; p.q4_inv = a.q4_inv;
    for (int i = 0; i < 5; ++i) p.zoff[i] = a.zoff[i];
    p.cond = cond; p.cond1 = cond1;
    const int ntiles = ((W + T_TW - 1) / T_TW) * ((H + T_TH - 1) / T_TH);
    const int per_cu = TQ_SUBS == 1 ? TRUNKQ_PER_CU : 1, want = (ntiles + TQ_SUBS - 1) / TQ_SUBS;
    const int grid = want < per_cu * n_cu ? want : per_cu * n_cu;
    hipLaunchKernelGGL(le_cond_trunk_q8_kernel, dim3(grid), dim3(TQ_NT), TRUNK_Q8_SMEM, s, p);
    return hipGetLastError();
}

hipError_t cond_tail_q8_launch(const int8_t *x, size_t npx, const int8_t *wfrag, const float *consts, float zoff2, f16 *out, int n_cu,
                               hipStream_t s)
{
    if (!npx) return hipErrorInvalidValue;
    TailQ8Params p{x, npx, reinterpret_cast<const i32x4 *>(wfrag), consts, zoff2, out};
    size_t grid = ((npx + 31) / 32 + 3) / 4;
    if (grid > (size_t)8 * n_cu) grid = (size_t)8 * n_cu;
    hipLaunchKernelGGL(cond_tail_q8_kernel, dim3((unsigned)grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t agcm_mlp_q8_launch(const f16 *in, f16 *out, size_t npix, const int8_t *wfrag, const float *consts, float q1_inv, float q1_zoff,
                              float zoff2, float zoff3, hipStream_t s)
{
    AgcmQ8Params p{in, out, npix, reinterpret_cast<const i32x4 *>(wfrag), consts, q1_inv, q1_zoff, zoff2, zoff3};
    size_t blocks = ((npix + 31) / 32 + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(agcm_mlp_q8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t planar3_to_q8_launch(const f16 *in, size_t npix, float inv, float zoff, int8_t *out, hipStream_t s)
{
    size_t blocks = (npix + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(planar3_to_q8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, npix, inv, zoff, out);
    return hipGetLastError();
}
