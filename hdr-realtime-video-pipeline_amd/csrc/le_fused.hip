// le_fused.hip -- LE condition trunk as ONE kernel (gfx950).
//
// Reference: HDRUNet3T1.cond_first (conv3x3 3->64, 1x1 64->64, 1x1 64->64, LeakyReLU(0.1) after
// each) followed by CondNet1 (1x1 64->64, 1x1 64->64, LeakyReLU(0.1), 1x1 64->16),
// HDRUNet3T1_arch.py:41-46, 160-161.  Unfused these six layers stream a 64-channel full-resolution
// tensor (1.06 GB at 4K) through HBM eleven times; here the whole chain runs per pixel in
// registers: each layer's 32x32 fp32 MFMA accumulator tile is activated, packed to f16 and used
// directly as the next layer's B operand (weights K-permuted at pack time, common.h
// acc_kperm16).  HBM traffic per pixel: 6 B in, 128 B (cond) + 32 B (cond1) out.
#include "launchers.h"

namespace {

constexpr int T_TH = 8, T_TW = 32;                 // pixels per workgroup: 8 rows x 32 cols
constexpr int T_HH = T_TH + 2, T_HW = T_TW + 2;
constexpr int NFRAG = 40;                          // 4 (L1) + 4x8 (L2..L5) + 4 (L6)
constexpr int NBIAS = 64 * 5 + 32;
constexpr int STG_ROWB = 128 + 16;                 // per-wave output staging row (64 ch f16 + pad)
#ifndef TRUNK_SUBS
#define TRUNK_SUBS 3                               // 4-wave groups per workgroup sharing one copy of the weight fragments (1: two workgroups per CU)
#endif
constexpr int SUBS = TRUNK_SUBS, TRUNK_NT = 256 * SUBS;
constexpr int SIN_B = ((3 * T_HH * (T_HW + 2) * 2 + 15) / 16) * 16, SUB_B = SIN_B + 4 * 32 * STG_ROWB;

__device__ __forceinline__ f32x16 bias_tile_l(const float *b, int lh)
{
    f32x16 a;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4 *>(b + 8 * g + 4 * lh);
        a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
    }
    return a;
}

// LeakyReLU(0.1) then f16 pack of accumulator registers 8s..8s+7 (= next layer's k-step fragment)
__device__ __forceinline__ f16x8 lrelu_pack(const f32x16 &a, int s)
{
    // round to f16 first, LeakyReLU in packed f16 (the reference's fp16 GPU graph does exactly this: the conv
    // result is an f16 tensor before nn.LeakyReLU sees it).  In fp32 the max costs 3.5 VALU ops per value
    // (hipcc canonicalises the MFMA result with an extra v_max); packed it is 1.5.
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)a[8 * s + j];
    return __builtin_elementwise_max(o, o * (f16)0.1f);
}

// Last layer of a chain as a W8A8 layer (CondNet1.4 / CondNet2.4 in the reference's mixed and full INT8 recipes): the 64
// LeakyReLU'd inputs of a pixel sit in four packed f16 fragments (fragment s, element e = channel 16s + 4lh + e for e < 4,
// 16s + 8 + 4lh + e - 4 otherwise); they are quantised in registers and consumed by two v_mfma_i32_32x32x32_i8 whose
// weight fragments the host packed in the same byte order (hdrtv_api.hip pack_q_last).  ss = scale[32] | shift[32].
struct QLast {
    const i32x4 *wq;       // [2][64 lanes]
    const float *ss;
    float q_inv, q_zoff;
};
__device__ __forceinline__ f32x16 qlast_apply(const QLast &q, const f16x8 *bf, int lane, int lh)
{
    i32x16 o;
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const f16x8 a = bf[2 * m], b = bf[2 * m + 1];
        i32x4 x;
        x[0] = (int)quant4((float)a[0], (float)a[1], (float)a[2], (float)a[3], q.q_inv, q.q_zoff);
        x[1] = (int)quant4((float)a[4], (float)a[5], (float)a[6], (float)a[7], q.q_inv, q.q_zoff);
        x[2] = (int)quant4((float)b[0], (float)b[1], (float)b[2], (float)b[3], q.q_inv, q.q_zoff);
        x[3] = (int)quant4((float)b[4], (float)b[5], (float)b[6], (float)b[7], q.q_inv, q.q_zoff);
        o = __builtin_amdgcn_mfma_i32_32x32x32_i8(q.wq[m * 64 + lane], x, o, 0, 0, 0);
    }
    f32x16 r;
#pragma unroll
    for (int g = 0; g < 2; ++g) {                  // rows 0..15 only: registers 0..3 (rows 4lh + k) and 4..7 (rows 8 + 4lh + k)
        const float4 sc = *reinterpret_cast<const float4 *>(q.ss + 8 * g + 4 * lh);
        const float4 sh = *reinterpret_cast<const float4 *>(q.ss + 32 + 8 * g + 4 * lh);
        r[4 * g + 0] = (float)o[4 * g + 0] * sc.x + sh.x; r[4 * g + 1] = (float)o[4 * g + 1] * sc.y + sh.y;
        r[4 * g + 2] = (float)o[4 * g + 2] * sc.z + sh.z; r[4 * g + 3] = (float)o[4 * g + 3] * sc.w + sh.w;
    }
    return r;
}

template <bool Q6>
__global__ __launch_bounds__(TRUNK_NT, SUBS == 1 ? 2 : 1) void le_cond_trunk_kernel(const f16 *__restrict__ img, int H, int W,
                                                               const f16 *__restrict__ wfrag, const float *__restrict__ bias,
                                                               f16 *__restrict__ cond, f16 *__restrict__ cond1, QLast ql)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f16x8 *s_w = reinterpret_cast<f16x8 *>(smem);                               // [NFRAG][64]
    float *s_b = reinterpret_cast<float *>(smem + NFRAG * 64 * 16);              // [NBIAS]
    const int sub = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);      // this thread's 4-wave group: its own tile, patch and staging
    char *s_sub = smem + NFRAG * 64 * 16 + ((NBIAS * 4 + 15) / 16) * 16 + sub * SUB_B;
    f16 *s_in = reinterpret_cast<f16 *>(s_sub);                                  // [3][T_HH][T_HW+2]
    char *s_stg = s_sub + SIN_B;                                                 // [4 waves][32][STG_ROWB]

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tiles_x = (W + T_TW - 1) / T_TW, ntiles = tiles_x * ((H + T_TH - 1) / T_TH);

    // persistent: the 40 KiB of weight fragments are staged once per workgroup, not once per 256 pixels
    for (int e = threadIdx.x; e < NFRAG * 64; e += TRUNK_NT) s_w[e] = reinterpret_cast<const f16x8 *>(wfrag)[e];
    for (int e = threadIdx.x; e < NBIAS; e += TRUNK_NT) s_b[e] = bias[e];

    // this thread's share of a tile's 3 x 10 x 34 input patch, fetched one tile ahead
    constexpr int NE = (3 * T_HH * T_HW + 255) / 256;
    f16 pre[NE];
    auto fetch = [&](int t) {
        const int ox0 = (t % tiles_x) * T_TW, oy0 = (t / tiles_x) * T_TH;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 256 * i;
            const int c = e / (T_HH * T_HW), r = (e / T_HW) % T_HH, q = e % T_HW;
            const int iy = oy0 - 1 + r, ix = ox0 - 1 + q;
            const bool ok = t < ntiles && e < 3 * T_HH * T_HW && iy >= 0 && iy < H && ix >= 0 && ix < W;
            pre[i] = img[ok ? ((size_t)c * H + iy) * W + ix : 0];
            if (!ok) pre[i] = (f16)0.f;
        }
    };
    // the groups of a workgroup walk tiles tb + sub in lockstep (shared barriers); a group past the last tile idles through them
    const int tstep = (int)gridDim.x * SUBS;
    int tb = blockIdx.x * SUBS;
    if (tb < ntiles) fetch(tb + sub);
    for (; tb < ntiles; tb += tstep) {
    const int t = tb + sub;
    const bool active = t < ntiles;
    const int ox0 = (t % tiles_x) * T_TW, oy0 = (t / tiles_x) * T_TH;
    __syncthreads();                                   // layer 1 of the previous tile is done with s_in
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + 256 * i;
        const int c = e / (T_HH * T_HW), r = (e / T_HW) % T_HH, q = e % T_HW;
        if (e < 3 * T_HH * T_HW) s_in[(c * T_HH + r) * (T_HW + 2) + q] = pre[i];
    }
    __syncthreads();
    if (tb + tstep < ntiles) fetch(tb + tstep + sub);
    if (!active) continue;

    // ---- layer 1: 3x3 conv as a K = 27 (padded 32) GEMM, im2col fragments gathered from LDS
    f16x8 bf[2][4];   // activations of the two 32-pixel groups (rows 2*wave, 2*wave+1) as B fragments
    {
        f16x8 xf[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int base = (2 * wave + j) * (T_HW + 2) + l31;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 16 * ks + 8 * lh + e;
                    const int tap = k / 3, c = k % 3;
                    const int off = (c * T_HH + tap / 3) * (T_HW + 2) + tap % 3;
                    xf[j][ks][e] = k < 27 ? s_in[base + off] : (f16)0.f;
                }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f16x8 w0 = s_w[(mt * 2 + 0) * 64 + lane], w1 = s_w[(mt * 2 + 1) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 h = bias_tile_l(s_b + 32 * mt, lh);
                h = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, xf[j][0], h, 0, 0, 0);
                h = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, xf[j][1], h, 0, 0, 0);
                bf[j][2 * mt] = lrelu_pack(h, 0);
                bf[j][2 * mt + 1] = lrelu_pack(h, 1);
            }
        }
    }

    char *stg = s_stg + wave * 32 * STG_ROWB;
    // ---- layers 2..5: 64 -> 64, LeakyReLU(0.1); the output of layer 3 is `cond`
#pragma unroll
    for (int layer = 2; layer <= 5; ++layer) {
        f16x8 nf[2][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x16 g[2];
            g[0] = bias_tile_l(s_b + 64 * (layer - 1) + 32 * mt, lh);
            g[1] = g[0];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f16x8 w = s_w[(4 + (layer - 2) * 8 + mt * 4 + s) * 64 + lane];
                g[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, bf[0][s], g[0], 0, 0, 0);
                g[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, bf[1][s], g[1], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                nf[j][2 * mt] = lrelu_pack(g[j], 0);
                nf[j][2 * mt + 1] = lrelu_pack(g[j], 1);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s) bf[j][s] = nf[j][s];
        if (layer == 3) {
            // store cond (NHWC 64): fragment s holds channels 16s+4lh+{0..3} and 16s+8+4lh+{0..3}
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = oy0 + 2 * wave + j;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f16x4 lo, hi;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { lo[k] = bf[j][s][k]; hi[k] = bf[j][s][4 + k]; }
                    *reinterpret_cast<f16x4 *>(stg + l31 * STG_ROWB + (16 * s + 4 * lh) * 2) = lo;
                    *reinterpret_cast<f16x4 *>(stg + l31 * STG_ROWB + (16 * s + 8 + 4 * lh) * 2) = hi;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (oy < H) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int e = lane + 64 * it, px = e >> 3, c8 = e & 7;
                        if (ox0 + px < W)
                            *reinterpret_cast<f16x8 *>(cond + ((size_t)oy * W + ox0 + px) * 64 + c8 * 8) =
                                *reinterpret_cast<const f16x8 *>(stg + px * STG_ROWB + c8 * 16);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    // ---- layer 6: 64 -> 16 (rows 0..15 of one tile), no activation -> cond1 (NHWC 16)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        f32x16 o;
        if constexpr (Q6) {
            o = qlast_apply(ql, bf[j], lane, lh);
        } else {
            o = bias_tile_l(s_b + 320, lh);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w[(36 + s) * 64 + lane], bf[j][s], o, 0, 0, 0);
        }
        // rows 0..15 live in registers 0..3 (rows 0-3 / 4-7 by lane half) and 4..7 (rows 8-11 / 12-15)
        f16x4 lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo[k] = (f16)o[k]; hi[k] = (f16)o[4 + k]; }
        *reinterpret_cast<f16x4 *>(stg + l31 * STG_ROWB + (4 * lh) * 2) = lo;
        *reinterpret_cast<f16x4 *>(stg + l31 * STG_ROWB + (8 + 4 * lh) * 2) = hi;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const int oy = oy0 + 2 * wave + j;
        const int px = lane >> 1, c8 = lane & 1;
        if (oy < H && ox0 + px < W)
            *reinterpret_cast<f16x8 *>(cond1 + ((size_t)oy * W + ox0 + px) * 16 + c8 * 8) =
                *reinterpret_cast<const f16x8 *>(stg + px * STG_ROWB + c8 * 16);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    }   // tile loop
}


// ------------------------------------------------------------------------------------------------------------------
// CondNet2's tail (HDRUNet3T1_arch.py:47-52: 1x1 64->64, LeakyReLU(0.1), 1x1 64->16) as one pass: the 64-channel
// intermediate never exists in HBM (two launches streamed 384 B per pixel; this one 128 B in, 32 B out).  A wave owns
// 32-pixel groups; both weight sets live in its registers; layer 1 reads its B fragments straight from the NHWC input
// (16 bytes per lane and k-step), layer 2 takes layer 1's accumulator tiles as operands (weights K-permuted at pack
// time, common.h acc_kperm16), exactly as the trunk above chains its layers.
template <bool Q2>
__global__ __launch_bounds__(256) void cond_tail_kernel(const f16 *__restrict__ x, int x_stride, size_t npx,
                                                        const f16 *__restrict__ wfrag, const float *__restrict__ bias,
                                                        f16 *__restrict__ out, QLast ql)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    const f16x8 *fr = reinterpret_cast<const f16x8 *>(wfrag);
    f16x8 w1[2][4], w2[4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int s = 0; s < 4; ++s) w1[mt][s] = fr[(mt * 4 + s) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 4; ++s) w2[s] = fr[(8 + s) * 64 + lane];
    const f32x16 b1a = bias_tile_l(bias, lh), b1b = bias_tile_l(bias + 32, lh), b2 = bias_tile_l(bias + 64, lh);

    const size_t ngroups = (npx + 31) / 32;
    const size_t gstep = (size_t)gridDim.x * 4;
    size_t g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    auto load = [&](size_t grp, f16x8 *xf) {
        size_t px = grp * 32 + l31;
        if (px >= npx) px = npx - 1;                       // tail group: duplicate the last pixel, stores are masked
        const f16 *p = x + px * x_stride + 8 * lh;
#pragma unroll
        for (int s = 0; s < 4; ++s) xf[s] = *reinterpret_cast<const f16x8 *>(p + 16 * s);
    };
    f16x8 cur[4], nxt[4];
    if (g < ngroups) load(g, cur);
    for (; g < ngroups; g += gstep) {
        if (g + gstep < ngroups) load(g + gstep, nxt);
        f32x16 h0 = b1a, h1 = b1b;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[0][s], cur[s], h0, 0, 0, 0);
            h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[1][s], cur[s], h1, 0, 0, 0);
        }
        const f16x8 bf[4] = {lrelu_pack(h0, 0), lrelu_pack(h0, 1), lrelu_pack(h1, 0), lrelu_pack(h1, 1)};
        f32x16 o;
        if constexpr (Q2) {
            o = qlast_apply(ql, bf, lane, lh);
        } else {
            o = b2;
#pragma unroll
            for (int s = 0; s < 4; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[s], bf[s], o, 0, 0, 0);
        }
        // rows 0..15 live in registers 0..3 (rows 4*lh + k) and 4..7 (rows 8 + 4*lh + k)
        const size_t px = g * 32 + l31;
        if (px < npx) {
            f16x4 lo, hi;
#pragma unroll
            for (int k = 0; k < 4; ++k) { lo[k] = (f16)o[k]; hi[k] = (f16)o[4 + k]; }
            *reinterpret_cast<f16x4 *>(out + px * 16 + 4 * lh) = lo;
            *reinterpret_cast<f16x4 *>(out + px * 16 + 8 + 4 * lh) = hi;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) cur[s] = nxt[s];
    }
}

constexpr int TRUNK_SMEM = NFRAG * 64 * 16 + ((NBIAS * 4 + 15) / 16) * 16 + SUBS * SUB_B;

}  // namespace

hipError_t le_cond_trunk_launch(const f16 *img, int H, int W, const f16 *wfrag, const float *bias, f16 *cond, f16 *cond1,
                                int n_cu, hipStream_t s, const QLastArgs *q6)
{
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(le_cond_trunk_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, TRUNK_SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(le_cond_trunk_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, TRUNK_SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const int ntiles = ((W + T_TW - 1) / T_TW) * ((H + T_TH - 1) / T_TH);
    const int per_cu = SUBS == 1 ? 2 : 1;                            // SUBS = 1: two workgroups per CU (LDS: ~60 KiB each)
    const int want = (ntiles + SUBS - 1) / SUBS;
    const int grid = want < per_cu * n_cu ? want : per_cu * n_cu;
    QLast ql{nullptr, nullptr, 0.f, 0.f};
    if (q6) {
        ql = QLast{reinterpret_cast<const i32x4 *>(q6->wq), q6->ss, q6->q_inv, q6->q_zoff};
        hipLaunchKernelGGL(le_cond_trunk_kernel<true>, dim3(grid), dim3(TRUNK_NT), TRUNK_SMEM, s, img, H, W, wfrag, bias, cond, cond1, ql);
    } else {
        hipLaunchKernelGGL(le_cond_trunk_kernel<false>, dim3(grid), dim3(TRUNK_NT), TRUNK_SMEM, s, img, H, W, wfrag, bias, cond, cond1, ql);
    }
    return hipGetLastError();
}

// x: NHWC with x_stride elements per pixel (the first 64 channels are read), out: NHWC 16.  wfrag: 12 fragments
// (layer 1: 2 x 4 natural-k, layer 2: 4 K-permuted, rows 16..31 zero), bias: [64] + [32, upper half zero].
hipError_t cond_tail_launch(const f16 *x, int x_stride, size_t npx, const f16 *wfrag, const float *bias, f16 *out, int n_cu,
                            hipStream_t s, const QLastArgs *q2)
{
    if (!npx || x_stride < 64 || (x_stride % 8)) return hipErrorInvalidValue;
    const size_t ngroups = (npx + 31) / 32;
    size_t grid = (ngroups + 3) / 4;
    if (grid > (size_t)8 * n_cu) grid = (size_t)8 * n_cu;
    QLast ql{nullptr, nullptr, 0.f, 0.f};
    if (q2) {
        ql = QLast{reinterpret_cast<const i32x4 *>(q2->wq), q2->ss, q2->q_inv, q2->q_zoff};
        hipLaunchKernelGGL(cond_tail_kernel<true>, dim3((unsigned)grid), dim3(256), 0, s, x, x_stride, npx, wfrag, bias, out, ql);
    } else {
        hipLaunchKernelGGL(cond_tail_kernel<false>, dim3((unsigned)grid), dim3(256), 0, s, x, x_stride, npx, wfrag, bias, out, ql);
    }
    return hipGetLastError();
}
