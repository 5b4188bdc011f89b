// le_hg_misc.hip -- the non-GEMM-shaped pieces of LE and HG on gfx950.
//
//  conv_c3     3x3 conv from a planar 3-channel image to NHWC f16: LE.conv_first (HDRUNet3T1_arch.py:14), HG.conv1
//              (Hallucination_arch.py:59).  The halo tile sits in LDS as 4-channel pixels (r, g, b, 0: 8 bytes) and the
//              GEMM's K axis is (ky | kx4, c4): one k-step of 16 per kernel row = 4 pixels x 4 channels, of which the
//              4th pixel and the 4th channel meet zero weights.  A lane's B fragment of a k-step is then 16 CONTIGUOUS
//              bytes (pixels kx = 2*lh, 2*lh+1 of row ky) -- one ds_read2_b64 instead of eight 2-byte gathers; 3 MFMAs per
//              32 output channels instead of 2 (K = 48 against 32: the matrix pipe is idle here anyway).  The 4th channel of
//              every staged pixel is 1: a layer without BatchNorm (conv_first) has its bias, rounded to f16 as the reference's
//              fp16 model holds it, in the centre tap's 4th-channel weight and a zero shift -- which is what lets conv32s.hip
//              fuse the layer in front of HR_conv1 without bias registers.  The gather was
//              half of these kernels' time (4K: conv_first 0.171 -> 0.087 ms, conv1 0.222 -> 0.146 ms with it stubbed out).
//  hg_prep     HG_Composite._make_mask + reflect pad to a multiple of 32 (HG_Composite_arch.py:78-101)
//  hg_final_fused   the HG tail: conv1 recomputed, second half of conv10 (1x1 over cat(Up_conv5, conv1)),
//              conv_last (1x1 over cat(conv10, img)) and out = mask*out + img, cropped
//              (Hallucination_arch.py:130-137, HG_Composite_arch.py:103)
// (the SFT layers run inside conv32p.hip, the 2x2 max-pools inside the producing convolutions' epilogues)
#include "launchers.h"

namespace {

// ================================================================================== conv_c3
constexpr int C3_TH = 8, C3_TW = 32;
constexpr int C3_HH = C3_TH + 2, C3_HW = C3_TW + 2;
constexpr int C3_PW = C3_HW + 2;          // pixel pitch of the LDS patch: columns 34, 35 are the kx = 3 dummy reads (zero weights)
constexpr int C3_NP = (C3_HH * C3_HW + 255) / 256;       // halo pixels per thread

// B fragment of kernel row ky for output pixel (row, l31): pixels l31 + 2*lh and l31 + 2*lh + 1 of patch row row + ky
__device__ __forceinline__ f16x8 c3_frag(const f16x4 *s_px, int row, int ky, int l31, int lh)
{
    const f16x4 *q = s_px + (row + ky) * C3_PW + l31 + 2 * lh;
    const f16x4 a = q[0], b = q[1];
    return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// one tile's halo pixels, fetched a tile ahead into registers: thread e = tid + 256 i owns patch pixel (e / 34, e % 34)
struct C3Pre { f16 v[C3_NP][3]; bool ok[C3_NP]; };
__device__ __forceinline__ void c3_fetch(C3Pre &pre, const f16 *__restrict__ in, int H, int W, int oy0, int ox0, int tid)
{
#pragma unroll
    for (int i = 0; i < C3_NP; ++i) {
        const int e = tid + 256 * i;
        const int r = e / C3_HW, q = e % C3_HW;
        const int iy = oy0 - 1 + r, ix = ox0 - 1 + q;
        const bool ok = e < C3_HH * C3_HW && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const size_t o = ok ? (size_t)iy * W + ix : 0;
        pre.ok[i] = ok;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pre.v[i][c] = in[(size_t)c * H * W + o];
            if (!ok) pre.v[i][c] = (f16)0.f;
        }
    }
}
__device__ __forceinline__ void c3_stage(const C3Pre &pre, f16x4 *s_px, int tid)
{
#pragma unroll
    for (int i = 0; i < C3_NP; ++i) {
        const int e = tid + 256 * i;
        // 4th channel = 1: a layer without BatchNorm carries its bias in the centre tap's (otherwise zero) 4th-channel weight
        if (e < C3_HH * C3_HW) s_px[(e / C3_HW) * C3_PW + e % C3_HW] = f16x4{pre.v[i][0], pre.v[i][1], pre.v[i][2], (f16)1.f};
    }
}

// DOT (HG.conv1 only): besides the pooled map the kernel leaves, per pixel, the three partial sums of conv10's second half
// (1x1 over conv1's 64 channels, Hallucination_arch.py:130-133) in part2 [H][W][4] f32 -- the f16 activations, as they lie in the
// accumulator registers, are the B operand of four MFMAs against the k-permuted weight fragments w2frag (the chain
// hg_final_fused recomputes conv1 for); with them the HG tail (hg_final_light) is a per-pixel kernel.
template <int COUT, bool DOT = false>
__global__ __launch_bounds__(256) void conv_c3_kernel(const f16 *__restrict__ in, int H, int W, const f16 *__restrict__ wfrag,
                                                      const float *__restrict__ scale, const float *__restrict__ shift,
                                                      int act, f16 *__restrict__ out, f16 *__restrict__ out_pool, float pool_q_inv,
                                                      float pool_q_zero, const f16 *__restrict__ w2frag = nullptr,
                                                      float *__restrict__ part2 = nullptr)
{
    static_assert(!DOT || COUT == 64, "the fused conv10 half belongs to HG.conv1");
    constexpr int MT = COUT / 32;
    constexpr int ROWB = COUT * 2 + 16;
    __shared__ __attribute__((aligned(16))) f16x4 s_px_all[(DOT ? 2 : 1) * C3_HH * C3_PW];      // DOT: two patches, one barrier per tile
    f16x4 *s_px = s_px_all;
    __shared__ __attribute__((aligned(16))) char s_out[C3_TH * C3_TW * ROWB];
    __shared__ __attribute__((aligned(16))) float s_ss[2 * COUT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tiles_x = (W + C3_TW - 1) / C3_TW, ntiles = tiles_x * ((H + C3_TH - 1) / C3_TH);
    if (tid < COUT) { s_ss[tid] = scale[tid]; s_ss[COUT + tid] = shift[tid]; }
    __shared__ __attribute__((aligned(16))) f16x8 s_w2[DOT ? 4 * 64 : 1];      // read at use: the kernel lives on 3 workgroups per CU
    if constexpr (DOT) s_w2[tid] = reinterpret_cast<const f16x8 *>(w2frag)[tid];
    f16x8 wf[MT][3];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) wf[i][ky] = reinterpret_cast<const f16x8 *>(wfrag)[(i * 3 + ky) * 64 + lane];
    for (int e = tid; e < C3_HH * 2 * (DOT ? 2 : 1); e += 256)
        s_px_all[(e / (C3_HH * 2)) * (C3_HH * C3_PW) + ((e % (C3_HH * 2)) >> 1) * C3_PW + C3_HW + (e & 1)] = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    // persistent over 8x32-pixel tiles, the next tile's image patch fetched while this one computes
    C3Pre pre;
    auto fetch = [&](int t) { c3_fetch(pre, in, H, W, (t / tiles_x) * C3_TH, (t % tiles_x) * C3_TW, tid); };
    const float aslope = act_slope(act);
    int t = blockIdx.x;
    if (t < ntiles) fetch(t);
    // DOT (HG.conv1: pooled output only): ONE barrier per tile.  The patch of tile t+1 is staged into the other buffer while tile
    // t computes (its registers were fetched a tile earlier, the fetch of t+2 follows at once), and everything behind the MFMAs
    // is wave-private: a wave's two pixel rows hold whole 2x2 pooling windows, so it reads back only what it staged itself
    // (LDS operations of one wave complete in order).
    int pbuf = 0;
    if constexpr (DOT) {
        if (t < ntiles) c3_stage(pre, s_px_all, tid);
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
    }
    for (; t < ntiles; t += gridDim.x) {
    const int ox0 = (t % tiles_x) * C3_TW, oy0 = (t / tiles_x) * C3_TH;
    if constexpr (DOT) {
        __syncthreads();                               // this tile's patch is staged; the other buffer's readers (tile t-1) are done
        s_px = s_px_all + pbuf * (C3_HH * C3_PW);
        if (t + (int)gridDim.x < ntiles) c3_stage(pre, s_px_all + (pbuf ^ 1) * (C3_HH * C3_PW), tid);
        if (t + 2 * (int)gridDim.x < ntiles) fetch(t + 2 * gridDim.x);
        pbuf ^= 1;
    } else {
        __syncthreads();                               // the previous tile is done with s_px and s_out
        c3_stage(pre, s_px, tid);
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
    }
    f32x16 acc[MT][2];
    // one row's epilogue (DOT): its 64 f16 activations go to the staging tile and, as they lie in the registers, into the four B
    // fragments of conv10's second half.  Run right behind the row's conv MFMAs, so that only one row of accumulators is live:
    // the kernel must stay under 168 VGPRs for its three workgroups per CU.
    auto dot_row = [&](int j) __attribute__((always_inline)) {
        const int q = (2 * wave + j) * C3_TW + l31;
        f16x8 bf[4];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int cl = i * 32 + 8 * qd + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(s_ss + cl);
                const float4 sh = *reinterpret_cast<const float4 *>(s_ss + COUT + cl);
                f16x4 o;
                o[0] = (f16)act_fast(acc[i][j][4 * qd + 0] * sc.x + sh.x, aslope);
                o[1] = (f16)act_fast(acc[i][j][4 * qd + 1] * sc.y + sh.y, aslope);
                o[2] = (f16)act_fast(acc[i][j][4 * qd + 2] * sc.z + sh.z, aslope);
                o[3] = (f16)act_fast(acc[i][j][4 * qd + 3] * sc.w + sh.w, aslope);
                *reinterpret_cast<f16x4 *>(s_out + q * ROWB + cl * 2) = o;
#pragma unroll
                for (int k = 0; k < 4; ++k) bf[2 * i + (qd >> 1)][4 * (qd & 1) + k] = o[k];
            }
        f32x16 o3;
#pragma unroll
        for (int k = 0; k < 16; ++k) o3[k] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) o3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(s_w2[s4 * 64 + lane], bf[s4], o3, 0, 0, 0);
        const int y = oy0 + 2 * wave + j, x = ox0 + l31;
        if (lh == 0 && y < H && x < W) *reinterpret_cast<float4 *>(part2 + ((size_t)y * W + x) * 4) = make_float4(o3[0], o3[1], o3[2], 0.f);
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 2 * wave + j;
        f16x8 xf[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) xf[ky] = c3_frag(s_px, row, ky, l31, lh);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i][ky], xf[ky], acc[i][j], 0, 0, 0);
        }
        if constexpr (DOT) {
            dot_row(j);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (!DOT) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int cl = i * 32 + 8 * qd + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(s_ss + cl);
                const float4 sh = *reinterpret_cast<const float4 *>(s_ss + COUT + cl);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int q = (2 * wave + j) * C3_TW + l31;
                    f16x4 o;
                    o[0] = (f16)act_fast(acc[i][j][4 * qd + 0] * sc.x + sh.x, aslope);
                    o[1] = (f16)act_fast(acc[i][j][4 * qd + 1] * sc.y + sh.y, aslope);
                    o[2] = (f16)act_fast(acc[i][j][4 * qd + 2] * sc.z + sh.z, aslope);
                    o[3] = (f16)act_fast(acc[i][j][4 * qd + 3] * sc.w + sh.w, aslope);
                    *reinterpret_cast<f16x4 *>(s_out + q * ROWB + cl * 2) = o;
                }
            }
    }
    constexpr int CPP = COUT / 8;
    if constexpr (DOT) {
        // this wave's pooled row: 16 pooled pixels x 8 chunks of 8 channels = two items per lane, from its own two staged rows
        const int Hq = H / 2, Wq = W / 2;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int e = lane + 64 * it;
            const int px = e / CPP, c8 = e % CPP;
            const int oy = oy0 / 2 + wave, ox = ox0 / 2 + px;
            const int q00 = 2 * wave * C3_TW + 2 * px;
            f16x8 v = *reinterpret_cast<const f16x8 *>(s_out + q00 * ROWB + c8 * 16);
            const f16x8 v1 = *reinterpret_cast<const f16x8 *>(s_out + (q00 + 1) * ROWB + c8 * 16);
            const f16x8 v2 = *reinterpret_cast<const f16x8 *>(s_out + (q00 + C3_TW) * ROWB + c8 * 16);
            const f16x8 v3 = *reinterpret_cast<const f16x8 *>(s_out + (q00 + C3_TW + 1) * ROWB + c8 * 16);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f16 m = v[k] > v1[k] ? v[k] : v1[k];
                const f16 m2 = v2[k] > v3[k] ? v2[k] : v3[k];
                v[k] = m > m2 ? m : m2;
            }
            if (oy < Hq && ox < Wq) {
                if (pool_q_inv > 0.f) {        // int8 codes q - 128 of a W8A8 reader, as below
                    unsigned lo = 0, hi = 0;
                    const float z128 = pool_q_zero + 128.f;
                    lo = __builtin_amdgcn_cvt_pk_u8_f32((float)v[0] * pool_q_inv + z128, 0, lo);
                    lo = __builtin_amdgcn_cvt_pk_u8_f32((float)v[1] * pool_q_inv + z128, 1, lo);
                    lo = __builtin_amdgcn_cvt_pk_u8_f32((float)v[2] * pool_q_inv + z128, 2, lo);
                    lo = __builtin_amdgcn_cvt_pk_u8_f32((float)v[3] * pool_q_inv + z128, 3, lo);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32((float)v[4] * pool_q_inv + z128, 0, hi);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32((float)v[5] * pool_q_inv + z128, 1, hi);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32((float)v[6] * pool_q_inv + z128, 2, hi);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32((float)v[7] * pool_q_inv + z128, 3, hi);
                    lo ^= 0x80808080u; hi ^= 0x80808080u;
                    *reinterpret_cast<uint2 *>(reinterpret_cast<int8_t *>(out_pool) + ((size_t)oy * Wq + ox) * COUT + c8 * 8) = make_uint2(lo, hi);
                } else {
                    *reinterpret_cast<f16x8 *>(out_pool + ((size_t)oy * Wq + ox) * COUT + c8 * 8) = v;
                }
            }
        }
        continue;
    }
    __syncthreads();
    for (int e = tid; e < C3_TH * C3_TW * CPP; e += 256) {
        const int q = e / CPP, c8 = e % CPP;
        const int oy = oy0 + q / C3_TW, ox = ox0 + q % C3_TW;
        if (out && oy < H && ox < W)
            *reinterpret_cast<f16x8 *>(out + ((size_t)oy * W + ox) * COUT + c8 * 8) =
                *reinterpret_cast<const f16x8 *>(s_out + q * ROWB + c8 * 16);
    }
    if (out_pool) {
        const int Hp = H / 2, Wp = W / 2;
        for (int e = tid; e < (C3_TH / 2) * (C3_TW / 2) * CPP; e += 256) {
            const int pq = e / CPP, c8 = e % CPP;
            const int py = pq / (C3_TW / 2), px = pq % (C3_TW / 2);
            const int oy = oy0 / 2 + py, ox = ox0 / 2 + px;
            if (oy < Hp && ox < Wp) {
                const int q00 = 2 * py * C3_TW + 2 * px;
                f16x8 v = *reinterpret_cast<const f16x8 *>(s_out + q00 * ROWB + c8 * 16);
                const f16x8 v1 = *reinterpret_cast<const f16x8 *>(s_out + (q00 + 1) * ROWB + c8 * 16);
                const f16x8 v2 = *reinterpret_cast<const f16x8 *>(s_out + (q00 + C3_TW) * ROWB + c8 * 16);
                const f16x8 v3 = *reinterpret_cast<const f16x8 *>(s_out + (q00 + C3_TW + 1) * ROWB + c8 * 16);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f16 m = v[k] > v1[k] ? v[k] : v1[k];
                    const f16 m2 = v2[k] > v3[k] ? v2[k] : v3[k];
                    v[k] = m > m2 ? m : m2;
                }
                if (pool_q_inv > 0.f) {
                    // the pooled map as int8 codes q - 128 of a W8A8 reader (W8A8Conv2d.forward, hdrtvnet_torch.py:351-356)
                    // clamp(rint(v * inv + zero), -128, 127) as the u8 code 128 higher (common.h quant4's convert), back by xor 0x80
                    unsigned lo = 0, hi = 0;
                    const float z128 = pool_q_zero + 128.f;
                    lo = __builtin_amdgcn_cvt_pk_u8_f32((float)v[0] * pool_q_inv + z128, 0, lo);
                    lo = __builtin_amdgcn_cvt_pk_u8_f32((float)v[1] * pool_q_inv + z128, 1, lo);
                    lo = __builtin_amdgcn_cvt_pk_u8_f32((float)v[2] * pool_q_inv + z128, 2, lo);
                    lo = __builtin_amdgcn_cvt_pk_u8_f32((float)v[3] * pool_q_inv + z128, 3, lo);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32((float)v[4] * pool_q_inv + z128, 0, hi);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32((float)v[5] * pool_q_inv + z128, 1, hi);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32((float)v[6] * pool_q_inv + z128, 2, hi);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32((float)v[7] * pool_q_inv + z128, 3, hi);
                    lo ^= 0x80808080u; hi ^= 0x80808080u;
                    *reinterpret_cast<uint2 *>(reinterpret_cast<int8_t *>(out_pool) + ((size_t)oy * Wp + ox) * COUT + c8 * 8) = make_uint2(lo, hi);
                } else {
                    *reinterpret_cast<f16x8 *>(out_pool + ((size_t)oy * Wp + ox) * COUT + c8 * 8) = v;
                }
            }
        }
    }
    }   // tile loop
}

// ================================================================================== conv_c3_q8
// LE.conv_first as a W8A8 layer (the full INT8 recipe; W8A8Conv2d.forward, hdrtvnet_torch.py:351-364; scheme in conv32p.hip's
// header): the three planes are quantised while the patch is staged -- 4 int8 codes per pixel (r, g, b, pad), out-of-image
// pixels code 0 -- and the K axis is (ky | kx4, c4) as in conv_c3: kernel rows 0 and 1 are the two 16-byte halves of ONE
// v_mfma_i32_32x32x32_i8 (lane half lh reads pixels l31 .. l31+3 of patch row y + lh), row 2 the first half of a second one
// whose other half meets zero weights.  Two MFMAs per 32 pixels x 32 channels; the generic path (planar3_to_q8 + conv_q8 on
// 32-byte pixels of which 3 bytes are real) needed nine and a 265 MB intermediate.
__global__ __launch_bounds__(256) void conv_c3_q8_kernel(const f16 *__restrict__ in, int H, int W, const i32x4 *__restrict__ wq,
                                                         const float *__restrict__ scale, const float *__restrict__ shift,
                                                         float q_inv, float q_zoff, int act, f16 *__restrict__ out)
{
    constexpr int COUT = 32, ROWB = COUT * 2 + 16;
    __shared__ __attribute__((aligned(16))) int s_px[C3_HH * C3_PW];
    __shared__ __attribute__((aligned(16))) char s_out[C3_TH * C3_TW * ROWB];
    __shared__ __attribute__((aligned(16))) float s_ss[17 * COUT];          // scale[32] | shift[16 border classes][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tiles_x = (W + C3_TW - 1) / C3_TW, ntiles = tiles_x * ((H + C3_TH - 1) / C3_TH);
    for (int e = tid; e < 17 * COUT; e += 256) s_ss[e] = e < COUT ? scale[e] : shift[e - COUT];
    const i32x4 w0 = wq[lane], w1 = wq[64 + lane];
    for (int e = tid; e < C3_HH * 2; e += 256) s_px[(e >> 1) * C3_PW + C3_HW + (e & 1)] = 0;
    C3Pre pre;
    auto fetch = [&](int t) { c3_fetch(pre, in, H, W, (t / tiles_x) * C3_TH, (t % tiles_x) * C3_TW, tid); };
    const float aslope = act_slope(act);
    int t = blockIdx.x;
    if (t < ntiles) fetch(t);
    for (; t < ntiles; t += gridDim.x) {
        const int ox0 = (t % tiles_x) * C3_TW, oy0 = (t / tiles_x) * C3_TH;
        __syncthreads();                                   // the previous tile is done with s_px and s_out
#pragma unroll
        for (int i = 0; i < C3_NP; ++i) {
            const int e = tid + 256 * i;
            if (e < C3_HH * C3_HW)
                s_px[(e / C3_HW) * C3_PW + e % C3_HW] =
                    pre.ok[i] ? (int)quant4((float)pre.v[i][0], (float)pre.v[i][1], (float)pre.v[i][2], 0.f, q_inv, q_zoff) : 0;
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 2 * wave + j;
            const int *q0 = s_px + (row + lh) * C3_PW + l31, *q1 = s_px + (row + 2) * C3_PW + l31;
            const i32x4 b0 = {q0[0], q0[1], q0[2], q0[3]}, b1 = {q1[0], q1[1], q1[2], q1[3]};
            i32x16 acc;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0;
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1, b1, acc, 0, 0, 0);
            // border class of this lane's pixel: which kernel rows / columns fall outside the image there
            const int oy = oy0 + row, ox = ox0 + l31;
            const int bcls = ((((oy == 0) | ((oy == H - 1) << 1)) << 2) | ((ox == 0) | ((ox == W - 1) << 1))) & 15;
            const int q = row * C3_TW + l31;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int cl = 8 * qd + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(s_ss + cl);
                const float4 sh = *reinterpret_cast<const float4 *>(s_ss + COUT + bcls * COUT + cl);
                f16x4 o;
                o[0] = (f16)act_fast((float)acc[4 * qd + 0] * sc.x + sh.x, aslope);
                o[1] = (f16)act_fast((float)acc[4 * qd + 1] * sc.y + sh.y, aslope);
                o[2] = (f16)act_fast((float)acc[4 * qd + 2] * sc.z + sh.z, aslope);
                o[3] = (f16)act_fast((float)acc[4 * qd + 3] * sc.w + sh.w, aslope);
                *reinterpret_cast<f16x4 *>(s_out + q * ROWB + cl * 2) = o;
            }
        }
        __syncthreads();
        constexpr int CPP = COUT / 8;
        for (int e = tid; e < C3_TH * C3_TW * CPP; e += 256) {
            const int q = e / CPP, c8 = e % CPP;
            const int oy = oy0 + q / C3_TW, ox = ox0 + q % C3_TW;
            if (oy < H && ox < W)
                *reinterpret_cast<f16x8 *>(out + ((size_t)oy * W + ox) * COUT + c8 * 8) =
                    *reinterpret_cast<const f16x8 *>(s_out + q * ROWB + c8 * 16);
        }
    }
}

// ================================================================================== hg_prep
__global__ __launch_bounds__(256) void hg_prep_kernel(const f16 *__restrict__ base, int H, int W, int Hp, int Wp,
                                                      f16 *__restrict__ img_pad, uint8_t *__restrict__ mask, float r,
                                                      float thresh)
{
    const size_t n = (size_t)Hp * Wp;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int yp = (int)(i / Wp), xp = (int)(i % Wp);
        const int ys = yp < H ? yp : 2 * (H - 1) - yp;   // F.pad(mode="reflect"), bottom/right only
        const int xs = xp < W ? xp : 2 * (W - 1) - xp;
        const size_t so = (size_t)ys * W + xs;
        const f16 cr = base[so], cg = base[(size_t)H * W + so], cb = base[2 * (size_t)H * W + so];
        img_pad[i] = cr;
        img_pad[n + i] = cg;
        img_pad[2 * n + i] = cb;
        float m = fmaxf((float)cr, fmaxf((float)cg, (float)cb));
        m = (m - r) / (1.f - r);
        m = fminf(fmaxf(m, 0.f), 1.f);
        mask[i] = m > thresh ? 1 : 0;
    }
}

// ============================================================================ hg_final_fused
// Same result as hg_final without ever materialising conv1_out or Up_conv5's output: conv1
// (3x3, 3->64, BN, ReLU) is recomputed from the padded image exactly as conv_c3 computes it,
// chained into the second half of conv10 (64->3) through the accumulator-as-operand trick, and
// added to the first half's partial sums that conv3x3_glds' ST_PS_DOT3 epilogue left per pixel.
struct HgFinalFusedParams {
    const f16 *img;        // planar f16 [3][Hp][Wp]
    const uint8_t *mask;   // [Hp][Wp]
    const float *part;     // f32 [Hp][Wp][4]: conv10 over Up_conv5's 64 channels
    const f16 *wfrag;      // 6 conv1 fragments (2 x 3 kernel rows) + 4 conv10-second-half fragments (k permuted)
    const float *scale, *shift;   // conv1 folded BatchNorm [64]
    const float *b10, *wl, *bl;   // conv10 bias [3], conv_last [3][6] + [3]
    void *out;
    int out_f32, H, W, Hp, Wp;
};

#ifdef HDRTV_AB      // superseded by conv_c3<64,dot3> + hg_final_light; kept in the A/B library as their bit-identity yardstick
__global__ __launch_bounds__(256) void hg_final_fused_kernel(HgFinalFusedParams p)
{
    __shared__ __attribute__((aligned(16))) f16x4 s_px[C3_HH * C3_PW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
    const int tiles_x = (p.W + C3_TW - 1) / C3_TW, ntiles = tiles_x * ((p.H + C3_TH - 1) / C3_TH);
    const f16x8 *fr = reinterpret_cast<const f16x8 *>(p.wfrag);
    f16x8 w1[2][3], w2[4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) w1[i][ky] = fr[(i * 3 + ky) * 64 + lane];
#pragma unroll
    for (int s = 0; s < 4; ++s) w2[s] = fr[(6 + s) * 64 + lane];
    for (int e = tid; e < C3_HH * 2; e += 256) s_px[(e >> 1) * C3_PW + C3_HW + (e & 1)] = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
    // folded-BatchNorm scale/shift through LDS (kept out of registers: this kernel is latency-bound and
    // lives on occupancy), and the per-pixel inputs of the tail prefetched before any arithmetic
    __shared__ __attribute__((aligned(16))) float s_ss[128];
    if (tid < 64) { s_ss[tid] = p.scale[tid]; s_ss[64 + tid] = p.shift[tid]; }
    // persistent over 8x32-pixel tiles; the next tile's image patch, partial sums and mask are fetched while
    // this one computes (a tile is ~10 MFMAs per wave: unpipelined, the kernel was all load latency)
    C3Pre pre;
    float4 pt_pre[2];
    uint8_t m_pre[2];
    auto fetch = [&](int t) {
        const int ox0 = (t % tiles_x) * C3_TW, oy0 = (t / tiles_x) * C3_TH;
        c3_fetch(pre, p.img, p.Hp, p.Wp, oy0, ox0, tid);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int y = oy0 + 2 * wave + j, x = ox0 + l31;
            const bool ok = lh == 0 && y < p.H && x < p.W;
            const size_t pix = ok ? (size_t)y * p.Wp + x : 0;
            pt_pre[j] = *reinterpret_cast<const float4 *>(p.part + pix * 4);
            m_pre[j] = p.mask[pix];
        }
    };
    const size_t plane_o = (size_t)p.H * p.W;
    int t = blockIdx.x;
    if (t < ntiles) fetch(t);
    for (; t < ntiles; t += gridDim.x) {
    const int ox0 = (t % tiles_x) * C3_TW, oy0 = (t / tiles_x) * C3_TH;
    __syncthreads();                                   // the previous tile is done with s_px
    c3_stage(pre, s_px, tid);
    const float4 pt_cur[2] = {pt_pre[0], pt_pre[1]};
    const float m_cur[2] = {(float)m_pre[0], (float)m_pre[1]};
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 2 * wave + j;
        f16x8 xf[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) xf[ky] = c3_frag(s_px, row, ky, l31, lh);
        f16x8 bf[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 h;
#pragma unroll
            for (int k = 0; k < 16; ++k) h[k] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) h = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[i][ky], xf[ky], h, 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 sc = *reinterpret_cast<const float4 *>(s_ss + 32 * i + 8 * (2 * s + g) + 4 * lh);
                    const float4 sh = *reinterpret_cast<const float4 *>(s_ss + 64 + 32 * i + 8 * (2 * s + g) + 4 * lh);
                    bf[2 * i + s][4 * g + 0] = (f16)fmaxf(h[8 * s + 4 * g + 0] * sc.x + sh.x, 0.f);
                    bf[2 * i + s][4 * g + 1] = (f16)fmaxf(h[8 * s + 4 * g + 1] * sc.y + sh.y, 0.f);
                    bf[2 * i + s][4 * g + 2] = (f16)fmaxf(h[8 * s + 4 * g + 2] * sc.z + sh.z, 0.f);
                    bf[2 * i + s][4 * g + 3] = (f16)fmaxf(h[8 * s + 4 * g + 3] * sc.w + sh.w, 0.f);
                }
        }
        f32x16 o;
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[s], bf[s], o, 0, 0, 0);
        const int y = oy0 + row, x = ox0 + l31;
        if (lh == 0 && y < p.H && x < p.W) {
            const float4 pt = pt_cur[j];
            const float c10[3] = {(float)(f16)(o[0] + pt.x + p.b10[0]), (float)(f16)(o[1] + pt.y + p.b10[1]),
                                  (float)(f16)(o[2] + pt.z + p.b10[2])};
            const f16x4 ctr = s_px[(row + 1) * C3_PW + l31 + 1];
            const float im[3] = {(float)ctr[0], (float)ctr[1], (float)ctr[2]};
            const float m = m_cur[j];
            const size_t oo = (size_t)y * p.W + x;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float v = p.bl[ch];
#pragma unroll
                for (int k = 0; k < 3; ++k) v += p.wl[ch * 6 + k] * c10[k] + p.wl[ch * 6 + 3 + k] * im[k];
                v = (float)(f16)v;
                const float res = m * v + im[ch];
                if (p.out_f32) reinterpret_cast<float *>(p.out)[ch * plane_o + oo] = res;
                else reinterpret_cast<f16 *>(p.out)[ch * plane_o + oo] = (f16)res;
            }
        }
    }
    }   // tile loop
}
#endif

// ============================================================================ hg_final_light
// The HG tail when conv1's kernel has already left conv10's second half per pixel (conv_c3<64, DOT>): per pixel
//   conv10 = f16((part2 + part) + b10), conv_last over cat(conv10, img) rounded to f16, out = mask * that + img, cropped
// (Hallucination_arch.py:130-137, HG_Composite_arch.py:103) -- the same expressions, in the same order, as hg_final_fused.
struct HgFinalLightParams {
    const f16 *img;        // planar f16 [3][Hp][Wp]
    const uint8_t *mask;   // [Hp][Wp]
    const float *part, *part2;    // f32 [Hp][Wp][4]: conv10 over Up_conv5's / conv1's 64 channels
    const float *b10, *wl, *bl;
    void *out;
    int out_f32, H, W, Hp, Wp;
};

__global__ __launch_bounds__(256) void hg_final_light_kernel(HgFinalLightParams p)
{
    const int nx = (p.W + 3) >> 2;
    const size_t total = (size_t)p.H * nx, plane_i = (size_t)p.Hp * p.Wp, plane_o = (size_t)p.H * p.W;
    float b10[3], wl[18], bl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { b10[k] = p.b10[k]; bl[k] = p.bl[k]; }
#pragma unroll
    for (int k = 0; k < 18; ++k) wl[k] = p.wl[k];
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int y = (int)(e / nx), x0 = (int)(e - (size_t)y * nx) * 4;
        const size_t pix = (size_t)y * p.Wp + x0;          // Wp is a multiple of 32: the four pixels exist in every padded tensor
        float4 pt[4], p2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { pt[k] = reinterpret_cast<const float4 *>(p.part)[pix + k]; p2[k] = reinterpret_cast<const float4 *>(p.part2)[pix + k]; }
        f16x4 iv[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) iv[ch] = *reinterpret_cast<const f16x4 *>(p.img + ch * plane_i + pix);
        const uint32_t m4 = *reinterpret_cast<const uint32_t *>(p.mask + pix);
        float res[3][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float c10[3] = {(float)(f16)(p2[k].x + pt[k].x + b10[0]), (float)(f16)(p2[k].y + pt[k].y + b10[1]),
                                  (float)(f16)(p2[k].z + pt[k].z + b10[2])};
            const float im[3] = {(float)iv[0][k], (float)iv[1][k], (float)iv[2][k]};
            const float m = (float)((m4 >> (8 * k)) & 0xff);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float v = bl[ch];
#pragma unroll
                for (int t = 0; t < 3; ++t) v += wl[ch * 6 + t] * c10[t] + wl[ch * 6 + 3 + t] * im[t];
                v = (float)(f16)v;
                res[ch][k] = m * v + im[ch];
            }
        }
        const size_t oo = (size_t)y * p.W + x0;
        const bool vec = (p.W & 3) == 0;                    // then x0 + 3 < W and the row starts are 16-byte (f32) / 8-byte (f16) aligned
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            if (p.out_f32) {
                float *d = reinterpret_cast<float *>(p.out) + ch * plane_o + oo;
                if (vec) *reinterpret_cast<float4 *>(d) = make_float4(res[ch][0], res[ch][1], res[ch][2], res[ch][3]);
                else for (int k = 0; k < 4 && x0 + k < p.W; ++k) d[k] = res[ch][k];
            } else {
                f16 *d = reinterpret_cast<f16 *>(p.out) + ch * plane_o + oo;
                if (vec) *reinterpret_cast<f16x4 *>(d) = f16x4{(f16)res[ch][0], (f16)res[ch][1], (f16)res[ch][2], (f16)res[ch][3]};
                else for (int k = 0; k < 4 && x0 + k < p.W; ++k) d[k] = (f16)res[ch][k];
            }
        }
    }
}

inline int grid_for(size_t n, int per_block)
{
    size_t g = (n + per_block - 1) / per_block;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

hipError_t conv_c3_launch(const f16 *in, int H, int W, const f16 *wfrag, const float *scale, const float *shift, int cout,
                          int act, f16 *out, f16 *out_pool, int n_cu, hipStream_t s, float pool_q_inv, float pool_q_zero,
                          const f16 *w2frag, float *part2)
{
    if ((w2frag != nullptr) != (part2 != nullptr) || (part2 && (cout != 64 || out || !out_pool))) return hipErrorInvalidValue;
    const int ntiles = ((W + C3_TW - 1) / C3_TW) * ((H + C3_TH - 1) / C3_TH);
    // persistent: as many workgroups as are resident at once (registers: 3 / 2 per CU today), one round
    static int occ[3] = {0, 0, 0};
    int &per_cu = occ[part2 ? 2 : (cout == 64)];
    if (per_cu == 0) {
        int nb = 0;
        const hipError_t e = cout == 32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_c3_kernel<32>, 256, 0)
                           : part2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_c3_kernel<64, true>, 256, 0)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_c3_kernel<64>, 256, 0);
        per_cu = (e == hipSuccess && nb >= 1) ? nb : 2;
    }
    const dim3 grid(ntiles < per_cu * n_cu ? ntiles : per_cu * n_cu);
    if (cout == 32)
        hipLaunchKernelGGL(conv_c3_kernel<32>, grid, dim3(256), 0, s, in, H, W, wfrag, scale, shift, act, out, out_pool, pool_q_inv, pool_q_zero,
                           (const f16 *)nullptr, (float *)nullptr);
    else if (cout == 64 && part2)
        hipLaunchKernelGGL((conv_c3_kernel<64, true>), grid, dim3(256), 0, s, in, H, W, wfrag, scale, shift, act, out, out_pool, pool_q_inv, pool_q_zero,
                           w2frag, part2);
    else if (cout == 64)
        hipLaunchKernelGGL(conv_c3_kernel<64>, grid, dim3(256), 0, s, in, H, W, wfrag, scale, shift, act, out, out_pool, pool_q_inv, pool_q_zero,
                           (const f16 *)nullptr, (float *)nullptr);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t conv_c3_q8_launch(const f16 *in, int H, int W, const int8_t *wq, const float *scale, const float *shift, float q_inv,
                             float q_zoff, int act, f16 *out, int n_cu, hipStream_t s)
{
    const int ntiles = ((W + C3_TW - 1) / C3_TW) * ((H + C3_TH - 1) / C3_TH);
    static int per_cu = 0;
    if (per_cu == 0) {
        int nb = 0;
        per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_c3_q8_kernel, 256, 0) == hipSuccess && nb >= 1) ? nb : 2;
    }
    const dim3 grid(ntiles < per_cu * n_cu ? ntiles : per_cu * n_cu);
    hipLaunchKernelGGL(conv_c3_q8_kernel, grid, dim3(256), 0, s, in, H, W, reinterpret_cast<const i32x4 *>(wq), scale, shift, q_inv, q_zoff,
                       act, out);
    return hipGetLastError();
}

hipError_t hg_prep_launch(const f16 *base, int H, int W, int Hp, int Wp, f16 *img_pad, uint8_t *mask, float r, float thresh,
                          hipStream_t s)
{
    hipLaunchKernelGGL(hg_prep_kernel, dim3(grid_for((size_t)Hp * Wp, 256)), dim3(256), 0, s, base, H, W, Hp, Wp, img_pad,
                       mask, r, thresh);
    return hipGetLastError();
}

hipError_t hg_final_light_launch(const HgFinalFusedArgs &a, const float *part2, hipStream_t s)
{
    if (!part2 || (a.Wp & 31) || a.W > a.Wp || a.H > a.Hp) return hipErrorInvalidValue;
    HgFinalLightParams p;
    p.img = a.img; p.mask = a.mask; p.part = a.part; p.part2 = part2; p.b10 = a.b10; p.wl = a.wl; p.bl = a.bl;
    p.out = a.out; p.out_f32 = a.out_f32; p.H = a.H; p.W = a.W; p.Hp = a.Hp; p.Wp = a.Wp;
    hipLaunchKernelGGL(hg_final_light_kernel, dim3(grid_for((size_t)a.H * ((a.W + 3) / 4), 256)), dim3(256), 0, s, p);
    return hipGetLastError();
}

#ifdef HDRTV_AB
hipError_t hg_final_fused_launch(const HgFinalFusedArgs &a, int n_cu, hipStream_t s)
{
    HgFinalFusedParams p;
    p.img = a.img; p.mask = a.mask; p.part = a.part; p.wfrag = a.wfrag; p.scale = a.scale; p.shift = a.shift;
    p.b10 = a.b10; p.wl = a.wl; p.bl = a.bl; p.out = a.out; p.out_f32 = a.out_f32; p.H = a.H; p.W = a.W; p.Hp = a.Hp; p.Wp = a.Wp;
    const int ntiles = ((a.W + C3_TW - 1) / C3_TW) * ((a.H + C3_TH - 1) / C3_TH);
    // persistent: exactly as many workgroups as are resident at once (registers allow 3 per CU today; a 4th per CU would
    // start when the first ones finish and run a second, mostly empty round)
    static int per_cu = 0;
    if (per_cu == 0) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, hg_final_fused_kernel, 256, 0) != hipSuccess || nb < 1) nb = 3;
        per_cu = nb;
    }
    const int grid = ntiles < per_cu * n_cu ? ntiles : per_cu * n_cu;
    hipLaunchKernelGGL(hg_final_fused_kernel, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}
#endif
