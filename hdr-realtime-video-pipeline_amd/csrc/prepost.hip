// prepost.hip -- the HBM-bound per-pixel stages either side of the network (gfx950).
//
//  pre_unpack      u8 HWC BGR -> f16 planar RGB * (1/255)     hdrtvnet_torch.py:2256-2261
//  cond_resize     0.25x antialiased bicubic -> f16 planar    hdrtvnet_torch.py:2278-2285
//  post_u8         planar -> u8 HWC BGR                       hdrtvnet_torch.py:2357-2361
//  post_rgb48      planar -> u16 HWC RGB (rgb48le)            gui_pipeline_worker_feeders.py:223-227
//  post_pq_rgb48   BT.709->BT.2020 matrix + ST.2084 PQ + u16  (north-star display variant)
//
// All are one pass, 8 pixels per lane, 16-byte planar accesses and 8/16-byte interleaved
// accesses, so every wave instruction moves whole cache lines.  Quantisers use __fmul_rn /
// __fadd_rn: the reference rounds the multiply and the add separately.
#include "launchers.h"

namespace {

__device__ __forceinline__ float clamp01(float v)
{
    // torch.clamp semantics incl. NaN propagation
    return v != v ? v : fminf(fmaxf(v, 0.f), 1.f);
}

__global__ __launch_bounds__(256) void pre_unpack_kernel(const uint8_t *__restrict__ bgr, f16 *__restrict__ out,
                                                         size_t npix)
{
    const float k = (float)(1.0 / 255.0);
    const size_t ngrp = npix / 8;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngrp; g += (size_t)gridDim.x * blockDim.x) {
        const uint2 *src = reinterpret_cast<const uint2 *>(bgr + g * 24);
        const uint2 a = src[0], b = src[1], c = src[2];
        const uint32_t w[6] = {a.x, a.y, b.x, b.y, c.x, c.y};
        f16x8 r, gg, bb;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = i * 3;
            const uint32_t vb = (w[o >> 2] >> ((o & 3) * 8)) & 0xff;
            const uint32_t vg = (w[(o + 1) >> 2] >> (((o + 1) & 3) * 8)) & 0xff;
            const uint32_t vr = (w[(o + 2) >> 2] >> (((o + 2) & 3) * 8)) & 0xff;
            r[i] = (f16)__fmul_rn((float)vr, k);
            gg[i] = (f16)__fmul_rn((float)vg, k);
            bb[i] = (f16)__fmul_rn((float)vb, k);
        }
        *reinterpret_cast<f16x8 *>(out + g * 8) = r;
        *reinterpret_cast<f16x8 *>(out + npix + g * 8) = gg;
        *reinterpret_cast<f16x8 *>(out + 2 * npix + g * 8) = bb;
    }
    // tail (< 8 pixels)
    if (blockIdx.x == 0 && threadIdx.x < (npix & 7)) {
        const size_t i = ngrp * 8 + threadIdx.x;
        out[i] = (f16)__fmul_rn((float)bgr[i * 3 + 2], k);
        out[npix + i] = (f16)__fmul_rn((float)bgr[i * 3 + 1], k);
        out[2 * npix + i] = (f16)__fmul_rn((float)bgr[i * 3 + 0], k);
    }
}

// ---- 0.25x antialiased bicubic.  Tap tables (ATen _upsample_bicubic2d_aa weights, computed on
// the host in fp32 exactly as ATen does) : wtab[o][0..16], first input index mn[o], count ns[o].
constexpr int RT_W = 32, RT_H = 8;          // output tile
constexpr int RT_IW = RT_W * 4 + 12;        // 140 input columns cover 32 outputs
constexpr int RT_IH = RT_H * 4 + 12;        // 44 input rows cover 8 outputs
constexpr int AA_TAPS = 17;

__global__ __launch_bounds__(256) void cond_resize_kernel(const f16 *__restrict__ in, f16 *__restrict__ out, int H,
                                                          int W, int Ho, int Wo, const float *__restrict__ wx,
                                                          const int *__restrict__ xmn, const int *__restrict__ xns,
                                                          const float *__restrict__ wy, const int *__restrict__ ymn,
                                                          const int *__restrict__ yns)
{
    __shared__ float s_in[RT_IH][RT_IW + 1];
    __shared__ float s_h[RT_IH][RT_W + 1];
    const int c = blockIdx.z;
    const int ox0 = blockIdx.x * RT_W, oy0 = blockIdx.y * RT_H;
    const int ix0 = xmn[ox0], iy0 = ymn[oy0];
    const f16 *src = in + (size_t)c * H * W;
    for (int e = threadIdx.x; e < RT_IH * RT_IW; e += 256) {
        const int r = e / RT_IW, q = e % RT_IW;
        const int iy = iy0 + r, ix = ix0 + q;
        s_in[r][q] = (iy < H && ix < W) ? (float)src[(size_t)iy * W + ix] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < RT_IH * RT_W; e += 256) {
        const int r = e / RT_W, o = e % RT_W;
        const int ox = ox0 + o;
        float s = 0.f;
        if (ox < Wo) {
            const int base = xmn[ox] - ix0, n = xns[ox];
            const float *w = wx + (size_t)ox * AA_TAPS;
            for (int j = 0; j < n; ++j) s = __fadd_rn(s, __fmul_rn(w[j], s_in[r][base + j]));
        }
        s_h[r][o] = s;
    }
    __syncthreads();
    {
        const int o = threadIdx.x % RT_W, r = threadIdx.x / RT_W;
        const int ox = ox0 + o, oy = oy0 + r;
        if (ox < Wo && oy < Ho) {
            const int base = ymn[oy] - iy0, n = yns[oy];
            const float *w = wy + (size_t)oy * AA_TAPS;
            float s = 0.f;
            for (int j = 0; j < n; ++j) s = __fadd_rn(s, __fmul_rn(w[j], s_h[base + j][o]));
            out[((size_t)c * Ho + oy) * Wo + ox] = (f16)s;
        }
    }
}

// ---- pre_fused: both halves of preprocess in one pass over the u8 frame ----------------------
// A workgroup owns a 32 x 8 tile of the 0.25x condition map = 128 x 32 input pixels.  It stages the 140 x 44 u8 BGR
// patch those outputs' 16-tap windows cover (aligned dword loads, each byte converted once to the f16 value
// fp16(float(u8) * fp32(1/255)) the reference's tensor holds), writes the 128 x 32 pixels it owns to the three f16 planes
// with 16-byte stores, and resamples the SAME values into the condition tile: horizontal pass then vertical pass,
// fp32, the tap order and rounding of cond_resize_kernel (= ATen's separable _upsample_bicubic2d_aa).  The frame is read
// from HBM once (3 B per pixel; halo re-reads hit L2) instead of 3 B + 6 B.
// mode 1: HDRTVNetTorch(fast_condition_resize=True) -- F.interpolate(0.25, bilinear, align_corners=False): source
// coordinate 4d + 1.5, i.e. the mean of pixels (4d+1, 4d+2) in each direction (hdrtvnet_torch.py:2269-2276).
// mode 2: HDRTVNET_ZERO_COND -- the condition map is zero (hdrtvnet_torch.py:2265-2267).
constexpr int PF_OW = 32, PF_OH = 8, PF_IW = 140, PF_IH = 44, PF_ROWDW = 107;   // 107 dwords cover 420 B at any alignment
constexpr int PF_PITCH = 144;

__global__ __launch_bounds__(256) void pre_fused_kernel(const uint8_t *__restrict__ bgr, f16 *__restrict__ out, f16 *__restrict__ cond,
                                                        int H, int W, int Ho, int Wo, const float *__restrict__ wx,
                                                        const int *__restrict__ xmn, const int *__restrict__ xns,
                                                        const float *__restrict__ wy, const int *__restrict__ ymn,
                                                        const int *__restrict__ yns, int mode)
{
    __shared__ __attribute__((aligned(16))) f16 s_in[3][PF_IH][PF_PITCH];     // 288-byte rows: 16-byte reads at columns 8k
    __shared__ float s_h[3][PF_IH][PF_OW + 1];
    __shared__ float s_wx[PF_OW][AA_TAPS], s_wy[PF_OH][AA_TAPS];      // this tile's tap tables (pitch 17: conflict-free across outputs)
    __shared__ int s_xb[PF_OW], s_xn[PF_OW], s_yb[PF_OH], s_yn[PF_OH];
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * PF_OW, oy0 = blockIdx.y * PF_OH;
    const int ix0 = xmn[ox0], iy0 = ymn[oy0];
    for (int e = tid; e < PF_OW * AA_TAPS; e += 256) {
        const int o = e / AA_TAPS;
        s_wx[o][e - o * AA_TAPS] = ox0 + o < Wo ? wx[(size_t)(ox0 + o) * AA_TAPS + (e - o * AA_TAPS)] : 0.f;
    }
    if (tid < PF_OH * AA_TAPS) {
        const int o = tid / AA_TAPS;
        s_wy[o][tid - o * AA_TAPS] = oy0 + o < Ho ? wy[(size_t)(oy0 + o) * AA_TAPS + (tid - o * AA_TAPS)] : 0.f;
    }
    if (tid < PF_OW) { const bool in = ox0 + tid < Wo; s_xb[tid] = in ? xmn[ox0 + tid] - ix0 : 0; s_xn[tid] = in ? xns[ox0 + tid] : 0; }
    if (tid >= 64 && tid < 64 + PF_OH) { const int o = tid - 64; const bool in = oy0 + o < Ho; s_yb[o] = in ? ymn[oy0 + o] - iy0 : 0; s_yn[o] = in ? yns[oy0 + o] : 0; }
    const size_t total = (size_t)H * W * 3;
    const float k255 = (float)(1.0 / 255.0);
    // all of this thread's dword loads first (19 independent loads in flight), then the byte unpacking: a loop that loads and
    // unpacks one dword at a time pays one HBM round trip per dword
    constexpr int NLD = (PF_IH * PF_ROWDW + 255) / 256;
    uint32_t word[NLD];
    long rel[NLD];                                     // byte offset of the dword's first byte from its row segment's start; < -3: skip
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int r = e / PF_ROWDW, d = e - r * PF_ROWDW;
        const int iy = iy0 + r;
        const size_t row0 = ((size_t)(iy < H ? iy : 0) * W + ix0) * 3;
        const size_t addr = (row0 & ~(size_t)3) + 4 * (size_t)d;
        const bool ok = e < PF_IH * PF_ROWDW && iy < H && addr < total;
        rel[i] = ok ? (long)addr - (long)row0 : -1000;
        word[i] = 0;
        if (ok) {
            if (addr + 4 <= total) {
                word[i] = *reinterpret_cast<const uint32_t *>(bgr + addr);
            } else {
                for (size_t k = 0; addr + k < total; ++k) word[i] |= (uint32_t)bgr[addr + k] << (8 * k);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int r = e / PF_ROWDW;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int b = (int)rel[i] + k;
            if (b >= 0 && b < PF_IW * 3) {
                const int q = (b * 683) >> 11, ch = b - 3 * q;             // b / 3 for b < 600
                if (ix0 + q < W) s_in[2 - ch][r][q] = (f16)__fmul_rn((float)((word[i] >> (8 * k)) & 0xff), k255);
            }
        }
    }
    __syncthreads();
    // ---- the pixels this tile owns -> f16 planes.  The last tile of a row / column also owns the W % 4 (H % 4) remainder.
    const int wx0 = 4 * ox0, wy0 = 4 * oy0;
    const int wx1 = (ox0 + PF_OW >= Wo) ? W : wx0 + 4 * PF_OW, wy1 = (oy0 + PF_OH >= Ho) ? H : wy0 + 4 * PF_OH;
    constexpr int NCX = 4 * PF_OW / 8 + 1, NRY = 4 * PF_OH + 3;
    const size_t npix = (size_t)H * W;
    for (int e = tid; e < 3 * NRY * NCX; e += 256) {
        const int c = e / (NRY * NCX), rr = (e / NCX) % NRY, j = e % NCX;
        const int y = wy0 + rr, x = wx0 + 8 * j;
        if (y >= wy1 || x >= wx1) continue;
        const f16 *src = &s_in[c][y - iy0][x - ix0];
        f16 *dst = out + c * npix + (size_t)y * W + x;
        if (x + 8 <= wx1) {
            f16x8 v;
            if (((x - ix0) & 1) == 0) {            // (always, for the 0.25x tables: xmn is even) four dword reads
                const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
                i32x4 u;
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = (int)s32[i];
                v = __builtin_bit_cast(f16x8, u);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = src[i];
            }
            *reinterpret_cast<f16x8 *>(dst) = v;
        } else {
            for (int i = 0; x + i < wx1; ++i) dst[i] = src[i];
        }
    }
    if (mode == 2) {
        for (int e = tid; e < 3 * PF_OH * PF_OW; e += 256) {
            const int c = e / (PF_OH * PF_OW), r = (e / PF_OW) % PF_OH, o = e % PF_OW;
            if (ox0 + o < Wo && oy0 + r < Ho) cond[((size_t)c * Ho + oy0 + r) * Wo + ox0 + o] = (f16)0.f;
        }
        return;
    }
    if (mode == 1) {
        for (int e = tid; e < 3 * PF_OH * PF_OW; e += 256) {
            const int c = e / (PF_OH * PF_OW), r = (e / PF_OW) % PF_OH, o = e % PF_OW;
            const int ox = ox0 + o, oy = oy0 + r;
            if (ox < Wo && oy < Ho) {
                const int y1 = 4 * oy + 1, x1 = 4 * ox + 1;
                const int yp = y1 < H - 1 ? 1 : 0, xp = x1 < W - 1 ? 1 : 0;
                const float a = (float)s_in[c][y1 - iy0][x1 - ix0], b = (float)s_in[c][y1 - iy0][x1 + xp - ix0];
                const float cc = (float)s_in[c][y1 + yp - iy0][x1 - ix0], dd = (float)s_in[c][y1 + yp - iy0][x1 + xp - ix0];
                // upsample_bilinear2d: h0lambda * (w0lambda * a + w1lambda * b) + h1lambda * (w0lambda * c + w1lambda * d), lambdas 0.5
                const float top = __fadd_rn(__fmul_rn(0.5f, a), __fmul_rn(0.5f, b)), bot = __fadd_rn(__fmul_rn(0.5f, cc), __fmul_rn(0.5f, dd));
                cond[((size_t)c * Ho + oy) * Wo + ox] = (f16)__fadd_rn(__fmul_rn(0.5f, top), __fmul_rn(0.5f, bot));
            }
        }
        return;
    }
    // horizontal pass: a thread owns the four adjacent outputs o0 .. o0 + 3 (o0 = 4 * (tid % 8), the same in every trip: their
    // 4 x 16 weights stay in registers) of one row per trip.  Interior outputs -- 16 taps each, windows 4 samples apart -- read
    // their 28 samples as four 16-byte LDS reads instead of 64 two-byte gathers; anything else (image borders) goes tap by tap.
    // Either way: fp32, taps in order, separate multiply and add (ATen's _upsample_bicubic2d_aa, as cond_resize_kernel).
    {
        const int o0 = 4 * (tid & 7);
        float wv[4][16];
        int xb[4], xn1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            xb[i] = s_xb[o0 + i]; xn1[i] = s_xn[o0 + i] - 1;     // n1 < 0 for outputs beyond the image
#pragma unroll
            for (int j = 0; j < 16; ++j) wv[i][j] = s_wx[o0 + i][j];
        }
        const int wstart = xb[0] & ~7, d = xb[0] - wstart;
        const bool fast = (d == 0 || d == 2) && xn1[0] == 15 && xn1[1] == 15 && xn1[2] == 15 && xn1[3] == 15 &&
                          xb[1] == xb[0] + 4 && xb[2] == xb[0] + 8 && xb[3] == xb[0] + 12;
        for (int e = tid >> 3; e < 3 * PF_IH; e += 32) {
            const int c = e / PF_IH, r = e - c * PF_IH;
            float sacc[4] = {0.f, 0.f, 0.f, 0.f};
            if (iy0 + r < H) {
                if (fast) {
                    const f16x8 *src = reinterpret_cast<const f16x8 *>(&s_in[c][r][wstart]);
                    const f16x8 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
                    float xv[32];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { xv[k] = (float)q0[k]; xv[8 + k] = (float)q1[k]; xv[16 + k] = (float)q2[k]; xv[24 + k] = (float)q3[k]; }
                    if (d == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 16; ++j) sacc[i] = __fadd_rn(sacc[i], __fmul_rn(wv[i][j], xv[4 * i + j]));
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 16; ++j) sacc[i] = __fadd_rn(sacc[i], __fmul_rn(wv[i][j], xv[2 + 4 * i + j]));
                    }
                } else {
                    // taps beyond an output's count have weight 0 in the table and re-read the last valid sample: they add exactly +0
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (xn1[i] >= 0) {
                            float xv[16];
#pragma unroll
                            for (int j = 0; j < 16; ++j) xv[j] = (float)s_in[c][r][xb[i] + (j < xn1[i] ? j : xn1[i])];
#pragma unroll
                            for (int j = 0; j < 16; ++j) sacc[i] = __fadd_rn(sacc[i], __fmul_rn(wv[i][j], xv[j]));
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s_h[c][r][o0 + i] = sacc[i];
        }
    }
    __syncthreads();
    for (int e = tid; e < 3 * PF_OH * PF_OW; e += 256) {
        const int c = e / (PF_OH * PF_OW), r = (e / PF_OW) % PF_OH, o = e % PF_OW;
        const int ox = ox0 + o, oy = oy0 + r;
        if (ox < Wo && oy < Ho) {
            const int base = s_yb[r], n1 = s_yn[r] - 1;
            float sacc = 0.f, xv[16], wv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { xv[j] = s_h[c][base + (j < n1 ? j : n1)][o]; wv[j] = s_wy[r][j]; }
#pragma unroll
            for (int j = 0; j < 16; ++j) sacc = __fadd_rn(sacc, __fmul_rn(wv[j], xv[j]));
            cond[((size_t)c * Ho + oy) * Wo + ox] = (f16)sacc;
        }
    }
}

// ---- post-process quantisers ----------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void load8(const T *p, float (&v)[8]);
template <>
__device__ __forceinline__ void load8<f16>(const f16 *p, float (&v)[8])
{
    const f16x8 x = *reinterpret_cast<const f16x8 *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
}
template <>
__device__ __forceinline__ void load8<float>(const float *p, float (&v)[8])
{
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// u8: the reference quantises in the tensor's own dtype (fp16 model: every op rounds to fp16).
template <typename T>
__device__ __forceinline__ uint32_t quant_u8(float x)
{
    if (sizeof(T) == 2) {
        const f16 c = (f16)clamp01(x);
        const f16 m = (f16)__fmul_rn((float)c, 255.f);
        const f16 a = (f16)__fadd_rn((float)m, 0.5f);
        return (uint32_t)(int)(float)a & 0xff;
    }
    return (uint32_t)(int)__fadd_rn(__fmul_rn(clamp01(x), 255.f), 0.5f) & 0xff;
}

__device__ __forceinline__ uint32_t quant_u16(float x)
{
    return (uint32_t)(int)__fadd_rn(__fmul_rn(clamp01(x), 65535.f), 0.5f) & 0xffff;
}

template <typename T>
__global__ __launch_bounds__(256) void post_u8_kernel(const T *__restrict__ in, uint8_t *__restrict__ bgr, size_t npix)
{
    const size_t ngrp = npix / 8;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngrp; g += (size_t)gridDim.x * blockDim.x) {
        float r[8], gg[8], b[8];
        load8<T>(in + g * 8, r);
        load8<T>(in + npix + g * 8, gg);
        load8<T>(in + 2 * npix + g * 8, b);
        uint32_t w[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = i * 3;
            w[o >> 2] |= quant_u8<T>(b[i]) << ((o & 3) * 8);
            w[(o + 1) >> 2] |= quant_u8<T>(gg[i]) << (((o + 1) & 3) * 8);
            w[(o + 2) >> 2] |= quant_u8<T>(r[i]) << (((o + 2) & 3) * 8);
        }
        uint2 *dst = reinterpret_cast<uint2 *>(bgr + g * 24);
        dst[0] = make_uint2(w[0], w[1]);
        dst[1] = make_uint2(w[2], w[3]);
        dst[2] = make_uint2(w[4], w[5]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (npix & 7)) {
        const size_t i = ngrp * 8 + threadIdx.x;
        bgr[i * 3 + 0] = (uint8_t)quant_u8<T>((float)in[2 * npix + i]);
        bgr[i * 3 + 1] = (uint8_t)quant_u8<T>((float)in[npix + i]);
        bgr[i * 3 + 2] = (uint8_t)quant_u8<T>((float)in[i]);
    }
}

// PQ constants: gui_objective_metrics.py:63-67
#define PQ_M1 0.1593017578125f
#define PQ_M2 78.84375f
#define PQ_C1 0.8359375f
#define PQ_C2 18.8515625f
#define PQ_C3 18.6875f

__device__ __forceinline__ float pq_oetf(float nits)
{
    float y = nits / 10000.f;
    y = fminf(fmaxf(y, 0.f), 1.f);
    const float yp = powf(y, PQ_M1);
    return powf((PQ_C1 + PQ_C2 * yp) / (1.f + PQ_C3 * yp), PQ_M2);
}

// Exact u16 code of a PQ level: code(y) = floor(pq(y) * 65535 + 0.5) with the OETF in double precision, as an integer
// function of the fp32 argument y.  bnd[v] (v = 1..65535) is the smallest fp32 y whose code is >= v (built on the host in
// double, hdrtv_api.hip pq_boundaries; 256 KiB, L2-resident); the fp32 evaluation lands within a few codes of the answer
// and two compares against the table settle it.  The result does not depend on any device math-library rounding.
// First guess without transcendental functions: lut[i] (appended to bnd at PQ_LUT_OFF) is the exact code at the fp32 value whose
// bit pattern is (PQ_LUT_BASE + i) << 17 -- 64 steps per binary octave from 2^-27 to 1, the curve's own near-logarithmic
// spacing -- and the code in between is interpolated on the low 17 mantissa bits (within 2 codes of the truth everywhere).
constexpr int PQ_LUT_BASE = (127 - 27) << 6, PQ_LUT_N = 27 * 64 + 2, PQ_LUT_OFF = 65536;
__device__ __forceinline__ uint32_t pq_code(float lin, float peak, const float *__restrict__ bnd)
{
    float y = __fdiv_rn(__fmul_rn(lin, peak), 10000.f);
    y = fminf(fmaxf(y, 0.f), 1.f);
    const uint32_t bits = __float_as_uint(y);
    int c = 0;
    if ((int)(bits >> 17) >= PQ_LUT_BASE) {
        const int i = (int)(bits >> 17) - PQ_LUT_BASE;
        const int c0 = (int)bnd[PQ_LUT_OFF + i], c1 = (int)bnd[PQ_LUT_OFF + i + 1];
        c = c0 + (int)(((uint32_t)(c1 - c0) * (bits & 0x1ffffu)) >> 17);
    }
    while (c < 65535 && y >= bnd[c + 1]) ++c;
    while (c > 0 && y < bnd[c]) --c;
    return (uint32_t)c;
}
__device__ __forceinline__ float gamut_row(float m0, float m1, float m2, float r, float g, float b)
{
    return __fmaf_rn(m2, b, __fmaf_rn(m1, g, __fmul_rn(m0, r)));      // the oracle's rounding sequence
}

template <typename T, bool PQ>
__global__ __launch_bounds__(256) void post_rgb48_kernel(const T *__restrict__ in, uint16_t *__restrict__ rgb,
                                                         size_t npix, float peak, const float *__restrict__ bnd)
{
    const size_t ngrp = npix / 8;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngrp; g += (size_t)gridDim.x * blockDim.x) {
        float r[8], gg[8], b[8];
        load8<T>(in + g * 8, r);
        load8<T>(in + npix + g * 8, gg);
        load8<T>(in + 2 * npix + g * 8, b);
        uint32_t w[12];
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            uint32_t q[6];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float cr = r[i + k], cg = gg[i + k], cb = b[i + k];
                if (PQ) {
                    // ITU-R BT.2087 BT.709 -> BT.2020 (linear light)
                    const float xr = gamut_row(0.6274f, 0.3293f, 0.0433f, cr, cg, cb);
                    const float xg = gamut_row(0.0691f, 0.9195f, 0.0114f, cr, cg, cb);
                    const float xb = gamut_row(0.0164f, 0.0880f, 0.8956f, cr, cg, cb);
                    const float c3[3] = {xr, xg, xb};
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) q[k * 3 + ch] = pq_code(fminf(fmaxf(c3[ch], 0.f), 1.f), peak, bnd);
                } else {
                    q[k * 3 + 0] = quant_u16(cr);
                    q[k * 3 + 1] = quant_u16(cg);
                    q[k * 3 + 2] = quant_u16(cb);
                }
            }
            w[(i / 2) * 3 + 0] = q[0] | (q[1] << 16);
            w[(i / 2) * 3 + 1] = q[2] | (q[3] << 16);
            w[(i / 2) * 3 + 2] = q[4] | (q[5] << 16);
        }
        uint4 *dst = reinterpret_cast<uint4 *>(rgb + g * 24);
        dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
        dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
        dst[2] = make_uint4(w[8], w[9], w[10], w[11]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (npix & 7)) {
        const size_t i = ngrp * 8 + threadIdx.x;
        float c3[3] = {(float)in[i], (float)in[npix + i], (float)in[2 * npix + i]};
        if (PQ) {
            const float cr = c3[0], cg = c3[1], cb = c3[2];
            c3[0] = gamut_row(0.6274f, 0.3293f, 0.0433f, cr, cg, cb);
            c3[1] = gamut_row(0.0691f, 0.9195f, 0.0114f, cr, cg, cb);
            c3[2] = gamut_row(0.0164f, 0.0880f, 0.8956f, cr, cg, cb);
        }
        for (int ch = 0; ch < 3; ++ch) {
            if (PQ) {
                rgb[i * 3 + ch] = (uint16_t)pq_code(fminf(fmaxf(c3[ch], 0.f), 1.f), peak, bnd);
            } else {
                rgb[i * 3 + ch] = (uint16_t)quant_u16(c3[ch]);
            }
        }
    }
}

inline int ew_grid(size_t ngrp)
{
    size_t g = (ngrp + 255) / 256;
    if (g > 2048) g = 2048;   // 256 CUs x 8 blocks, grid-stride beyond (guide: Guideline 11)
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

hipError_t pre_unpack_launch(const uint8_t *bgr, f16 *out, int H, int W, hipStream_t s)
{
    const size_t npix = (size_t)H * W;
    hipLaunchKernelGGL(pre_unpack_kernel, dim3(ew_grid(npix / 8)), dim3(256), 0, s, bgr, out, npix);
    return hipGetLastError();
}

hipError_t cond_resize_launch(const f16 *in, f16 *out, int H, int W, int Ho, int Wo, const float *wx, const int *xmn,
                              const int *xns, const float *wy, const int *ymn, const int *yns, hipStream_t s)
{
    dim3 grid((Wo + RT_W - 1) / RT_W, (Ho + RT_H - 1) / RT_H, 3);
    hipLaunchKernelGGL(cond_resize_kernel, grid, dim3(256), 0, s, in, out, H, W, Ho, Wo, wx, xmn, xns, wy, ymn, yns);
    return hipGetLastError();
}

// mode 0: 0.25x antialiased bicubic; 1: bilinear (fast_condition_resize); 2: zero condition.  Needs Ho = H / 4 >= 1, Wo = W / 4 >= 1.
hipError_t pre_fused_launch(const uint8_t *bgr, f16 *out, f16 *cond, int H, int W, int Ho, int Wo, const float *wx, const int *xmn,
                            const int *xns, const float *wy, const int *ymn, const int *yns, int mode, hipStream_t s)
{
    if (Ho != H / 4 || Wo != W / 4 || Ho < 1 || Wo < 1) return hipErrorInvalidValue;
    dim3 grid((Wo + PF_OW - 1) / PF_OW, (Ho + PF_OH - 1) / PF_OH, 1);
    hipLaunchKernelGGL(pre_fused_kernel, grid, dim3(256), 0, s, bgr, out, cond, H, W, Ho, Wo, wx, xmn, xns, wy, ymn, yns, mode);
    return hipGetLastError();
}

hipError_t post_u8_launch(const void *in, int is_f32, int H, int W, uint8_t *bgr, hipStream_t s)
{
    const size_t npix = (size_t)H * W;
    if (is_f32)
        hipLaunchKernelGGL(post_u8_kernel<float>, dim3(ew_grid(npix / 8)), dim3(256), 0, s, (const float *)in, bgr, npix);
    else
        hipLaunchKernelGGL(post_u8_kernel<f16>, dim3(ew_grid(npix / 8)), dim3(256), 0, s, (const f16 *)in, bgr, npix);
    return hipGetLastError();
}

// pq != 0 needs pq_bnd: device table of the 65536 code boundaries (entry 0 unused)
hipError_t post_rgb48_launch(const void *in, int is_f32, int H, int W, uint16_t *rgb, int pq, float peak, hipStream_t s,
                             const float *pq_bnd)
{
    if (pq && !pq_bnd) return hipErrorInvalidValue;
    const size_t npix = (size_t)H * W;
    const dim3 g(ew_grid(npix / 8)), b(256);
    if (is_f32) {
        if (pq) hipLaunchKernelGGL((post_rgb48_kernel<float, true>), g, b, 0, s, (const float *)in, rgb, npix, peak, pq_bnd);
        else hipLaunchKernelGGL((post_rgb48_kernel<float, false>), g, b, 0, s, (const float *)in, rgb, npix, peak, pq_bnd);
    } else {
        if (pq) hipLaunchKernelGGL((post_rgb48_kernel<f16, true>), g, b, 0, s, (const f16 *)in, rgb, npix, peak, pq_bnd);
        else hipLaunchKernelGGL((post_rgb48_kernel<f16, false>), g, b, 0, s, (const f16 *)in, rgb, npix, peak, pq_bnd);
    }
    return hipGetLastError();
}
