// precision="fp32" (HDRTVNetTorch._resolve_precision, hdrtvnet_torch.py:1694-1712): the reference's graph on fp32 tensors.
//
// The fp16 path is the product's speed path (hand-fused MFMA kernels, NHWC f16).  This file serves the reference's "maximum
// precision" preset: every tensor is planar CHW fp32 as in the reference, every layer is one generic kernel, the graph is
// hdrtv_api.hip: fp32_graph.inc.  fp32 has no fast matrix path on gfx950 (fp32 MFMA = vector rate, 157 TFLOP/s), so the
// convolution is a vector-FMA kernel: a lane owns one output pixel and COT output channels, the filter is read through
// the scalar cache (wave-uniform addresses -> s_load), one coalesced pixel load feeds COT FMAs.
#include "common.h"
#include "launchers.h"

namespace {

// ---------------------------------------------------------------- conv2d (k = 1 / 3, stride 1 / 2, zero padding)
// w: [cout group][cin][tap][COT]; bias / bn / gfm vectors padded to the group size.
// epilogue, in the reference's op order: v = conv + bias; BatchNorm2d(eval) v*bn_s + bn_t (Hallucination_arch.py:24-29);
// GFM v*s + t + v (Condition_arch.py:573-583); ReLU / LeakyReLU; + residual; store (optionally through PixelShuffle(2)).
template <int KS, int STRIDE, int COT>
__global__ __launch_bounds__(256) void conv_f32_kernel(F32ConvParams p)
{
    // a lane: one output pixel x COT output channels; a workgroup: 64 columns x 4 rows; blockIdx.z: the channel group.
    // The filter is packed in groups of p.cot (>= COT) channels: a COT-wide kernel reads its slice of a group.
    constexpr int KK = KS * KS;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ox = blockIdx.x * 64 + lane, oy = blockIdx.y * 4 + wv;
    const int co0 = blockIdx.z * COT;
    const bool live = ox < p.Wo && oy < p.Ho;
    int off[KK];
    bool ok[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        const int iy = oy * STRIDE + t / KS - p.pad, ix = ox * STRIDE + t % KS - p.pad;
        ok[t] = live && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        off[t] = ok[t] ? iy * p.Wi + ix : 0;
    }
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = 0.f;
    const size_t plane = (size_t)p.Hi * p.Wi;
    const int cin = p.c0 + p.c1, PW = p.cot;
    const float *__restrict__ w = p.w + (size_t)(co0 / PW) * cin * KK * PW + co0 % PW;
    auto plane_of = [&](int ci) { return ci < p.c0 ? p.x0 + (size_t)ci * plane : p.x1 + (size_t)(ci - p.c0) * plane; };
    for (int ci = 0; ci < cin; ++ci) {
        const float *__restrict__ xp = plane_of(ci);
        float xv[KK];
#pragma unroll
        for (int t = 0; t < KK; ++t) xv[t] = xp[off[t]];              // off = 0 where the tap is outside: always a valid address
#pragma unroll
        for (int t = 0; t < KK; ++t) xv[t] = ok[t] ? xv[t] : 0.f;
        // (loading channel ci + 1's taps under these FMAs by hand was 5 % slower: the compiler's own schedule and six waves
        // per SIMD already cover the latency)
        const float *__restrict__ wc = w + (size_t)ci * KK * PW;
#pragma unroll
        for (int t = 0; t < KK; ++t) {
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[j] = fmaf(xv[t], wc[t * PW + j], acc[j]);   // wave-uniform address: scalar loads
        }
    }
    if (!live) return;
    const size_t oplane = (size_t)p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = co0 + j;
        if (co >= p.cout) break;
        float v = __fadd_rn(acc[j], p.bias[co]);
        if (p.bn_s) v = __fadd_rn(__fmul_rn(v, p.bn_s[co]), p.bn_t[co]);
        if (p.gfm_s) v = __fadd_rn(__fadd_rn(__fmul_rn(v, p.gfm_s[co]), p.gfm_t[co]), v);
        if (p.act == 1) v = v > 0.f ? v : 0.f;
        else if (p.act == 2) v = v >= 0.f ? v : __fmul_rn(v, p.slope);
        if (p.res) v = __fadd_rn(p.res[(size_t)co * oplane + (size_t)oy * p.Wo + ox], v);
        if (p.ps) {
            const int c = co >> 2, dy = (co >> 1) & 1, dx = co & 1;
            p.y[((size_t)c * (2 * p.Ho) + 2 * oy + dy) * (size_t)(2 * p.Wo) + 2 * ox + dx] = v;
        } else {
            p.y[(size_t)co * oplane + (size_t)oy * p.Wo + ox] = v;
        }
    }
}

template <int KS, int STRIDE>
hipError_t conv_f32_pick(const F32ConvParams &p, int n_cu, hipStream_t s)
{
    // 32 channels per lane (one pixel load feeds 32 FMAs) unless that leaves fewer than `few` workgroups per CU (the
    // low-resolution layers of the HG head): then 8 per lane, four times the workgroups
    const int sp = ((p.Wo + 63) / 64) * ((p.Ho + 3) / 4);
    const int few = p.narrow_below > 0 ? p.narrow_below : 3;
    const int cot = p.cot == 32 && sp * ((p.cout + 31) / 32) >= few * n_cu ? 32 : 8;
    dim3 grid((p.Wo + 63) / 64, (p.Ho + 3) / 4, (p.cout + cot - 1) / cot);
    if (cot == 32) hipLaunchKernelGGL((conv_f32_kernel<KS, STRIDE, 32>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv_f32_kernel<KS, STRIDE, 8>), grid, dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------- element-wise, pooling, normalisation
__device__ inline size_t gtid() { return (size_t)blockIdx.x * blockDim.x + threadIdx.x; }
__device__ inline size_t gstride() { return (size_t)gridDim.x * blockDim.x; }

// op 0: y = a + b; op 1: SFTLayer (arch_util.py:68-72) y = a * (b + 1) + c
__global__ __launch_bounds__(256) void ew_f32_kernel(int op, const float *__restrict__ a, const float *__restrict__ b,
                                                     const float *__restrict__ c, float *__restrict__ y, size_t n)
{
    for (size_t i = gtid(); i < n; i += gstride()) {
        if (op == 0) y[i] = __fadd_rn(a[i], b[i]);
        else y[i] = __fadd_rn(__fmul_rn(a[i], __fadd_rn(b[i], 1.f)), c[i]);
    }
}

// AvgPool2d(3, stride 2, padding 1, count_include_pad) then LeakyReLU(slope) (Condition_arch.py:8-16)
__global__ __launch_bounds__(256) void avgpool3s2_leaky_kernel(const float *__restrict__ x, float *__restrict__ y, int C, int H,
                                                               int W, int Ho, int Wo, float slope)
{
    const size_t n = (size_t)C * Ho * Wo;
    for (size_t i = gtid(); i < n; i += gstride()) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), c = (int)(i / ((size_t)Wo * Ho));
        const float *xp = x + (size_t)c * H * W;
        float s = 0.f;
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * oy + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * ox + dx;
                if (ix >= 0 && ix < W) s = __fadd_rn(s, xp[(size_t)iy * W + ix]);
            }
        }
        const float v = __fdiv_rn(s, 9.f);
        y[i] = v >= 0.f ? v : __fmul_rn(v, slope);
    }
}

__device__ inline double block_sum(double v, double *sh)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}

// InstanceNorm2d(affine, eps, biased variance) in place; one workgroup per channel (Condition_arch.py:13-15)
__global__ __launch_bounds__(256) void instnorm_f32_kernel(float *__restrict__ x, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, int n, float eps)
{
    __shared__ double sh[4];
    float *xp = x + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)xp[i];
    const double mean = block_sum(s, sh) / n;
    double q = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const double d = (double)xp[i] - mean; q += d * d; }
    const double var = block_sum(q, sh) / n;
    const float m = (float)mean, r = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma[blockIdx.x], b = beta[blockIdx.x];
    for (int i = threadIdx.x; i < n; i += 256) xp[i] = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(xp[i], m), r), g), b);
}

// mean over the plane, one workgroup per channel (Color_Condition.forward's global average, Condition_arch.py:33-35)
__global__ __launch_bounds__(256) void plane_mean_f32_kernel(const float *__restrict__ x, float *__restrict__ y, int n)
{
    __shared__ double sh[4];
    const float *xp = x + (size_t)blockIdx.x * n;
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)xp[i];
    const double t = block_sum(s, sh);
    if (threadIdx.x == 0) y[blockIdx.x] = (float)(t / n);
}

// the six GFM heads (Condition_arch.py:566-583): out[slot*64 + o] = W_slot[o,:] . fea + b_slot[o]; slots padded with zeros
__global__ __launch_bounds__(64) void gfm_heads_f32_kernel(F32GfmParams p)
{
    const int slot = blockIdx.x, o = threadIdx.x;
    float v = 0.f;
    if (o < p.n[slot]) {
        v = p.b[slot][o];
        for (int k = 0; k < 6; ++k) v = fmaf(p.w[slot][o * 6 + k], p.fea[k], v);
    }
    p.out[slot * 64 + o] = v;
}

__global__ __launch_bounds__(256) void maxpool2_f32_kernel(const float *__restrict__ x, float *__restrict__ y, int C, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;
    const size_t n = (size_t)C * Ho * Wo;
    for (size_t i = gtid(); i < n; i += gstride()) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), c = (int)(i / ((size_t)Wo * Ho));
        const float *xp = x + ((size_t)c * H + 2 * oy) * W + 2 * ox;
        y[i] = fmaxf(fmaxf(xp[0], xp[1]), fmaxf(xp[W], xp[W + 1]));
    }
}

// general planar window copy: y[c][oy][ox] = x[c][map(oy)][map(ox)]
//   mode 0: HDRUNet3T1._align_to (HDRUNet3T1_arch.py:79-104): centre crop, then replicate padding split floor / ceil
//   mode 1: F.pad(mode="reflect") on the bottom / right (HG_Composite_arch.py:97-103), or a plain crop when Ho <= H
__global__ __launch_bounds__(256) void window_f32_kernel(const float *__restrict__ x, float *__restrict__ y, int C, int H, int W,
                                                         int Ho, int Wo, int mode)
{
    const size_t n = (size_t)C * Ho * Wo;
    const int top = H > Ho ? (H - Ho) / 2 : 0, left = W > Wo ? (W - Wo) / 2 : 0;
    const int pt = H < Ho ? (Ho - H) / 2 : 0, pl = W < Wo ? (Wo - W) / 2 : 0;
    for (size_t i = gtid(); i < n; i += gstride()) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), c = (int)(i / ((size_t)Wo * Ho));
        int iy, ix;
        if (mode == 0) {
            iy = oy + top - pt; ix = ox + left - pl;
            iy = iy < 0 ? 0 : (iy > H - 1 ? H - 1 : iy);
            ix = ix < 0 ? 0 : (ix > W - 1 ? W - 1 : ix);
        } else {
            iy = oy < H ? oy : 2 * (H - 1) - oy;
            ix = ox < W ? ox : 2 * (W - 1) - ox;
        }
        y[i] = x[((size_t)c * H + iy) * W + ix];
    }
}

// HG_Composite._make_mask (HG_Composite_arch.py:78-84): m = max_c(base); ((m - r) / (1 - r)).clamp(0, 1) > 0.1 -> 0 / 1
__global__ __launch_bounds__(256) void hg_mask_f32_kernel(const float *__restrict__ base, float *__restrict__ mask, size_t npix, float r)
{
    const float den = __fsub_rn(1.f, r);
    for (size_t i = gtid(); i < npix; i += gstride()) {
        const float m = fmaxf(fmaxf(base[i], base[npix + i]), base[2 * npix + i]);
        float v = __fdiv_rn(__fsub_rn(m, r), den);
        v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
        mask[i] = v > 0.1f ? 1.f : 0.f;
    }
}

// Hallucination_Generator.forward's last line + the crop (Hallucination_arch.py:135-137, HG_Composite_arch.py:104-106):
// out[c][y][x] = mask[y][x] * t[c][y][x] + img[c][y][x] on the padded grid, stored for y < H, x < W
__global__ __launch_bounds__(256) void hg_blend_f32_kernel(const float *__restrict__ t, const float *__restrict__ img,
                                                           const float *__restrict__ mask, float *__restrict__ out, int H, int W,
                                                           int Hp, int Wp)
{
    const size_t n = (size_t)3 * H * W;
    for (size_t i = gtid(); i < n; i += gstride()) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((size_t)W * H));
        const size_t j = ((size_t)c * Hp + y) * Wp + x;
        out[i] = __fadd_rn(__fmul_rn(mask[(size_t)y * Wp + x], t[j]), img[j]);
    }
}

// ---------------------------------------------------------------- preprocess (hdrtvnet_torch.py:2238-2296, dtype = fp32)
// u8 BGR HWC -> planar RGB fp32, float(u8) * fp32(1/255)
__global__ __launch_bounds__(256) void pre_unpack_f32_kernel(const uint8_t *__restrict__ bgr, float *__restrict__ out, size_t npix)
{
    const float k255 = (float)(1.0 / 255.0);
    for (size_t i = gtid(); i < npix; i += gstride()) {
        const uint8_t *p = bgr + 3 * i;
        out[i] = __fmul_rn((float)p[2], k255);
        out[npix + i] = __fmul_rn((float)p[1], k255);
        out[2 * npix + i] = __fmul_rn((float)p[0], k255);
    }
}

// 0.25x condition map.  mode 0: bicubic with antialiasing, ATen's separable form -- horizontal pass (rounded to fp32 per
// row), then vertical, taps accumulated in index order; mode 1: bilinear (mean of pixels 4d+1, 4d+2 per direction, ATen's
// operation order); mode 2: zeros.  Tap tables as in prepost.hip (17 per output).
__global__ __launch_bounds__(256) void cond_resize_f32_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W,
                                                              int Ho, int Wo, const float *__restrict__ wx,
                                                              const int *__restrict__ xmn, const int *__restrict__ xns,
                                                              const float *__restrict__ wy, const int *__restrict__ ymn,
                                                              const int *__restrict__ yns, int mode)
{
    const size_t n = (size_t)3 * Ho * Wo;
    for (size_t i = gtid(); i < n; i += gstride()) {
        const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), c = (int)(i / ((size_t)Wo * Ho));
        const float *src = in + (size_t)c * H * W;
        float v = 0.f;
        if (mode == 0) {
            const int xb = xmn[ox], xn = xns[ox], yb = ymn[oy], yn = yns[oy];
            const float *wxx = wx + (size_t)ox * 17, *wyy = wy + (size_t)oy * 17;
            for (int r = 0; r < yn; ++r) {
                const float *row = src + (size_t)(yb + r) * W + xb;
                float h = 0.f;
                for (int j = 0; j < xn; ++j) h = __fadd_rn(h, __fmul_rn(wxx[j], row[j]));
                v = __fadd_rn(v, __fmul_rn(wyy[r], h));
            }
        } else if (mode == 1) {
            const int y1 = 4 * oy + 1, x1 = 4 * ox + 1;
            const int y2 = y1 + 1 < H ? y1 + 1 : H - 1, x2 = x1 + 1 < W ? x1 + 1 : W - 1;
            const float top = __fadd_rn(__fmul_rn(0.5f, src[(size_t)y1 * W + x1]), __fmul_rn(0.5f, src[(size_t)y1 * W + x2]));
            const float bot = __fadd_rn(__fmul_rn(0.5f, src[(size_t)y2 * W + x1]), __fmul_rn(0.5f, src[(size_t)y2 * W + x2]));
            v = __fadd_rn(__fmul_rn(0.5f, top), __fmul_rn(0.5f, bot));
        }
        out[i] = v;
    }
}

inline dim3 ew_grid_f32(size_t n)
{
    size_t g = (n + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

}  // namespace

hipError_t conv_f32_launch(const F32ConvParams &p, int ks, int stride, int n_cu, hipStream_t s)
{
    if (p.Ho <= 0 || p.Wo <= 0 || p.cout <= 0 || p.c0 <= 0 || (p.cot != 8 && p.cot != 32)) return hipErrorInvalidValue;
    if (ks == 1 && stride == 1) return conv_f32_pick<1, 1>(p, n_cu, s);
    if (ks == 3 && stride == 1) return conv_f32_pick<3, 1>(p, n_cu, s);
    if (ks == 3 && stride == 2) return conv_f32_pick<3, 2>(p, n_cu, s);
    if (ks == 1 && stride == 2) return conv_f32_pick<1, 2>(p, n_cu, s);
    return hipErrorInvalidValue;
}

hipError_t ew_f32_launch(int op, const float *a, const float *b, const float *c, float *y, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(ew_f32_kernel, ew_grid_f32(n), dim3(256), 0, s, op, a, b, c, y, n);
    return hipGetLastError();
}

hipError_t avgpool3s2_leaky_f32_launch(const float *x, float *y, int C, int H, int W, float slope, hipStream_t s)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(avgpool3s2_leaky_kernel, ew_grid_f32((size_t)C * Ho * Wo), dim3(256), 0, s, x, y, C, H, W, Ho, Wo, slope);
    return hipGetLastError();
}

hipError_t instnorm_f32_launch(float *x, const float *gamma, const float *beta, int C, int n, float eps, hipStream_t s)
{
    hipLaunchKernelGGL(instnorm_f32_kernel, dim3(C), dim3(256), 0, s, x, gamma, beta, n, eps);
    return hipGetLastError();
}

hipError_t plane_mean_f32_launch(const float *x, float *y, int C, int n, hipStream_t s)
{
    hipLaunchKernelGGL(plane_mean_f32_kernel, dim3(C), dim3(256), 0, s, x, y, n);
    return hipGetLastError();
}

hipError_t gfm_heads_f32_launch(const F32GfmParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(gfm_heads_f32_kernel, dim3(6), dim3(64), 0, s, p);
    return hipGetLastError();
}

hipError_t maxpool2_f32_launch(const float *x, float *y, int C, int H, int W, hipStream_t s)
{
    hipLaunchKernelGGL(maxpool2_f32_kernel, ew_grid_f32((size_t)C * (H / 2) * (W / 2)), dim3(256), 0, s, x, y, C, H, W);
    return hipGetLastError();
}

hipError_t window_f32_launch(const float *x, float *y, int C, int H, int W, int Ho, int Wo, int mode, hipStream_t s)
{
    hipLaunchKernelGGL(window_f32_kernel, ew_grid_f32((size_t)C * Ho * Wo), dim3(256), 0, s, x, y, C, H, W, Ho, Wo, mode);
    return hipGetLastError();
}

hipError_t hg_mask_f32_launch(const float *base, float *mask, size_t npix, float r, hipStream_t s)
{
    hipLaunchKernelGGL(hg_mask_f32_kernel, ew_grid_f32(npix), dim3(256), 0, s, base, mask, npix, r);
    return hipGetLastError();
}

hipError_t hg_blend_f32_launch(const float *t, const float *img, const float *mask, float *out, int H, int W, int Hp, int Wp,
                               hipStream_t s)
{
    hipLaunchKernelGGL(hg_blend_f32_kernel, ew_grid_f32((size_t)3 * H * W), dim3(256), 0, s, t, img, mask, out, H, W, Hp, Wp);
    return hipGetLastError();
}

hipError_t pre_f32_launch(const uint8_t *bgr, float *rgb, float *cond, int H, int W, int Ho, int Wo, const float *wx,
                          const int *xmn, const int *xns, const float *wy, const int *ymn, const int *yns, int mode, hipStream_t s)
{
    hipLaunchKernelGGL(pre_unpack_f32_kernel, ew_grid_f32((size_t)H * W), dim3(256), 0, s, bgr, rgb, (size_t)H * W);
    hipLaunchKernelGGL(cond_resize_f32_kernel, ew_grid_f32((size_t)3 * Ho * Wo), dim3(256), 0, s, rgb, cond, H, W, Ho, Wo, wx, xmn,
                       xns, wy, ymn, yns, mode);
    return hipGetLastError();
}
