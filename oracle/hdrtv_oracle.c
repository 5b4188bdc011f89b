/*
 * hdrtv_oracle.c -- TEST INFRASTRUCTURE ONLY (the parity oracle), never shipped.
 *
 * Plain-C fp32 restatement of the tensor operators the reference's SDR->HDR hot
 * path calls.  The reference is Python on PyTorch: its arithmetic lives in ATen
 * (pin: torch==2.9.1+rocm7.2.1, requirements/requirements-amd.txt; container has
 * torch 2.10.0 CPU kernels), which is absent from /root/reference, so each
 * operator below restates ATen's published definition and is pinned by golden
 * vectors produced by running the reference itself (tests/golden/gen_golden.py).
 * The network graphs that compose these operators are in hdrtvnet_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  All tensors: batch 1, NCHW (planar), contiguous float32.
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XT 256 /* output-x tile held in the accumulator block */
#define CB 8   /* output channels sharing one input-row load   */

/* nn.Conv2d(Ci, Co, K, stride, padding=pad), zero padding, bias may be NULL.
 * Call sites: every Conv2d in Condition_arch.py:8-35,491-493, HDRUNet3T1_arch.py:14-55,
 * arch_util.py:63-66,78-84, Hallucination_arch.py:24-36,59-89.
 * Ho = (H + 2*pad - K)/stride + 1 (floor), same for Wo. */
void orc_conv2d(const float *x, int Ci, int H, int W, const float *w, const float *b, int Co,
                int K, int stride, int pad, float *y)
{
    const int Ho = (H + 2 * pad - K) / stride + 1;
    const int Wo = (W + 2 * pad - K) / stride + 1;
    const int ncb = (Co + CB - 1) / CB;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int oy = 0; oy < Ho; ++oy) {
        for (int cb = 0; cb < ncb; ++cb) {
            const int c0 = cb * CB, cn = (Co - c0 < CB) ? Co - c0 : CB;
            for (int x0 = 0; x0 < Wo; x0 += XT) {
                const int xn = (Wo - x0 < XT) ? Wo - x0 : XT;
                float acc[CB][XT];
                for (int c = 0; c < cn; ++c) {
                    const float bv = b ? b[c0 + c] : 0.0f;
                    for (int i = 0; i < xn; ++i) acc[c][i] = bv;
                }
                for (int ci = 0; ci < Ci; ++ci) {
                    for (int ky = 0; ky < K; ++ky) {
                        const int iy = oy * stride + ky - pad;
                        if (iy < 0 || iy >= H) continue;
                        const float *xr = x + ((size_t)ci * H + iy) * W;
                        for (int kx = 0; kx < K; ++kx) {
                            /* valid output range for this tap */
                            int lo = 0, hi = xn;
                            while (lo < hi && (x0 + lo) * stride + kx - pad < 0) ++lo;
                            while (hi > lo && (x0 + hi - 1) * stride + kx - pad >= W) --hi;
                            const float *xp = xr + (size_t)x0 * stride + kx - pad;
                            for (int c = 0; c < cn; ++c) {
                                const float wv = w[(((size_t)(c0 + c) * Ci + ci) * K + ky) * K + kx];
                                float *a = acc[c];
                                if (stride == 1) {
                                    for (int i = lo; i < hi; ++i) a[i] += wv * xp[i];
                                } else {
                                    for (int i = lo; i < hi; ++i) a[i] += wv * xp[(size_t)i * stride];
                                }
                            }
                        }
                    }
                }
                for (int c = 0; c < cn; ++c)
                    memcpy(y + ((size_t)(c0 + c) * Ho + oy) * Wo + x0, acc[c], sizeof(float) * xn);
            }
        }
    }
}

/* nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=True): always /9.
 * Condition_arch.py:10.  Ho = (H-1)/2 + 1. */
void orc_avgpool3s2p1(const float *x, int C, int H, int W, float *y)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
#pragma omp parallel for
    for (int c = 0; c < C; ++c)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                float s = 0.0f;
                for (int ky = -1; ky <= 1; ++ky) {
                    const int iy = 2 * oy + ky;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = -1; kx <= 1; ++kx) {
                        const int ix = 2 * ox + kx;
                        if (ix < 0 || ix >= W) continue;
                        s += x[((size_t)c * H + iy) * W + ix];
                    }
                }
                y[((size_t)c * Ho + oy) * Wo + ox] = s / 9.0f;
            }
}

/* nn.InstanceNorm2d(C, affine=True, eps=1e-5), eval with no running stats:
 * per-channel mean and BIASED variance over H*W.  Condition_arch.py:14. */
void orc_instnorm(const float *x, int C, int HW, const float *gamma, const float *beta, float eps,
                  float *y)
{
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        const float *p = x + (size_t)c * HW;
        double m = 0.0;
        for (int i = 0; i < HW; ++i) m += p[i];
        m /= HW;
        double v = 0.0;
        for (int i = 0; i < HW; ++i) v += (p[i] - m) * (p[i] - m);
        v /= HW;
        const float rstd = (float)(1.0 / sqrt(v + (double)eps));
        for (int i = 0; i < HW; ++i)
            y[(size_t)c * HW + i] = (float)(p[i] - m) * rstd * gamma[c] + beta[c];
    }
}

/* nn.BatchNorm2d eval: (x-mean)/sqrt(var+eps)*gamma+beta.  Hallucination_arch.py:26. */
void orc_batchnorm(const float *x, int C, int HW, const float *gamma, const float *beta,
                   const float *mean, const float *var, float eps, float *y)
{
#pragma omp parallel for
    for (int c = 0; c < C; ++c) {
        const float inv = 1.0f / sqrtf(var[c] + eps);
        for (int i = 0; i < HW; ++i)
            y[(size_t)c * HW + i] = (x[(size_t)c * HW + i] - mean[c]) * inv * gamma[c] + beta[c];
    }
}

/* nn.MaxPool2d(2).  Hallucination_arch.py:57.  Ho = H/2 (floor). */
void orc_maxpool2(const float *x, int C, int H, int W, float *y)
{
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for
    for (int c = 0; c < C; ++c)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const float *p = x + ((size_t)c * H + 2 * oy) * W + 2 * ox;
                float m = p[0];
                if (p[1] > m) m = p[1];
                if (p[W] > m) m = p[W];
                if (p[W + 1] > m) m = p[W + 1];
                y[((size_t)c * Ho + oy) * Wo + ox] = m;
            }
}

/* nn.PixelShuffle(2): out[c, 2h+i, 2w+j] = in[4c + 2i + j, h, w].
 * HDRUNet3T1_arch.py:31-33, Hallucination_arch.py:34. */
void orc_pixelshuffle2(const float *x, int Cin, int H, int W, float *y)
{
    const int Co = Cin / 4;
#pragma omp parallel for
    for (int c = 0; c < Co; ++c)
        for (int h = 0; h < H; ++h)
            for (int i = 0; i < 2; ++i)
                for (int w = 0; w < W; ++w)
                    for (int j = 0; j < 2; ++j)
                        y[((size_t)c * 2 * H + 2 * h + i) * 2 * W + 2 * w + j] =
                            x[((size_t)(4 * c + 2 * i + j) * H + h) * W + w];
}

/* ---- F.interpolate(scale_factor=0.25, mode="bicubic", align_corners=False,
 *      recompute_scale_factor=False, antialias=True)   hdrtvnet_torch.py:2278-2285
 * ATen _upsample_bicubic2d_aa, separable: width pass, then height pass; kernel
 * scale exactly 4, support 8, Keys cubic a=-0.5, taps normalised by their sum
 * (SURVEY.md appendix A.1).  out = floor(in/4). */
static float cubic_aa(float x)
{
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
    if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
    return 0.0f;
}

static void aa_weights(int in, int i, float scale, int *xmin_o, int *xsize_o, float *wt)
{
    const float support = 2.0f * scale;
    const float center = scale * ((float)i + 0.5f);
    int xmin = (int)(center - support + 0.5f);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5f);
    if (xmax > in) xmax = in;
    const int xs = xmax - xmin;
    float total = 0.0f;
    for (int j = 0; j < xs; ++j) {
        wt[j] = cubic_aa(((float)(j + xmin) - center + 0.5f) / scale);
        total += wt[j];
    }
    for (int j = 0; j < xs; ++j) wt[j] /= total;
    *xmin_o = xmin;
    *xsize_o = xs;
}

void orc_bicubic_aa_quarter(const float *x, int C, int H, int W, float *y)
{
    const int Ho = H / 4 > 0 ? H / 4 : 1, Wo = W / 4 > 0 ? W / 4 : 1;
    const float scale = 4.0f;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)C * H * Wo);
#pragma omp parallel for
    for (int c = 0; c < C; ++c)
        for (int ox = 0; ox < Wo; ++ox) {
            float wt[20];
            int xmin, xs;
            aa_weights(W, ox, scale, &xmin, &xs, wt);
            for (int yy = 0; yy < H; ++yy) {
                const float *p = x + ((size_t)c * H + yy) * W + xmin;
                float s = 0.0f;
                for (int j = 0; j < xs; ++j) s += wt[j] * p[j];
                tmp[((size_t)c * H + yy) * Wo + ox] = s;
            }
        }
#pragma omp parallel for
    for (int c = 0; c < C; ++c)
        for (int oy = 0; oy < Ho; ++oy) {
            float wt[20];
            int ymin, ys;
            aa_weights(H, oy, scale, &ymin, &ys, wt);
            for (int ox = 0; ox < Wo; ++ox) {
                float s = 0.0f;
                for (int j = 0; j < ys; ++j) s += wt[j] * tmp[((size_t)c * H + ymin + j) * Wo + ox];
                y[((size_t)c * Ho + oy) * Wo + ox] = s;
            }
        }
    free(tmp);
}

/* ---- pre/post quantisers ------------------------------------------------- */

/* hdrtvnet_torch.py:2256-2261: u8 HWC BGR -> planar RGB float, x * fp32(1/255). */
void orc_pre_unpack(const uint8_t *bgr, int H, int W, float *rgb_chw)
{
    const float k = (float)(1.0 / 255.0);
    const size_t HW = (size_t)H * W;
    for (size_t i = 0; i < HW; ++i)
        for (int c = 0; c < 3; ++c) rgb_chw[(size_t)c * HW + i] = (float)bgr[i * 3 + (2 - c)] * k;
}

static inline float clamp01(float v)
{
    /* torch.clamp: NaN propagates; min then max */
    if (v != v) return v;
    return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}

/* hdrtvnet_torch.py:2357-2361: clamp(0,1)*255+0.5 (two fp32 roundings), trunc to u8,
 * RGB planar -> BGR HWC. */
void orc_post_u8(const float *rgb_chw, int H, int W, uint8_t *bgr)
{
    const size_t HW = (size_t)H * W;
    for (size_t i = 0; i < HW; ++i)
        for (int c = 0; c < 3; ++c) {
            volatile float m = clamp01(rgb_chw[(size_t)c * HW + i]) * 255.0f;
            volatile float a = m + 0.5f;
            bgr[i * 3 + (2 - c)] = (uint8_t)a;
        }
}

/* gui_pipeline_worker_feeders.py:223-227: fp32 clamp(0,1), *65535 then +0.5 as two
 * separately rounded fp32 ops, float->u16 truncation, RGB planar -> RGB HWC (rgb48le). */
void orc_post_rgb48(const float *rgb_chw, int H, int W, uint16_t *rgb)
{
    const size_t HW = (size_t)H * W;
    for (size_t i = 0; i < HW; ++i)
        for (int c = 0; c < 3; ++c) {
            volatile float m = clamp01(rgb_chw[(size_t)c * HW + i]) * 65535.0f;
            volatile float a = m + 0.5f;
            rgb[i * 3 + c] = (uint16_t)a;
        }
}

/* ---- north-star display stages with no reference implementation on the playback
 * path.
 * PQ OETF: restated from gui_objective_metrics.py:486-491 (constants 63-67):
 *   y = clip(L/10000, 0, 1); ((c1 + c2*y^m1) / (1 + c3*y^m1))^m2, numpy float32.
 * u16 quantiser: gui_objective_metrics.py:531-539  clip(pq*65535+0.5, 0, 65535). */
#define PQ_M1 0.1593017578125f
#define PQ_M2 78.84375f
#define PQ_C1 0.8359375f
#define PQ_C2 18.8515625f
#define PQ_C3 18.6875f

float orc_pq_oetf(float nits)
{
    float y = nits / 10000.0f;
    y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
    const float yp = powf(y, PQ_M1);
    return powf((PQ_C1 + PQ_C2 * yp) / (1.0f + PQ_C3 * yp), PQ_M2);
}

/* BT.709 -> BT.2020 linear-light primaries conversion, ITU-R BT.2087-0 section 4 matrix
 * (PARITY UNPINNED: the reference has no gamut matrix anywhere; known-answer tested). */
static const float M709_2020[9] = {0.6274f, 0.3293f, 0.0433f, 0.0691f, 0.9195f, 0.0114f,
                                   0.0164f, 0.0880f, 0.8956f};

void orc_gamut709_2020(const float *rgb_chw, int H, int W, float *out_chw)
{
    const size_t HW = (size_t)H * W;
    for (size_t i = 0; i < HW; ++i) {
        const float r = rgb_chw[i], g = rgb_chw[HW + i], b = rgb_chw[2 * HW + i];
        for (int c = 0; c < 3; ++c)
            out_chw[(size_t)c * HW + i] = M709_2020[3 * c] * r + M709_2020[3 * c + 1] * g + M709_2020[3 * c + 2] * b;
    }
}

/* The u16 code of a PQ signal level, as an INTEGER function of the fp32 argument y = clip(L / 10000, 0, 1): the OETF
 * evaluated in double precision, code = floor(pq(y) * 65535 + 0.5).  fp32 powf differs between libraries by an ulp or two,
 * which moves one code in a few thousand; in double the result is the correctly rounded code for every fp32 y (the OETF
 * is monotone and the fp32 grid is ~1e8 times coarser than double rounding).  The device reproduces this function
 * exactly from a table of the 65535 code boundaries (csrc/hdrtv_api.hip pq_boundaries). */
uint16_t orc_pq_code(float y)
{
    const double yp = pow((double)y, 0.1593017578125);
    const double v = pow((0.8359375 + 18.8515625 * yp) / (1.0 + 18.6875 * yp), 78.84375);
    const double q = floor(v * 65535.0 + 0.5);
    return (uint16_t)(q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q));
}

/* Display post-process: linear-light BT.709 in [0,1] (1.0 = peak_nits) -> BT.2020 ->
 * PQ -> u16 RGB HWC.  fp32 with every rounding spelled out: lin = fma(m2, b, fma(m1, g, m0 * r)), clipped to [0, 1];
 * y = clip((lin * peak) / 10000, 0, 1); code = orc_pq_code(y). */
void orc_post_pq_rgb48(const float *rgb_chw, int H, int W, float peak_nits, uint16_t *rgb)
{
    const size_t HW = (size_t)H * W;
    for (size_t i = 0; i < HW; ++i) {
        const float r = rgb_chw[i], g = rgb_chw[HW + i], b = rgb_chw[2 * HW + i];
        for (int c = 0; c < 3; ++c) {
            volatile float m0 = M709_2020[3 * c] * r;
            float lin = fmaf(M709_2020[3 * c + 2], b, fmaf(M709_2020[3 * c + 1], g, m0));
            lin = lin < 0.0f ? 0.0f : (lin > 1.0f ? 1.0f : lin);
            volatile float nits = lin * peak_nits;
            float y = nits / 10000.0f;
            y = y < 0.0f ? 0.0f : (y > 1.0f ? 1.0f : y);
            rgb[i * 3 + c] = orc_pq_code(y);
        }
    }
}

/* ---- INT8 fake-quant (config 5), hdrtvnet_torch.py:351-364.
 * asymmetric: x_q = clamp(round((x - zero)/scale), 0, 255); x_deq = x_q*scale + zero
 * symmetric : x_q = clamp(round(x/scale), -128, 127);       x_deq = x_q*scale
 * torch.round = round-half-to-even = nearbyintf under the default rounding mode. */
void orc_fake_quant_act(const float *x, size_t n, float scale, float zero, int asymmetric, float *y)
{
    for (size_t i = 0; i < n; ++i) {
        if (asymmetric) {
            float q = nearbyintf((x[i] - zero) / scale);
            q = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);
            y[i] = q * scale + zero;
        } else {
            float q = nearbyintf(x[i] / scale);
            q = q < -128.0f ? -128.0f : (q > 127.0f ? 127.0f : q);
            y[i] = q * scale;
        }
    }
}
