// letterbox.hip -- the host step in front of preprocess, on the GPU: aspect-preserving resize of a u8 BGR
// frame onto a black canvas (_letterbox_bgr, src/gui_scaling.py:228-244: cv2.resize with INTER_AREA when
// shrinking, INTER_CUBIC when enlarging).  Arithmetic = oracle/letterbox_oracle.py, which restates OpenCV's
// 8-bit algorithms (PARITY UNPINNED against cv2 itself: OpenCV is not part of the reference tree).
// One thread per destination pixel; HBM-bound (reads <= area x 3 B, writes 3 B per pixel).
#include "launchers.h"

// The oracle multiplies and adds separately in float32.  This file is compiled with -ffp-contract=off (csrc/Makefile):
// hipcc fuses a*b+c by default and __fmul_rn / __fadd_rn are plain operators to it.

namespace {

__device__ __forceinline__ uint8_t sat_u8_rint(float v)
{
    const float r = rintf(v);                       // cvRound: half to even
    return (uint8_t)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

__global__ __launch_bounds__(256) void letterbox_kernel(LetterboxParams p)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= p.dw) return;
    uint8_t *d = p.dst + ((size_t)y * p.dw + x) * 3;
    const int rx = x - p.x0, ry = y - p.y0;
    if (rx < 0 || ry < 0 || rx >= p.new_w || ry >= p.new_h) { d[0] = d[1] = d[2] = 0; return; }
    const uint8_t *S = p.src;
    const size_t pitch = (size_t)p.sw * 3;
    if (p.mode == LB_COPY) {
        const uint8_t *s = S + (size_t)ry * pitch + (size_t)rx * 3;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    } else if (p.mode == LB_AREA_INT) {
        int sum[3] = {0, 0, 0};
        for (int j = 0; j < p.iy; ++j) {
            const uint8_t *s = S + (size_t)(ry * p.iy + j) * pitch + (size_t)rx * p.ix * 3;
            for (int i = 0; i < p.ix; ++i, s += 3) { sum[0] += s[0]; sum[1] += s[1]; sum[2] += s[2]; }
        }
        if (p.ix == 2 && p.iy == 2) {
            for (int c = 0; c < 3; ++c) d[c] = (uint8_t)((sum[c] + 2) >> 2);
        } else {
            const float scale = __fdiv_rn(1.0f, (float)(p.ix * p.iy));
            for (int c = 0; c < 3; ++c) d[c] = sat_u8_rint(__fmul_rn((float)sum[c], scale));
        }
    } else if (p.mode == LB_AREA_FRAC) {
        // tables: per destination index a run [beg, beg+cnt) of (source index, weight); float32, no FMA contraction,
        // accumulation in table order -- exactly the oracle's sequence
        const int xb = p.xbeg[rx], xn = p.xbeg[rx + 1] - xb, yb = p.ybeg[ry], yn = p.ybeg[ry + 1] - yb;
        float sum[3] = {0.f, 0.f, 0.f};
        for (int j = 0; j < yn; ++j) {
            const uint8_t *row = S + (size_t)p.ysrc[yb + j] * pitch;
            const float beta = p.yw[yb + j];
            float buf[3] = {0.f, 0.f, 0.f};
            for (int i = 0; i < xn; ++i) {
                const uint8_t *s = row + (size_t)p.xsrc[xb + i] * 3;
                const float a = p.xw[xb + i];
                for (int c = 0; c < 3; ++c) buf[c] = __fadd_rn(buf[c], __fmul_rn((float)s[c], a));
            }
            for (int c = 0; c < 3; ++c) sum[c] = j == 0 ? __fmul_rn(beta, buf[c]) : __fadd_rn(sum[c], __fmul_rn(beta, buf[c]));
        }
        for (int c = 0; c < 3; ++c) d[c] = sat_u8_rint(sum[c]);
    } else {   // LB_CUBIC: 11-bit fixed-point Keys kernel (A = -0.75), horizontal then vertical, replicated borders
        const int xo = p.xsrc[rx], yo = p.ysrc[ry];
        long long acc[3] = {0, 0, 0};
        for (int k = 0; k < 4; ++k) {
            int sy = yo + k;
            sy = sy < 0 ? 0 : (sy > p.sh - 1 ? p.sh - 1 : sy);
            const uint8_t *row = S + (size_t)sy * pitch;
            long long hor[3] = {0, 0, 0};
            for (int i = 0; i < 4; ++i) {
                int sx = xo + i;
                sx = sx < 0 ? 0 : (sx > p.sw - 1 ? p.sw - 1 : sx);
                const int cx = p.xc[rx * 4 + i];
                const uint8_t *s = row + (size_t)sx * 3;
                for (int c = 0; c < 3; ++c) hor[c] += (long long)s[c] * cx;
            }
            const int cy = p.yc[ry * 4 + k];
            for (int c = 0; c < 3; ++c) acc[c] += hor[c] * cy;
        }
        for (int c = 0; c < 3; ++c) {
            const long long v = (acc[c] + (1ll << 21)) >> 22;
            d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

}  // namespace

hipError_t letterbox_launch(const LetterboxParams &p, hipStream_t s)
{
    hipLaunchKernelGGL(letterbox_kernel, dim3((p.dw + 255) / 256, p.dh), dim3(256), 0, s, p);
    return hipGetLastError();
}
