// precision="fp32" (HDRTVNetTorch._resolve_precision, hdrtvnet_torch.py:1694-1712): the reference's graph on fp32 tensors.
//
// The fp16 path is the product's speed path (hand-fused MFMA kernels, NHWC f16).  This file serves the reference's "maximum
// precision" preset: every tensor is planar CHW fp32 as in the reference, every layer is one generic kernel, the graph is
// fp32_graph.hip.  Two convolution kernels:
//   * conv_f32_mfma (round 5): the 3x3 and 1x1 / stride-1 layers whose channel counts are multiples of 32 / 16 -- 97 % of the MACs -- as an
//     implicit GEMM on the fp32 MATRIX pipe, v_mfma_f32_32x32x2_f32: exact fp32 products and sums (an fmaf chain per output
//     element, MI355X_MICROARCH.md "Matrix cores"), 64 FLOP / clock / SIMD = the vector rate, but one instruction per 4096
//     FLOPs instead of per 128, so the issue slots, the scalar cache and the register file stop being the limit;
//   * conv_f32 (round 4): everything else (1x1, stride 2, 3 input or output channels, maps too small to fill the chip) as a
//     vector-FMA kernel: a lane owns one output pixel and COT output channels, the filter arrives through the scalar cache.
// Both accumulate in the order (input channel, tap) from zero and add the bias behind the sum.
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

// ---------------------------------------------------------------- conv2d 3x3 / stride 1 / pad 1 on the fp32 matrix pipe
// GEMM view: M = 32 MT output channels, N = a tile of 8 rows x 32 columns (wave w owns row w: one 32-pixel N tile and MT
// accumulator tiles), K = (input channel, tap) in that order, 2 per v_mfma_f32_32x32x2_f32: lane (n, kk) of a wave feeds
// B[k0 + kk][n] = the pixel of column n for tap k0 + kk, and A[m][k0 + kk] = the weight of channel m -- the host's layout
// [cout group][cin][tap][32] IS that fragment order, so a chunk's weights are one contiguous copy.  Input channels go through
// LDS in chunks of 16 (halo tile 16 x 10 x 34 floats + MT x 144 x 32 weights, double-buffered, one barrier per chunk): the next
// chunk's global loads are issued in front of this chunk's 144 MT MFMAs and written to LDS behind them.
// The 1x1 layers (KS = 1: the condition nets' 64 -> 64 convs over the full-resolution map, the HG head's 1x1 fuse convs over a
// channel concat) are the same GEMM without a halo, in chunks of 32 input channels (16 MFMAs per chunk and accumulator tile).
constexpr int MF_TR = 8, MF_TC = 32;
template <int MT, int KS> struct MfGeo {
    static constexpr int TAPS = KS * KS, CIB = KS == 3 ? 16 : 32;    // input channels per chunk
    static constexpr int HR = MF_TR + KS - 1, HC = MF_TC + KS - 1;
    static constexpr int HALO = CIB * HR * HC;                        // 5440 / 8192 floats
    static constexpr int KQ = CIB * TAPS / 2;                         // 72 / 16 MFMAs per chunk and accumulator tile
    static constexpr int WCH = MT * CIB * TAPS * 32;                  // floats of a chunk's weights
    static constexpr int SMEM = 2 * (HALO + WCH) * 4;                 // KS = 3: 117 248 B (MT = 2) / 80 384 B; KS = 1: 81 920 B / 73 728 B
    static constexpr int NH = (HALO + 511) / 512, NW4 = (WCH / 4 + 511) / 512;
};

template <int MT, int KS>
__global__ __launch_bounds__(512) void conv_f32_mfma_kernel(F32ConvParams p)
{
    using G = MfGeo<MT, KS>;
    constexpr int MF_CIB = G::CIB, MF_HR = G::HR, MF_HC = G::HC, MF_HALO = G::HALO, MF_KQ = G::KQ, TAPS = G::TAPS;
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float *const sX = smf, *const sW = smf + 2 * MF_HALO;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, kk = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ox0 = blockIdx.x * MF_TC, oy0 = blockIdx.y * MF_TR, g0 = blockIdx.z * MT;
    const int cin = p.c0 + p.c1, nch = cin / MF_CIB;
    const size_t plane = (size_t)p.Hi * p.Wi;
    // ---- staging: this thread's elements of a chunk (halo: element e = ((channel, row), column); weights: float4 number e)
    int hoff[G::NH];                       // offset inside a plane, or -1 outside the image (zero padding)
    int hcl[G::NH];
#pragma unroll
    for (int i = 0; i < G::NH; ++i) {
        const int e = tid + 512 * i;
        const int cl = e / (MF_HR * MF_HC), r = (e - cl * (MF_HR * MF_HC)) / MF_HC, c = e - cl * (MF_HR * MF_HC) - r * MF_HC;
        const int iy = oy0 - KS / 2 + r, ix = ox0 - KS / 2 + c;
        hcl[i] = cl;
        hoff[i] = (e < MF_HALO && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) ? iy * p.Wi + ix : -1;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    const int xb = wv * MF_HC + n;                                   // this lane's pixel in row `wv` of the halo tile (tap 0)
    // pass ch = -1 only stages chunk 0; pass ch fetches chunk ch + 1 into registers, runs chunk ch's MFMAs, writes the registers to
    // the other buffer (whose readers all passed the previous barrier)
    for (int ch = -1; ch < nch; ++ch) {
        const int buf = ch & 1, nx = ch + 1;
        float hv[G::NH];
        f32x4 wv4[G::NW4];                                       // (clang vectors: arrays of HIP's float4 struct end up in scratch)
        if (nx < nch) {
#pragma unroll
            for (int i = 0; i < G::NH; ++i) {
                const int ci = nx * MF_CIB + hcl[i];
                const float *pl = ci < p.c0 ? p.x0 + (size_t)ci * plane : p.x1 + (size_t)(ci - p.c0) * plane;
                hv[i] = hoff[i] >= 0 ? pl[hoff[i]] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < G::NW4; ++i) {
                int e = tid + 512 * i;                                // float4 index inside the chunk: [mt][1152]
                e = e < G::WCH / 4 ? e : 0;                           // (past the end: any valid address; not staged)
                const int mt = e / (MF_CIB * TAPS * 8), r = e - mt * (MF_CIB * TAPS * 8);
                wv4[i] = *reinterpret_cast<const f32x4 *>(p.w + ((size_t)(g0 + mt) * cin + (size_t)nx * MF_CIB) * (TAPS * 32) + (size_t)r * 4);
            }
        }
        if (ch >= 0) {
            const float *bx = sX + buf * MF_HALO + xb, *bw = sW + buf * G::WCH + lane;
#pragma unroll
            for (int q = 0; q < MF_KQ; ++q) {
                // k = 2 q + kk -> (channel, tap): compile-time for either lane half
                const int k0 = 2 * q, k1 = 2 * q + 1;
                const int o0 = ((k0 / TAPS) * MF_HR + (k0 % TAPS) / KS) * MF_HC + (k0 % TAPS) % KS, o1 = ((k1 / TAPS) * MF_HR + (k1 % TAPS) / KS) * MF_HC + (k1 % TAPS) % KS;
                const float b = bx[kk ? o1 : o0];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[(mt * MF_KQ + q) * 64], b, acc[mt], 0, 0, 0);
            }
        }
        if (nx < nch) {
            const int nb = nx & 1;
#pragma unroll
            for (int i = 0; i < G::NH; ++i) { const int e = tid + 512 * i; if (e < MF_HALO) sX[nb * MF_HALO + e] = hv[i]; }
#pragma unroll
            for (int i = 0; i < G::NW4; ++i) { const int e = tid + 512 * i; if (e < G::WCH / 4) reinterpret_cast<f32x4 *>(sW + nb * G::WCH)[e] = wv4[i]; }
        }
        __syncthreads();
    }
    // ---- epilogue, in the reference's op order (see conv_f32_kernel): register r of tile mt = channel 32 (g0 + mt) + 8 (r >> 2) + 4 kk + (r & 3)
    const int oy = oy0 + wv, ox = ox0 + n;
    if (oy >= p.Ho || ox >= p.Wo) return;
    const size_t oplane = (size_t)p.Ho * p.Wo;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (g0 + mt) * 32 + 8 * (r >> 2) + 4 * kk + (r & 3);
            float v = __fadd_rn(acc[mt][r], p.bias[co]);
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

template <int MT, int KS> hipError_t conv_f32_mfma_go(const F32ConvParams &p, hipStream_t s)
{
    static DevOnce once;
    auto kern = conv_f32_mfma_kernel<MT, KS>;
    constexpr int smem = MfGeo<MT, KS>::SMEM;
    if (once.need()) {
        if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem)) return e;
        once.done();
    }
    hipLaunchKernelGGL(kern, dim3((p.Wo + MF_TC - 1) / MF_TC, (p.Ho + MF_TR - 1) / MF_TR, p.cout / (32 * MT)), dim3(512), smem, s, p);
    return hipGetLastError();
}
// two 32-channel groups per workgroup (each pixel fragment feeds two MFMAs) while that still leaves two workgroups per CU
template <int KS> hipError_t conv_f32_mfma_pick(const F32ConvParams &p, int n_cu, hipStream_t s)
{
    const long tiles = (long)((p.Wo + MF_TC - 1) / MF_TC) * ((p.Ho + MF_TR - 1) / MF_TR);
    return (p.cout % 64 == 0 && tiles * (p.cout / 64) >= 2L * n_cu) ? conv_f32_mfma_go<2, KS>(p, s) : conv_f32_mfma_go<1, KS>(p, s);
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

// which layers run on the matrix pipe: 3x3 or 1x1 / stride 1 / same size, filter packed in groups of 32, whole groups of output channels and
// whole chunks of input channels on either side of a concat, at least 64 input channels, and enough tiles to give at least a
// quarter of the CUs a workgroup
bool conv_f32_on_mfma(const F32ConvParams &p, int ks, int n_cu)
{
    const int cib = ks == 3 ? 16 : 32;
    if ((ks != 1 && ks != 3) || p.no_mfma || p.pad != ks / 2 || p.cot != 32 || p.cout % 32 || p.c0 % cib || p.c1 % cib || p.Hi != p.Ho || p.Wi != p.Wo) return false;
    // 32 input channels are two chunks: prologue, barriers and epilogue then weigh as much as the MFMAs (LE's 32 -> 32 layers measured
    // 52 TFLOP/s here against 74 on the vector kernel, profiles/r05_fp32_layers.txt)
    if (p.c0 + p.c1 < 64) return false;
    const long tiles = (long)((p.Wo + MF_TC - 1) / MF_TC) * ((p.Ho + MF_TR - 1) / MF_TR) * (p.cout / 32);
    return tiles * 4 >= n_cu;
}

hipError_t conv_f32_launch(const F32ConvParams &p, int ks, int stride, int n_cu, hipStream_t s)
{
    if (p.Ho <= 0 || p.Wo <= 0 || p.cout <= 0 || p.c0 <= 0 || (p.cot != 8 && p.cot != 32)) return hipErrorInvalidValue;
    if (stride == 1 && conv_f32_on_mfma(p, ks, n_cu)) return ks == 3 ? conv_f32_mfma_pick<3>(p, n_cu, s) : conv_f32_mfma_pick<1>(p, n_cu, s);
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
