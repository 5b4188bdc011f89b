// agcm.hip -- Adaptive Global Colour Mapping (AGCM) on gfx950.
//
// Reference: ConditionNet.forward dynamic branch, Condition_arch.py:559-585, and its
// Color_Condition classifier, Condition_arch.py:8-35.
//
//  cls_block   one classifier block: [InstanceNorm of the previous block on load] -> conv1x1 ->
//              AvgPool2d(3,2,1,count_include_pad) -> LeakyReLU(0.2).  conv1x1 and the average
//              commute, so the 3x3 window is summed first (per input channel, with the number
//              of in-image taps kept for the bias) and the 1x1 matrix applied once: 9x fewer MACs.
//  cls_stats   per-channel mean / rstd of a block output (two-pass, fp32) for the next block's
//              InstanceNorm, and the plain mean that the final GAP needs.
//  agcm_fold   6-vector = conv1x1(128->6)(GAP(.)) (GAP and a 1x1 conv commute), the six
//              Linear(6->64/64/3) scale/shift heads, and the fold  conv(x)*s + t + conv(x)
//              == conv'(x) with W' = W(1+s), b' = b(1+s)+t  (SURVEY.md appendix A.3), written
//              out as ready-to-load MFMA A-fragments for this frame.
//  agcm_mlp    the per-pixel 3->64->64->3 MLP: three chained v_mfma_f32_32x32x16_f16 stages per
//              32 pixels; the fp32 accumulator tile of one stage is packed to f16 and used
//              directly as the B operand of the next (no LDS round trip), biases enter through
//              the accumulator init in fp32.
#include "launchers.h"

namespace {

// ------------------------------------------------------------------------------- classifier
// in : Ci planes of Hi x Wi (f16 when IN_F16 else f32);  out: Co planes of Ho x Wo f32
// W8A8 classifier convs (full INT8 recipe): the conv input is fake-quantised element by element in fp32, exactly the
// reference's CPU arithmetic (W8A8Conv2d.forward, hdrtvnet_torch.py:353-356); the classifier is 0.001 % of the frame's
// MACs and runs on the vector ALU in every precision.
__device__ __forceinline__ float fake_q(float x, const FakeQ &q)
{
    const float c = fminf(fmaxf(__builtin_rintf(__builtin_fmaf(x, q.inv, q.zoff)), 0.f), 255.f);
    return c * q.scale + q.zero;      // q.zero = x_zero (asymmetric) or -128 * x_scale (symmetric: the u8 code is x / x_scale + 128)
}

// PX output pixels per workgroup: 16, or 4 for the small maps of the last blocks (four times the workgroups, a quarter of
// the dependent FMA chain per thread: block 5 at 4K is 510 pixels)
template <bool IN_F16, int PX>
__global__ __launch_bounds__(256) void cls_block_kernel(const void *__restrict__ in_, int Ci, int Hi, int Wi,
                                                        const float *__restrict__ nmean, const float *__restrict__ nrstd,
                                                        const float *__restrict__ ngamma, const float *__restrict__ nbeta,
                                                        const float *__restrict__ Wt, const float *__restrict__ bias, int Co,
                                                        float *__restrict__ out, int Ho, int Wo, float2 *__restrict__ part,
                                                        FakeQ qin, FakeQ qstat)
{
    constexpr int NCG = 256 / PX;                 // channel lanes per pixel
    __shared__ float s_sum[128][PX + 1];
    __shared__ float s_cnt[PX];
    const int npix = Ho * Wo;
    const int p0 = blockIdx.x * PX;
    // phase 1: windowed sums of the (normalised) input, PX pixels x Ci channels
    for (int e = threadIdx.x; e < Ci * PX; e += 256) {
        const int px = e % PX, ci = e / PX;
        const int p = p0 + px;
        float s = 0.f;
        int cnt = 0;
        if (p < npix) {
            const int oy = p / Wo, ox = p % Wo;
            for (int ky = -1; ky <= 1; ++ky) {
                const int iy = 2 * oy + ky;
                if (iy < 0 || iy >= Hi) continue;
                for (int kx = -1; kx <= 1; ++kx) {
                    const int ix = 2 * ox + kx;
                    if (ix < 0 || ix >= Wi) continue;
                    const size_t off = ((size_t)ci * Hi + iy) * Wi + ix;
                    float xv = IN_F16 ? (float)reinterpret_cast<const f16 *>(in_)[off] : reinterpret_cast<const float *>(in_)[off];
                    if (qin.on) {             // InstanceNorm of the previous block, then this conv's activation quantiser, per element
                        if (nmean) xv = (xv - nmean[ci]) * nrstd[ci] * ngamma[ci] + nbeta[ci];
                        xv = fake_q(xv, qin);
                    }
                    s += xv;
                    ++cnt;
                }
            }
            if (nmean && !qin.on) {
                const float a = nrstd[ci] * ngamma[ci];
                s = a * s + (nbeta[ci] - nmean[ci] * a) * (float)cnt;
            }
        }
        s_sum[ci][px] = s;
        if (ci == 0) s_cnt[px] = (float)cnt;
    }
    __syncthreads();
    const int px = threadIdx.x % PX, cg = threadIdx.x / PX;
    const int p = p0 + px;
    const bool ok = p < npix;
    for (int co = cg; co < Co; co += NCG) {
        const float *w = Wt + (size_t)co * Ci;
        float acc = 0.f;
        for (int ci = 0; ci < Ci; ++ci) acc += w[ci] * s_sum[ci][px];
        float v = (acc + bias[co] * s_cnt[px]) / 9.f;
        v = v >= 0.f ? v : v * 0.2f;
        if (ok) out[(size_t)co * npix + p] = v;
        // per-block partial sums for the InstanceNorm statistics (fixed order: deterministic); qstat: the reader of the
        // mean is a W8A8 conv (model.20 behind the global average): average its fake-quantised input instead
        const float vs = qstat.on ? fake_q(v, qstat) : v;
        float s1 = ok ? vs : 0.f, s2 = ok ? vs * vs : 0.f;
#pragma unroll
        for (int o = 1; o < PX; o <<= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        if (px == 0) part[(size_t)co * gridDim.x + blockIdx.x] = make_float2(s1, s2);
    }
}

__device__ __forceinline__ float block_sum(float v, float *scratch)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__device__ __forceinline__ double block_sum_d(double v, double *scratch)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// one workgroup per channel: combine the per-block (sum, sum of squares) partials in fp64
__global__ __launch_bounds__(256) void cls_stats_kernel(const float2 *__restrict__ part, int nblk, int n, float eps,
                                                        float *__restrict__ mean, float *__restrict__ rstd)
{
    __shared__ double scratch[4];
    const float2 *p = part + (size_t)blockIdx.x * nblk;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        const float2 v = p[i];
        s1 += (double)v.x;
        s2 += (double)v.y;
    }
    s1 = block_sum_d(s1, scratch);
    s2 = block_sum_d(s2, scratch);
    if (threadIdx.x == 0) {
        const double m = s1 / n;
        double var = s2 / n - m * m;
        if (var < 0.0) var = 0.0;
        mean[blockIdx.x] = (float)m;
        rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// ------------------------------------------------------------------------------------- fold
struct FoldWeights {
    const float *w20, *b20;                           // 6x128, 6
    const float *ws[3], *bs[3], *wt[3], *bt[3];       // scale / shift Linear heads: first, HR, last
    const float *w1, *b1, *w2, *b2, *w3, *b3;         // conv_first 64x3, HRconv 64x64, conv_last 3x64
};

// frag buffer layout (f16x8 per lane): [0..1] layer-1 A tiles, [2..9] layer-2 (mt*4+s), [10..13] layer-3 (s)
constexpr int AGCM_NFRAG = 14;
// bias buffer (f32): [0..63] b1', [64..127] b2', [128..159] b3' (rows 3..31 zero), [160..165] fea6

__global__ __launch_bounds__(256) void agcm_fold_kernel(const float *__restrict__ mean5, FoldWeights fw,
                                                        f16 *__restrict__ frags, float *__restrict__ biasbuf)
{
    __shared__ float fea[6];
    __shared__ float sc[3][64], sh[3][64];
    const int tid = threadIdx.x;
    if (tid < 6) {
        float a = fw.b20[tid];
        for (int k = 0; k < 128; ++k) a += fw.w20[tid * 128 + k] * mean5[k];
        fea[tid] = a;
        biasbuf[160 + tid] = a;
    }
    __syncthreads();
    for (int e = tid; e < 3 * 64; e += 256) {
        const int st = e / 64, m = e % 64;
        const int n = st == 2 ? 3 : 64;
        float s = 0.f, t = 0.f;
        if (m < n) {
            s = fw.bs[st][m];
            t = fw.bt[st][m];
            for (int k = 0; k < 6; ++k) {
                s += fw.ws[st][m * 6 + k] * fea[k];
                t += fw.wt[st][m * 6 + k] * fea[k];
            }
        }
        sc[st][m] = s;
        sh[st][m] = t;
    }
    __syncthreads();
    for (int e = tid; e < 160; e += 256) {
        float v = 0.f;
        if (e < 64) v = fw.b1[e] * (1.f + sc[0][e]) + sh[0][e];
        else if (e < 128) v = fw.b2[e - 64] * (1.f + sc[1][e - 64]) + sh[1][e - 64];
        else if (e - 128 < 3) v = fw.b3[e - 128] * (1.f + sc[2][e - 128]) + sh[2][e - 128];
        biasbuf[e] = v;
    }
    // A fragments: lane l holds row m = tile*32 + (l&31), k slots 8*(l>>5)+j of one 16-wide k-step
    for (int e = tid; e < AGCM_NFRAG * 64 * 8; e += 256) {
        const int j = e & 7, lane = (e >> 3) & 63, f = e >> 9;
        const int r = lane & 31, p = 8 * (lane >> 5) + j;
        float v = 0.f;
        if (f < 2) {                       // layer 1: natural k order, k = colour channel (3 used)
            const int m = f * 32 + r;
            if (p < 3) v = fw.w1[m * 3 + p] * (1.f + sc[0][m]);
        } else if (f < 10) {               // layer 2: k permuted (operand comes from an accumulator)
            const int mt = (f - 2) >> 2, s = (f - 2) & 3;
            const int m = mt * 32 + r, k = 16 * s + acc_kperm16(p);
            v = fw.w2[m * 64 + k] * (1.f + sc[1][m]);
        } else {                           // layer 3: rows 0..2 real
            const int s = f - 10, k = 16 * s + acc_kperm16(p);
            if (r < 3) v = fw.w3[r * 64 + k] * (1.f + sc[2][r]);
        }
        frags[e] = (f16)v;
    }
}

// W8A8 AGCM (full INT8 recipe): the six Linear heads see the 6-vector through their own quantisers (W8A8Linear.forward,
// hdrtvnet_torch.py:398-409, fp32), and the GFM modulation  conv(x) * s + t + conv(x)  of an int8 conv
//     conv(x)[m] = P[m] * acc + Q[m]          (P = x_scale * w_scale, Q = w_scale * (128 x_scale + x_zero) * sum(w) + bias)
// becomes the per-frame dequantisation constants of agcm_mlp_q8:  A = P (1 + s) / x_scale',  B = (Q (1 + s) + t) / x_scale'
// (x_scale' = the next layer's; the last layer keeps real units), in the [mt][lane half][A16 | B16] register order.
__global__ __launch_bounds__(256) void agcm_fold_q8_kernel(const float *__restrict__ mean5, FoldWeights fw, AgcmFoldQ8Args q,
                                                           float *__restrict__ biasbuf)
{
    __shared__ float fea[6];
    __shared__ float sc[3][64], sh[3][64];
    const int tid = threadIdx.x;
    if (tid < 6) {
        float a = fw.b20[tid];
        for (int k = 0; k < 128; ++k) a += fw.w20[tid * 128 + k] * mean5[k];     // mean5: mean of model.20's fake-quantised input
        fea[tid] = a;
        biasbuf[160 + tid] = a;
    }
    __syncthreads();
    for (int e = tid; e < 3 * 64; e += 256) {
        const int st = e / 64, m = e % 64;
        const int n = st == 2 ? 3 : 64;
        float s = 0.f, t = 0.f;
        if (m < n) {
            s = fw.bs[st][m];
            t = fw.bt[st][m];
            for (int k = 0; k < 6; ++k) {
                s += fw.ws[st][m * 6 + k] * (q.qlin[st].on ? fake_q(fea[k], q.qlin[st]) : fea[k]);
                t += fw.wt[st][m * 6 + k] * (q.qlin[3 + st].on ? fake_q(fea[k], q.qlin[3 + st]) : fea[k]);
            }
        }
        sc[st][m] = s;
        sh[st][m] = t;
    }
    __syncthreads();
    // layer l = 0, 1: [mt][lh][A16|B16] at l * 128; layer 2: [lh][A16|B16] at 256 (rows 0..2 real)
    for (int e = tid; e < 320; e += 256) {
        const int l = e < 256 ? e / 128 : 2, r = e < 256 ? e % 128 : e - 256;
        const int mt = l < 2 ? r / 64 : 0, lh = (r % 64) / 32, isB = (r % 32) / 16, j = r % 16;
        const int m = 32 * mt + 8 * (j >> 2) + 4 * lh + (j & 3);
        const float inv = l == 0 ? q.inv2 : (l == 1 ? q.inv3 : 1.f);
        float v = 0.f;
        if (l < 2 || m < 3) {
            const float g = 1.f + sc[l][m];
            v = isB ? (q.Q[l * 64 + m] * g + sh[l][m]) * inv : q.P[l * 64 + m] * g * inv;
        }
        q.consts[e] = v;
    }
}

// -------------------------------------------------------------------------------------- MLP
__device__ __forceinline__ f32x16 bias_tile(const float *b, int lh)
{
    f32x16 a;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4 *>(b + 8 * g + 4 * lh);
        a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
    }
    return a;
}

__device__ __forceinline__ f16x8 relu_pack(const f32x16 &a, int s)
{
    f16x8 o, z;                // f16(max(v, 0)) == max(f16(v), 0): rounding is monotonic; packed max is 3x cheaper
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = (f16)a[8 * s + j]; z[j] = (f16)0.f; }
    return __builtin_elementwise_max(o, z);
}

__global__ __launch_bounds__(256) void agcm_mlp_kernel(const f16 *__restrict__ in, f16 *__restrict__ out, size_t npix,
                                                       const f16 *__restrict__ frags, const float *__restrict__ biasbuf)
{
    __shared__ __attribute__((aligned(16))) float s_bias0[160];
    for (int e = threadIdx.x; e < 160; e += 256) s_bias0[e] = biasbuf[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
    f16x8 a1[2], a2[8], a3[4];
    const f16x8 *fr = reinterpret_cast<const f16x8 *>(frags);
#pragma unroll
    for (int i = 0; i < 2; ++i) a1[i] = fr[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 8; ++i) a2[i] = fr[(2 + i) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) a3[i] = fr[(10 + i) * 64 + lane];

    const size_t ngrp = (npix + 31) / 32;
    const size_t wave_id = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
    // the three input values of a group are fetched one trip ahead: at two waves per SIMD (220 VGPRs) nothing else hides the
    // load latency in front of the dependent MFMA chain
    f16 nx[3] = {(f16)0.f, (f16)0.f, (f16)0.f};
    auto fetch = [&](size_t g) {
        const size_t pix = g * 32 + l31;
        const size_t o = (g < ngrp && pix < npix && lh == 0) ? pix : 0;       // masked lanes read pixel 0 and drop it
        nx[0] = in[o]; nx[1] = in[npix + o]; nx[2] = in[2 * npix + o];
    };
    fetch(wave_id);
    for (size_t g = wave_id; g < ngrp; g += nwave) {
        const size_t pix = g * 32 + l31;
        const bool ok = pix < npix;
        f16x8 x;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (f16)0.f;
        if (ok && lh == 0) { x[0] = nx[0]; x[1] = nx[1]; x[2] = nx[2]; }
        fetch(g + nwave);
        // the five bias tiles are read from LDS in every trip: hoisted out of the loop (80 VGPRs) they cost the kernel its third
        // wave per SIMD
        int bo = 0;
        asm volatile("" : "+v"(bo));
        const float *s_bias = s_bias0 + bo;
        // layer 1: 3 -> 64, ReLU
        f32x16 h0 = bias_tile(s_bias + 0, lh), h1 = bias_tile(s_bias + 32, lh);
        h0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[0], x, h0, 0, 0, 0);
        h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[1], x, h1, 0, 0, 0);
        f16x8 b[4] = {relu_pack(h0, 0), relu_pack(h0, 1), relu_pack(h1, 0), relu_pack(h1, 1)};
        // layer 2: 64 -> 64, ReLU
        f32x16 g0 = bias_tile(s_bias + 64, lh), g1 = bias_tile(s_bias + 96, lh);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[s], b[s], g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2[4 + s], b[s], g1, 0, 0, 0);
        }
        f16x8 c[4] = {relu_pack(g0, 0), relu_pack(g0, 1), relu_pack(g1, 0), relu_pack(g1, 1)};
        // layer 3: 64 -> 3 (rows 0..2 of a 32-row tile)
        f32x16 o = bias_tile(s_bias + 128, lh);
#pragma unroll
        for (int s = 0; s < 4; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(a3[s], c[s], o, 0, 0, 0);
        if (ok && lh == 0) {
            out[pix] = (f16)o[0];
            out[npix + pix] = (f16)o[1];
            out[2 * npix + pix] = (f16)o[2];
        }
    }
}

}  // namespace

// returns the number of workgroups (= partials per channel) through *nblk
hipError_t cls_block_launch(const void *in, int in_f16, int Ci, int Hi, int Wi, const float *nmean, const float *nrstd,
                            const float *ngamma, const float *nbeta, const float *Wt, const float *bias, int Co, float *out,
                            int Ho, int Wo, float *part, hipStream_t s, const FakeQ *qin, const FakeQ *qstat, int *nblk)
{
    const FakeQ off{0, 0.f, 0.f, 0.f, 0.f};
    const FakeQ qi = qin ? *qin : off, qs = qstat ? *qstat : off;
    float2 *pt = reinterpret_cast<float2 *>(part);
    const int npix = Ho * Wo;
    const bool small = npix <= 4096;
    const int grid = small ? (npix + 3) / 4 : (npix + 15) / 16;
    if (nblk) *nblk = grid;
#define CLS_LAUNCH(F16, PXV) hipLaunchKernelGGL((cls_block_kernel<F16, PXV>), dim3(grid), dim3(256), 0, s, in, Ci, Hi, Wi, nmean, nrstd, \
                                                 ngamma, nbeta, Wt, bias, Co, out, Ho, Wo, pt, qi, qs)
    if (small) { if (in_f16) CLS_LAUNCH(true, 4); else CLS_LAUNCH(false, 4); }
    else { if (in_f16) CLS_LAUNCH(true, 16); else CLS_LAUNCH(false, 16); }
#undef CLS_LAUNCH
    return hipGetLastError();
}
hipError_t cls_stats_launch(const float *part, int C, int nblk, int n, float eps, float *mean, float *rstd, hipStream_t s)
{
    hipLaunchKernelGGL(cls_stats_kernel, dim3(C), dim3(256), 0, s, reinterpret_cast<const float2 *>(part), nblk, n, eps, mean,
                       rstd);
    return hipGetLastError();
}

hipError_t agcm_fold_launch(const AgcmFoldArgs &a, f16 *frags, float *biasbuf, hipStream_t s)
{
    FoldWeights fw;
    fw.w20 = a.w20; fw.b20 = a.b20;
    for (int i = 0; i < 3; ++i) { fw.ws[i] = a.ws[i]; fw.bs[i] = a.bs[i]; fw.wt[i] = a.wt[i]; fw.bt[i] = a.bt[i]; }
    fw.w1 = a.w1; fw.b1 = a.b1; fw.w2 = a.w2; fw.b2 = a.b2; fw.w3 = a.w3; fw.b3 = a.b3;
    hipLaunchKernelGGL(agcm_fold_kernel, dim3(1), dim3(256), 0, s, a.mean5, fw, frags, biasbuf);
    return hipGetLastError();
}

hipError_t agcm_fold_q8_launch(const AgcmFoldArgs &a, const AgcmFoldQ8Args &q, float *biasbuf, hipStream_t s)
{
    FoldWeights fw;
    fw.w20 = a.w20; fw.b20 = a.b20;
    for (int i = 0; i < 3; ++i) { fw.ws[i] = a.ws[i]; fw.bs[i] = a.bs[i]; fw.wt[i] = a.wt[i]; fw.bt[i] = a.bt[i]; }
    fw.w1 = a.w1; fw.b1 = a.b1; fw.w2 = a.w2; fw.b2 = a.b2; fw.w3 = a.w3; fw.b3 = a.b3;
    hipLaunchKernelGGL(agcm_fold_q8_kernel, dim3(1), dim3(256), 0, s, a.mean5, fw, q, biasbuf);
    return hipGetLastError();
}

hipError_t agcm_mlp_launch(const f16 *in, f16 *out, size_t npix, const f16 *frags, const float *biasbuf, hipStream_t s)
{
    size_t ngrp = (npix + 31) / 32;
    size_t blocks = (ngrp + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(agcm_mlp_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, out, npix, frags, biasbuf);
    return hipGetLastError();
}
