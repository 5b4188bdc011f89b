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
    const int tiles_x = (p.Wo + Q8_TW - 1) / Q8_TW, ntiles = tiles_x * ((p.Ho + Q8_TH - 1) / Q8_TH);

    // ---- weights, scale and the border-class shifts: once per (persistent) workgroup -- 9 .. 36 KiB that a per-tile workgroup
    // staged again for every 18 .. 36 KiB halo patch
    const int wrows = KS * KS * p.CoutPad;
    for (int e = tid; e < wrows * NCH; e += 256) {
        const int r = e / NCH, ch = e - r * NCH;
        *reinterpret_cast<i32x4 *>(sW + r * CIN + ((ch ^ rsw<NCH>(r)) << 4)) =
            *reinterpret_cast<const i32x4 *>(p.wpk8 + (size_t)r * CIN + ch * 16);
    }
    for (int e = tid; e < 17 * p.CoutPad; e += 256) sS[e] = e < p.CoutPad ? p.scale[e] : p.shift[e - p.CoutPad];

    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int ty = t / tiles_x, tx = t - ty * tiles_x;
    const int oy0 = ty * Q8_TH, ox0 = tx * Q8_TW;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    if (t != (int)blockIdx.x) __syncthreads();                 // the previous tile's fragment reads are done
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
    }   // tile loop
}

// ---------------------------------------------------------------------------------------------------------------------
// Up to three 3x3 / stride-2 / 64 -> 64 W8A8 layers that read ONE f16 tensor through their own quantisers (CondNet2.0,
// CondNet3.0, CondNet4.0 all read the full-resolution condition map, HDRUNet3T1_arch.py:47-55), organised like the fp16
// kernel of the same layers (conv3x3s2_preg.hip): persistent workgroups walk 8 x 16 output tiles, four waves per layer,
// wave w keeps the K = 576 filter rows of 16 output channels of its layer in 36 VGPRs (nine 16x16x64 int8 A fragments:
// one tap = 64 channels = one K step), no weight traffic and no per-tap barrier.  The 17 x 33 x 64-channel f16 halo of
// the NEXT tile is fetched into registers (six 16-byte loads per thread in flight) while the current tile is convolved
// from LDS; between tiles every thread quantises its share once per layer -- each layer's (x_scale, x_zero) -- into that
// layer's 36 KiB code tile.  The condition map is read from HBM once.  Halo columns are staged de-interleaved (even
// columns first), so the 16 pixels of a stride-2 tap are 16 consecutive 64-byte LDS rows, conflict-free under the
// (row >> 2) & 3 chunk swizzle.  Epilogue straight from the accumulators: dequantise with the border-class shift,
// LeakyReLU, then f16 or the reading layer's int8 codes, 8 / 4 bytes per lane.
template <int NG>
__global__ __launch_bounds__(256 * NG, 1) void conv_q8_multi_kernel(ConvQ8MultiParams p)
{
    constexpr int NT = 256 * NG, TH = Q8_TH, TW = Q8_TW;
    constexpr int HH = 2 * TH + 1, HWD = 2 * TW + 1, NPX = HH * HWD, NEVEN = TW + 1;     // 17 x 33 halo, even columns first
    constexpr int XB = (NPX * 64 + 255) & ~255;
    constexpr int NIT = (NPX * 8 + NT - 1) / NT;               // 16-byte f16 items (8 channels) per thread and tile
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, sub = wave & 3;
    const int l15 = lane & 15, kg = lane >> 4;
    const int tiles_x = (p.Wo + TW - 1) / TW, ntiles = tiles_x * ((p.Ho + TH - 1) / TH);
    const ConvQ8Group &G = p.g[grp];

    // ---- this wave's filter rows: one A fragment per tap (16 output channels x 64 input channels)
    i32x4 wfr[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
        wfr[tap] = *reinterpret_cast<const i32x4 *>(G.wpk8 + ((size_t)tap * 64 + sub * 16 + l15) * 64 + kg * 16);
    const int n0 = sub * 16 + 4 * kg;
    const float4 sc = *reinterpret_cast<const float4 *>(G.scale + n0);
    const float aslope = act_slope(G.act);

    // ---- halo items of this thread: LDS row q (de-interleaved column order), 8-channel chunk c8
    f16x8 pre[NIT];
    auto fetch = [&](int t) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int iy0 = 2 * ty * TH - 1, ix0 = 2 * tx * TW - 1;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + k * NT;
            const int q = e >> 3, c8 = e & 7;
            const int hy = q / HWD, col = q - hy * HWD;
            const int hx = col < NEVEN ? 2 * col : 2 * (col - NEVEN) + 1;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = q < NPX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            // out-of-image items are marked with a NaN in channel 0 (never produced by the network: inputs are finite)
            f16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (f16)0.f;
            if (ok) v = *reinterpret_cast<const f16x8 *>(p.src + ((size_t)iy * p.Wi + ix) * p.src_stride + c8 * 8);
            pre[k] = v;
        }
    };
    auto okmask = [&](int t, int k) -> bool {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int e = tid + k * NT, q = e >> 3;
        const int hy = q / HWD, col = q - hy * HWD;
        const int hx = col < NEVEN ? 2 * col : 2 * (col - NEVEN) + 1;
        const int iy = 2 * ty * TH - 1 + hy, ix = 2 * tx * TW - 1 + hx;
        return q < NPX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
    };
    auto quantise = [&](int t) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int e = tid + k * NT;
            const int q = e >> 3, c8 = e & 7;
            if (q >= NPX) continue;
            const bool ok = okmask(t, k);
            const int off = q * 64 + (((c8 >> 1) ^ ((q >> 2) & 3)) << 4) + (c8 & 1) * 8;
            const f16x8 v = pre[k];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                uint2 w = make_uint2(0u, 0u);                  // out-of-image pixels: code 0
                if (ok) {
                    const float inv = p.g[g].q_inv, zo = p.g[g].q_zoff;
                    w.x = quant4((float)v[0], (float)v[1], (float)v[2], (float)v[3], inv, zo);
                    w.y = quant4((float)v[4], (float)v[5], (float)v[6], (float)v[7], inv, zo);
                }
                *reinterpret_cast<uint2 *>(smem + g * XB + off) = w;
            }
        }
    };

    // read offsets: LDS row = c + l15 for a compile-time c; the swizzle needs bits 2..3 of (c + l15) = of ((c & 3) + l15) + (c & 12):
    // four lane constants (carry of the low two bits) instead of one per c
    int xl[4];
#pragma unroll
    for (int c3 = 0; c3 < 4; ++c3) xl[c3] = (c3 + l15) >> 2;
    const char *X = smem + grp * XB + l15 * 64;

    int t = blockIdx.x;
    const int step = gridDim.x;
    if (t < ntiles) { fetch(t); quantise(t); }
    __syncthreads();
    for (; t < ntiles; t += step) {
        if (t + step < ntiles) fetch(t + step);            // in flight while this tile is convolved
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int ox = tx * TW + l15;
        const int tx0 = 2 * ox - 1;
        const int bx = (tx0 < 0) | ((tx0 + 2 >= p.Wi) << 1);
#pragma unroll 1
        for (int hf = 0; hf < 2; ++hf) {                   // two halves of four output rows: 16 accumulator registers, not 32
            i32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = i32x4{0, 0, 0, 0};
            const char *Xh = X + hf * (8 * HWD * 64);      // output row 4 hf + i reads halo rows 2 (4 hf + i) + ky: 8 rows further down
            const int swh = hf * ((8 * HWD) >> 2);         // ... and (8 * HWD) = 264 rows = 66 swizzle periods of 4: bits 2..3 shift by 66 & 3 = 2
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = (2 * i + ky) * HWD + (kx & 1) * NEVEN + (kx >> 1);
                    const int sw = (xl[c & 3] + ((c >> 2) & 3) + swh) & 3;
                    const i32x4 x = *reinterpret_cast<const i32x4 *>(Xh + c * 64 + ((kg ^ sw) << 4));
                    acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wfr[tap], x, acc[i], 0, 0, 0);
                }
            }
            // ---- epilogue from the accumulators: lane = (pixel column l15, output channels n0 .. n0 + 3), one row per i
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int oy = ty * TH + 4 * hf + i;
                if (oy < p.Ho && ox < p.Wo) {
                    const int ty0 = 2 * oy - 1;
                    const int bcls = ((((ty0 < 0) | ((ty0 + 2 >= p.Hi) << 1)) << 2) | bx) & 15;
                    const float4 sh = *reinterpret_cast<const float4 *>(G.shift + bcls * 64 + n0);
                    const float v0 = act_fast((float)acc[i][0] * sc.x + sh.x, aslope), v1 = act_fast((float)acc[i][1] * sc.y + sh.y, aslope),
                                v2 = act_fast((float)acc[i][2] * sc.z + sh.z, aslope), v3 = act_fast((float)acc[i][3] * sc.w + sh.w, aslope);
                    const size_t o = ((size_t)oy * p.Wo + ox) * 64 + n0;
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
        __syncthreads();                                   // every wave is done reading the code tiles
        if (t + step < ntiles) quantise(t + step);
        __syncthreads();
    }
}

template <int CIN, int KS, int S>
hipError_t launch_q8(const ConvQ8Params &p, int n_cu, hipStream_t s)
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
    // persistent: as many workgroups as are resident at once with this layer's LDS footprint
    static int per_cu_of[161];                                 // by KiB of dynamic LDS; 0 = not asked yet
    int &per_cu = per_cu_of[(smem + 1023) / 1024];
    if (per_cu == 0) {
        int nb = 0;
        per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, 256, smem) == hipSuccess && nb >= 1) ? nb : 1;
    }
    const int ntiles = ((p.Wo + Q8_TW - 1) / Q8_TW) * ((p.Ho + Q8_TH - 1) / Q8_TH);
    const int grid = ntiles < per_cu * n_cu ? ntiles : per_cu * n_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, s, p);
    return hipGetLastError();
}

}  // namespace

template <int NG>
hipError_t launch_multi(const ConvQ8MultiParams &p, int n_cu, hipStream_t s)
{
    constexpr int NPX = 17 * 33, XB = (NPX * 64 + 255) & ~255;
    constexpr int smem = NG * XB;
    static_assert(smem <= 160 * 1024, "LDS budget");
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv_q8_multi_kernel<NG>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const int ntiles = ((p.Wo + Q8_TW - 1) / Q8_TW) * ((p.Ho + Q8_TH - 1) / Q8_TH);
    hipLaunchKernelGGL(kern, dim3(ntiles < n_cu ? ntiles : n_cu), dim3(256 * NG), smem, s, p);
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
hipError_t conv_q8_launch(ConvQ8Params p, hipStream_t s, int n_cu)
{
    if ((p.CoutPad % 32) || (p.Cout % 4) || p.Cout > p.CoutPad || (p.src_stride % 16) || (p.dstC % 4) || !p.wpk8 || !p.scale || !p.shift)
        return hipErrorInvalidValue;
    const int pad = p.ks / 2;
    if (p.Ho != (p.Hi + 2 * pad - p.ks) / p.stride + 1 || p.Wo != (p.Wi + 2 * pad - p.ks) / p.stride + 1) return hipErrorInvalidValue;
    if (p.Cin == 32 && p.ks == 3 && p.stride == 2) return launch_q8<32, 3, 2>(p, n_cu, s);
    if (p.Cin == 64 && p.ks == 3 && p.stride == 2) return launch_q8<64, 3, 2>(p, n_cu, s);
    if (p.Cin == 32 && p.ks == 3 && p.stride == 1) return launch_q8<32, 3, 1>(p, n_cu, s);
    if (p.Cin == 64 && p.ks == 3 && p.stride == 1) return launch_q8<64, 3, 1>(p, n_cu, s);
    if (p.Cin == 64 && p.ks == 1 && p.stride == 1) return launch_q8<64, 1, 1>(p, n_cu, s);
    if (p.Cin == 32 && p.ks == 1 && p.stride == 1) return launch_q8<32, 1, 1>(p, n_cu, s);
    return hipErrorInvalidValue;
}
