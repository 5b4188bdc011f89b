// conv32p.hip -- persistent 3x3 / stride-1 convolution for the 32-channel LE main branch, with
// the SFT layer in front of it fused in (gfx950).
//
// Reference: every `conv(sft(x, cond))` pair of HDRUNet3T1 (SFT_layer1 -> HR_conv1, SFT_layer2 ->
// HR_conv2, ResBlock_with_SFT's sft1 -> conv1 and sft2 -> conv2; HDRUNet3T1_arch.py:168-200,
// arch_util.py:60-95), the three up-convs (3x3 32->128 + PixelShuffle + ReLU + skip) and conv_last.
//
// These layers are HBM-bound (32 channels: 64 B per pixel in, 64 B out, 288 MAC per output
// channel), so the kernel is organised around bytes, not MFMA rate:
//   * one persistent workgroup per CU (8 waves) walks 16x16-pixel tiles; the layer's whole weight
//     set (18 KiB for 32->32, 72 KiB for 32->128) is staged into LDS ONCE per workgroup;
//   * the activation halo tile (and the 16-channel condition halo tile when SFT is fused) of tile
//     t+1 is in flight by LDS-DMA (global_load_lds_dwordx4, swizzle on the source address) while
//     tile t is transformed, convolved and stored: double-buffered, no VGPR staging;
//   * fused SFT: the two 16->16->32 1x1 MLPs run as three MFMAs per 32 halo pixels on the
//     condition tile, and x*(scale+1)+shift rewrites the activation tile IN PLACE in LDS before
//     the conv reads it (out-of-image halo pixels are forced to 0 = the conv's zero padding), so
//     the modulated tensor never exists in HBM (saves 128 B/pixel per SFT);
//   * epilogue through LDS: fp32 scale/shift/activation, residual adds, PixelShuffle (one pass
//     per sub-position for the 128-channel up-convs) or the planar 3-channel head.
#include <cstdlib>

#include "launchers.h"

namespace {

// Halo tile: 18 x 18 pixels stored with an LDS row pitch of 20 pixels.  With 64-byte pixels the
// pitch makes (pixel index mod 4) == (column mod 4), and the chunk swizzle (column >> 2) & 3 then
// spreads every ds_read_b128 lane group of a fragment read over all 16 slots of the bank row.
constexpr int TW = 16, HC = TW + 2, HW = 20;
constexpr int OUT_ROWB = 64 + 16;
// NW waves per workgroup, each owning 2 tile rows of 16 pixels: NW = 8 -> 16x16 tile, one workgroup
// per CU; NW = 4 -> 8x16 tile, 76 KiB of LDS, TWO independent workgroups per CU whose phases
// (DMA wait, SFT, conv, store) interleave instead of running in lockstep.
template <int NW> struct Til {
    static constexpr int TH = 2 * NW, NT = 64 * NW;
    static constexpr int NPIX = (TH + 2) * HW;                       // 360 / 200 slots (324 / 180 real)
    static constexpr int A_PW = NW == 8 ? 3 : 4;                     // 1-KiB pieces (16 px x 64 B) per wave
    static constexpr int C_PW = 2;                                   // 1-KiB pieces (32 px x 32 B) per wave
    static constexpr int A_BYTES = NW * A_PW * 1024, C_BYTES = NW * C_PW * 1024;
    static constexpr int OUT_BYTES = TH * TW * OUT_ROWB;
    static_assert(NW * A_PW * 16 >= NPIX && NW * C_PW * 32 >= NPIX, "halo buffers must hold the tile");
};

__device__ __forceinline__ int swz32(int row) { return (row >> 2) & 3; }   // weight rows: by row; halo: by column

__device__ __forceinline__ void glds16(const void *g, void *lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

__device__ __forceinline__ f32x16 tile16(const float *b, int lh)
{
    f32x16 a;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4 *>(b + 8 * g + 4 * lh);
        a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
    }
    return a;
}

template <int NPASS, bool SFT, int NW>
struct Lay {
    static constexpr int A_BYTES = Til<NW>::A_BYTES, C_BYTES = Til<NW>::C_BYTES, OUT_BYTES = Til<NW>::OUT_BYTES;
    static constexpr int COUTP = 32 * NPASS;
    static constexpr int W_BYTES = 9 * COUTP * 64;
    static constexpr int SS_BYTES = COUTP * 8;
    static constexpr int OFF_SS = W_BYTES;
    static constexpr int OFF_A = OFF_SS + SS_BYTES;
    static constexpr int OFF_C = OFF_A + 2 * A_BYTES;
    static constexpr int OFF_OUT = OFF_C + (SFT ? 2 * C_BYTES : 0);
    static constexpr int SMEM = OFF_OUT + OUT_BYTES;
};

template <int NPASS, bool SFT, int NW>
__global__ __launch_bounds__(64 * NW) void conv32p_kernel(Conv32Params p)
{
    using L = Lay<NPASS, SFT, NW>;
    using T = Til<NW>;
    constexpr int TH = T::TH, NT = T::NT, NPIX = T::NPIX, A_BYTES = T::A_BYTES, C_BYTES = T::C_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sW = smem;
    float *sSS = reinterpret_cast<float *>(smem + L::OFF_SS);
    char *sA = smem + L::OFF_A;
    char *sC = smem + L::OFF_C;
    char *sO = smem + L::OFF_OUT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int ntiles = p.tiles_x * p.tiles_y;

    auto issue_tile = [&](int t, int buf) {
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
#pragma unroll
        for (int it = 0; it < T::A_PW; ++it) {
            const int piece = wave + it * NW;
            const int hp = piece * 16 + (lane >> 2), slot = lane & 3;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = hp < NPIX && hx < HC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const f16 *g = ok ? p.src + ((size_t)iy * p.W + ix) * 32 + ((slot ^ swz32(hx)) << 3) : p.zeros + (slot << 3);
            glds16(g, sA + buf * A_BYTES + piece * 1024);
        }
        if (SFT) {
#pragma unroll
            for (int it = 0; it < T::C_PW; ++it) {
                const int piece = wave + it * NW;
                const int hp = piece * 32 + (lane >> 1), half = lane & 1;
                const int hy = hp / HW, hx = hp - hy * HW;
                const int iy = iy0 + hy, ix = ix0 + hx;
                const bool ok = hp < NPIX && hx < HC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const f16 *g = ok ? p.cond + ((size_t)iy * p.W + ix) * 16 + (half << 3) : p.zeros + (half << 3);
                glds16(g, sC + buf * C_BYTES + piece * 1024);
            }
        }
    };

    // ---- once per workgroup: the whole weight set and the per-channel scale/shift into LDS
    for (int piece = wave; piece < 9 * L::COUTP / 16; piece += NW) {
        const int r = piece * 16 + (lane >> 2), slot = lane & 3;     // r = tap*COUTP + n
        const int n = r % L::COUTP;
        glds16(p.wpk + (size_t)r * 32 + ((slot ^ swz32(n)) << 3), sW + piece * 1024);
    }
    for (int e = tid; e < L::COUTP; e += NT) {
        sSS[e] = p.scale[e];
        sSS[L::COUTP + e] = p.shift[e];
    }
    f16x8 sa0, sa1s, sa1t;
    f32x16 sbh, sbs, sbt;
    if (SFT) {
        const f16x8 *fr = reinterpret_cast<const f16x8 *>(p.sft_wfrag);
        sa0 = fr[lane]; sa1s = fr[64 + lane]; sa1t = fr[128 + lane];
        sbh = tile16(p.sft_bias, lh); sbs = tile16(p.sft_bias + 32, lh); sbt = tile16(p.sft_bias + 64, lh);
    }
    // y = x*(scale+1)+shift in place on a landed halo tile (arch_util.py:68-72)
    auto sft_tile = [&](int tt, int buf) {
        const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        char *a = sA + buf * A_BYTES;
        const char *cbuf = sC + buf * C_BYTES;
        for (int g = wave; g < (NPIX + 31) / 32; g += NW) {
            const int hp = g * 32 + l31;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            const bool inimg = hp < NPIX && hx < HC && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const f16x8 cf = *reinterpret_cast<const f16x8 *>(cbuf + hp * 32 + lh * 16);
            const f32x16 h = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa0, cf, sbh, 0, 0, 0);
            f16x8 hs, ht;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float u = h[j], v = h[8 + j];
                hs[j] = (f16)fmaxf(u, 0.1f * u);
                ht[j] = (f16)fmaxf(v, 0.1f * v);
            }
            const f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa1s, hs, sbs, 0, 0, 0);
            const f32x16 sh = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa1t, ht, sbt, 0, 0, 0);
            if (hp < NPIX) {
                // f16 arithmetic like the reference's fp16 model (x*(scale+1)+shift), 4 channels per op
                const f16 keep = inimg ? (f16)1.f : (f16)0.f;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    char *addr = a + hp * 64 + ((qd ^ swz32(hx)) << 4) + 8 * lh;
                    const f16x4 xv = *reinterpret_cast<const f16x4 *>(addr);
                    f16x4 s1, s0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s1[k] = (f16)(sc[4 * qd + k] + 1.f); s0[k] = (f16)sh[4 * qd + k]; }
                    *reinterpret_cast<f16x4 *>(addr) = (xv * s1 + s0) * keep;
                }
            }
        }
    };

    // ---- prologue: tile 0 landed (and SFT-transformed), tile 1 in flight
    int t = blockIdx.x;
    const int step = gridDim.x;
    if (t < ntiles) issue_tile(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (SFT && t < ntiles) sft_tile(t, 0);
    if (t + step < ntiles) issue_tile(t + step, 1);
    __syncthreads();

    const int q = wave * 32 + l31;                       // this lane's output pixel in the tile
    const int qx = q % TW;
    const int hp_base = (q / TW) * HW + qx;

    // Steady state, two barriers per tile:
    //   conv(t) -> stage result in LDS -> [DMA(t+1) landed] -> barrier -> stores(t) fly while the SFT of
    //   tile t+1 runs and DMA(t+2) is issued into the buffer conv(t) just released -> barrier.
    for (int buf = 0; t < ntiles; t += step, buf ^= 1) {
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const char *a = sA + buf * A_BYTES;

        // output offsets of this thread's two 16-byte chunks per pass (-1: outside) and residual prefetch
        long ooff[NPASS][2];
        f16x8 rs1[NPASS][2], rs2[NPASS][2];
        if (p.mode != ST_PLANAR3) {
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int qq = (tid + it * NT) >> 2, c8 = tid & 3;
                    const int oy = oy0 + qq / TW, ox = ox0 + qq % TW;
                    long off = -1;
                    if (oy < p.H && ox < p.W && c8 * 8 < p.Cout) {
                        if (p.mode == ST_PS) {
                            const int Y = 2 * oy + (pass >> 1), X = 2 * ox + (pass & 1);
                            if (Y < p.Hd && X < p.Wd) off = ((long)Y * p.Wd + X) * 32 + c8 * 8;
                        } else {
                            off = ((long)oy * p.W + ox) * p.dstC + c8 * 8;
                        }
                    }
                    ooff[pass][it] = off;
                    f16x8 z;
#pragma unroll
                    for (int k = 0; k < 8; ++k) z[k] = (f16)0.f;
                    rs1[pass][it] = (p.res1 && off >= 0) ? *reinterpret_cast<const f16x8 *>(p.res1 + off) : z;
                    rs2[pass][it] = (p.res2 && off >= 0) ? *reinterpret_cast<const f16x8 *>(p.res2 + off) : z;
                }
        }

#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            f32x16 acc;
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            const int n = pass * 32 + l31;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hp = hp_base + (tap / 3) * HW + (tap % 3);
                const int hx = qx + tap % 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int chunk = ks * 2 + lh;
                    const f16x8 wf = *reinterpret_cast<const f16x8 *>(sW + (tap * L::COUTP + n) * 64 + ((chunk ^ swz32(n)) << 4));
                    const f16x8 xf = *reinterpret_cast<const f16x8 *>(a + hp * 64 + ((chunk ^ swz32(hx)) << 4));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, acc, 0, 0, 0);
                }
            }
            // ---- this pass's 32 channels x 256 pixels into the LDS staging tile
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int cl = 8 * qd + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(sSS + pass * 32 + cl);
                const float4 sh = *reinterpret_cast<const float4 *>(sSS + L::COUTP + pass * 32 + cl);
                f16x4 o;
                o[0] = (f16)act_apply(acc[4 * qd + 0] * sc.x + sh.x, p.act);
                o[1] = (f16)act_apply(acc[4 * qd + 1] * sc.y + sh.y, p.act);
                o[2] = (f16)act_apply(acc[4 * qd + 2] * sc.z + sh.z, p.act);
                o[3] = (f16)act_apply(acc[4 * qd + 3] * sc.w + sh.w, p.act);
                *reinterpret_cast<f16x4 *>(sO + q * OUT_ROWB + cl * 2) = o;
            }
            // the last pass also makes sure the next tile's LDS-DMA has landed before the barrier
            if (pass == NPASS - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (p.mode == ST_PLANAR3) {
                for (int e = tid; e < TH * TW * 3; e += NT) {
                    const int ch = e / (TH * TW), qq = e % (TH * TW);
                    const int oy = oy0 + qq / TW, ox = ox0 + qq % TW;
                    if (oy < p.H && ox < p.W) {
                        float v = (float)*reinterpret_cast<const f16 *>(sO + qq * OUT_ROWB + ch * 2);
                        const size_t off = (size_t)ch * p.H * p.W + (size_t)oy * p.W + ox;
                        if (p.res_planar) v += (float)p.res_planar[off];
                        p.dst_planar[off] = (f16)v;
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int qq = (tid + it * NT) >> 2, c8 = tid & 3;
                    f16x8 v = *reinterpret_cast<const f16x8 *>(sO + qq * OUT_ROWB + c8 * 16);
                    const f16x8 r1 = rs1[pass][it], r2 = rs2[pass][it];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (f16)(((float)v[k] + (float)r1[k]) + (float)r2[k]);
                    f16 *dp = ooff[pass][it] >= 0 ? p.dst + ooff[pass][it] : p.dump + (size_t)tid * 8;
                    *reinterpret_cast<f16x8 *>(dp) = v;
                }
            }
            if (pass < NPASS - 1) __syncthreads();      // staging tile is reused by the next pass
        }
        // stores of tile t are in flight; prepare tile t+1 and prefetch tile t+2
        const int t1 = t + step, t2 = t + 2 * step;
        if (SFT && t1 < ntiles) sft_tile(t1, buf ^ 1);
        if (t2 < ntiles) issue_tile(t2, buf);
        __syncthreads();
    }
}

template <int NPASS, bool SFT, int NW>
hipError_t launch_t(const Conv32Params &p, hipStream_t s)
{
    using L = Lay<NPASS, SFT, NW>;
    static bool attr_set = false;
    auto kern = conv32p_kernel<NPASS, SFT, NW>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::SMEM);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int ntiles = p.tiles_x * p.tiles_y;
    const int cap = 256 * (160 * 1024 / L::SMEM);          // persistent: as many workgroups as fit the chip
    const int grid = ntiles < cap ? ntiles : cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), L::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

hipError_t conv32p_launch(Conv32Params p, hipStream_t s)
{
    static int nw = 0;
    if (!nw) {
        const char *e = getenv("HDRTV_CONV32_NW");      // developer A/B switch: 8 = 16x16 tile, 4 = 8x16 tile x 2 workgroups/CU
        nw = (e && atoi(e) == 8) ? 8 : 4;
    }
    const bool sft = p.cond != nullptr;
    p.tiles_x = (p.W + TW - 1) / TW;
    if (nw == 8 || p.CoutPad == 128) {                   // the 72 KiB weight set of the up-convs leaves room for one workgroup only
        p.tiles_y = (p.H + 15) / 16;
        if (p.CoutPad == 32) return sft ? launch_t<1, true, 8>(p, s) : launch_t<1, false, 8>(p, s);
        if (p.CoutPad == 128 && !sft) return launch_t<4, false, 8>(p, s);
        return hipErrorInvalidValue;
    }
    p.tiles_y = (p.H + 7) / 8;
    if (p.CoutPad == 32) return sft ? launch_t<1, true, 4>(p, s) : launch_t<1, false, 4>(p, s);
    return hipErrorInvalidValue;
}
