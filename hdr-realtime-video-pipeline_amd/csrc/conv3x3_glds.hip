// conv3x3_glds.hip -- the MFMA-bound 3x3 convolutions of the HG head (Cin >= 64, Cout >= 128):
// implicit GEMM with LDS-DMA staging (global_load_lds_dwordx4) and a counted-vmcnt pipeline.
//
// Same GEMM view and LDS images as conv_igemm.hip (M = 128 output channels, N = pixels,
// K = 64-channel chunk x tap, activation halo tile re-read for all 9 taps, XOR-swizzled
// 16-byte chunks) with what the first profile asked for:
//   * 16x16-pixel tile, 8 waves (2 per SIMD): the weight tile is shared by 256 pixels and the
//     halo overhead drops from 41 % to 27 %;
//   * staging never touches VGPRs: every wave issues 1-KiB global_load_lds pieces whose
//     per-lane SOURCE address carries the swizzle (the LDS image stays lane-linear; rule 21 of
//     the CDNA4 guide); out-of-image halo pixels read a zero page;
//   * weight tiles live in a 3-slot ring and are issued two iterations ahead, the next
//     chunk's halo tile three iterations ahead; the only wait in the loop is a counted
//     s_waitcnt vmcnt(N) in front of ONE raw s_barrier per iteration (never vmcnt(0), never
//     __syncthreads(): its fence would drain the LDS-DMA queue).
#include "launchers.h"

namespace {

#ifndef GLDS_PIN
#define GLDS_PIN 1
#endif
constexpr int TH = 16, TW = 16;
constexpr int CT = 64, PIXB = CT * 2;                    // 64-channel chunk = 128 B per pixel
// KS = 3: 18x18 halo tile (324 px -> 6 pieces per wave, 48 KiB), double-buffered per channel chunk.
// KS = 1: 16x16 tile (256 px -> 4 pieces per wave, 32 KiB), one buffer per iteration in a 3-slot ring.
template <int KS> struct Geo {
    static constexpr int HW = TW + KS - 1, NPIX = (TH + KS - 1) * (TW + KS - 1);
    static constexpr int A_PIECES_PER_WAVE = KS == 3 ? 6 : 4;
    static constexpr int A_BYTES = 8 * A_PIECES_PER_WAVE * 1024;
    static constexpr int A_SLOTS = KS == 3 ? 2 : 3;
};
constexpr int BN = 128;
constexpr int B_BYTES = BN * PIXB;                       // 16 KiB = 16 pieces = 2 per wave
constexpr int B_PIECES_PER_WAVE = 2;
constexpr int SMEM = 2 * Geo<3>::A_BYTES + 3 * B_BYTES;  // 144 KiB (== 3 * Geo<1>::A_BYTES + 3 * B_BYTES)
static_assert(3 * Geo<1>::A_BYTES + 3 * B_BYTES == SMEM, "both geometries use the same LDS budget");
constexpr int OUT_ROWB = BN * 2 + 16;
static_assert(TH * TW * OUT_ROWB + 3 * 64 * 4 <= SMEM, "epilogue tile must fit");

// 16-byte chunk swizzle.  Weight rows use their row index; halo pixels use their COLUMN in the
// 18-wide halo tile: a 32-lane fragment covers 16 columns of two tile rows, and with the column as
// key every ds_read_b128 lane group hits 16 distinct 16-byte slots of the 256-byte bank row
// (keying on the linear pixel index costs a 2-way conflict on every activation read).
__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }
// The 16x16x32 variant reads 16 rows x 4 k-groups per fragment: there the plain low bits of the
// row / halo column are the conflict-free key (emulated over all 9 taps and 4 lane groups).
template <bool M16> __device__ __forceinline__ int swzk(int row) { return M16 ? (row & 7) : ((row >> 1) & 7); }

__device__ __forceinline__ void glds16(const void *g, void *lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

template <int KS, bool M16>
__global__ __launch_bounds__(512) void conv_glds_kernel(ConvParams p)
{
    using G = Geo<KS>;
    constexpr int HW = G::HW, NPIX = G::NPIX, A_BYTES = G::A_BYTES, A_PIECES_PER_WAVE = G::A_PIECES_PER_WAVE;
    constexpr int NTAP = KS * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;
    char *sB = smem + G::A_SLOTS * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    const int nwg = gridDim.x;
    int t;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int ntn = p.CoutPad / BN;
    const int nt_i = t % ntn, sp = t / ntn;
    const int ty = sp / p.tiles_x, tx = sp % p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = nt_i * BN;
    const int iy0 = oy0 - KS / 2, ix0 = ox0 - KS / 2;

    const int nchunk = (p.c0 + p.c1) / CT, nchunk0 = p.c0 / CT;
    const int nit = nchunk * NTAP;

    // ---- LDS-DMA issue helpers (wave-uniform LDS base, per-lane swizzled source) ------------
    const int l_row = lane >> 3, l_slot = lane & 7;
    auto issue_A = [&](int cc, int buf) {
        const f16 *src;
        int cs, coff;
        if (cc < nchunk0) { src = p.src0; cs = p.s0_stride; coff = cc * CT; }
        else { src = p.src1; cs = p.s1_stride; coff = (cc - nchunk0) * CT; }
#pragma unroll
        for (int it = 0; it < A_PIECES_PER_WAVE; ++it) {
            const int piece = wave + it * 8;
            const int hp = piece * 8 + l_row;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = hp < NPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
            const f16 *g = ok ? src + ((size_t)iy * p.Wi + ix) * cs + coff + ((l_slot ^ swzk<M16>(hx)) << 3)
                              : p.zeros + (l_slot << 3);
            glds16(g, sA + buf * A_BYTES + piece * 1024);
        }
    };
    auto issue_B = [&](int it_i, int slot) {
        const int cc = it_i / NTAP, tap = it_i - cc * NTAP;
        const f16 *base = p.wpk + ((size_t)(tap * nchunk + cc) * p.CoutPad + n0) * CT;
#pragma unroll
        for (int k = 0; k < B_PIECES_PER_WAVE; ++k) {
            const int piece = wave * B_PIECES_PER_WAVE + k;
            const int n = piece * 8 + l_row;
            glds16(base + (size_t)n * CT + ((l_slot ^ swzk<M16>(n)) << 3), sB + slot * B_BYTES + piece * 1024);
        }
    };

    // ---- wave tiling: 2 (channels) x 4 (pixels) waves, each 64 ch x 64 px = 2x2 MFMA tiles ---
    const int wc = wave & 1, wp = wave >> 1;
    int hp_base[2], hx_base[2], wrow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (wp * 2 + j) * 32 + l31;
        hp_base[j] = (q / TW) * HW + (q % TW);
        hx_base[j] = q % TW;
        wrow[j] = (wc * 2 + j) * 32 + l31;
    }
    f32x16 acc[2][2];
    f32x4 acc16[4][4];
    if constexpr (M16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    if constexpr (KS == 1) {
        // 1x1: every iteration is a new 64-channel chunk (of src0, then of src1 = channel concat);
        // activations and weights both ride 3-slot rings, issued two iterations ahead.
        issue_A(0, 0);
        issue_B(0, 0);
        if (nit > 1) {
            issue_A(1, 1);
            issue_B(1, 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        int slot = 0;
#pragma unroll 1
        for (int it_i = 0; it_i < nit; ++it_i) {
            const bool pf = it_i + 2 < nit;
            const int s2 = slot == 0 ? 2 : slot - 1;          // (slot + 2) % 3
            if (pf) { issue_A(it_i + 2, s2); issue_B(it_i + 2, s2); }
            const char *a = sA + slot * A_BYTES;
            const char *b = sB + slot * B_BYTES;
#pragma unroll
            for (int ks = 0; ks < CT / 16; ++ks) {
                const int chunk = ks * 2 + lh;
                f16x8 wf[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    wf[i] = *reinterpret_cast<const f16x8 *>(b + wrow[i] * PIXB + ((chunk ^ swz64(wrow[i])) << 4));
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    xf[j] = *reinterpret_cast<const f16x8 *>(a + hp_base[j] * PIXB + ((chunk ^ swz64(hx_base[j])) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
            if (pf) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A(it+2): 4 pieces, B(it+2): 2 pieces
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            slot = slot == 2 ? 0 : slot + 1;
        }
    } else {
    // ---- prologue: halo(0), weights(0), weights(1) -------------------------------------------
    issue_A(0, 0);
    issue_B(0, 0);
    if (nit > 1) {
        issue_B(1, 1);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    int it_i = 0;
    for (int cc = 0; cc < nchunk; ++cc) {
        const char *a = sA + (cc & 1) * A_BYTES;
        const bool next_chunk = cc + 1 < nchunk;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++it_i) {
            const bool pf = it_i + 2 < nit;
            if (pf) issue_B(it_i + 2, (it_i + 2) % 3);
            if (tap == 6 && next_chunk) issue_A(cc + 1, (cc + 1) & 1);

            const char *b = sB + (it_i % 3) * B_BYTES;
            const int tap_off = (tap / 3) * HW + (tap % 3);
            if constexpr (M16) {
                // 16x16x32 MFMAs: wave tile 64 ch x 64 px = 4x4 tiles, k-step 32, lane = (row & 15, k-group)
                const int l15 = lane & 15, kg = lane >> 4;
                const char *bw = b + ((wc * 64 + l15) * PIXB);
                const char *ax = a + ((wp * 4 * HW + l15 + tap_off) * PIXB);
                const int kw = l15 & 7, kx = (l15 + tap % 3) & 7;
                f16x8 wf[2][4], xf[2][4];
                auto ldw = [&](int ks, int i) {
                    wf[ks][i] = *reinterpret_cast<const f16x8 *>(bw + i * 16 * PIXB + (((ks * 4 + kg) ^ kw) << 4));
                };
                auto ldx = [&](int ks, int j) {
                    xf[ks][j] = *reinterpret_cast<const f16x8 *>(ax + j * HW * PIXB + (((ks * 4 + kg) ^ kx) << 4));
                };
                // program order IS the schedule: sched_barrier(0) lets nothing cross
#pragma unroll
                for (int i = 0; i < 4; ++i) { ldw(0, i); ldx(0, i); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    // k-step 1 fragments in the order its MFMAs want them: w0 x0 x1 x2 x3 w1 w2 w3
                    acc16[g >> 2][g & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][g >> 2], xf[0][g & 3], acc16[g >> 2][g & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == 0) ldw(1, 0); else if (g < 5) ldx(1, g - 1); else ldw(1, g - 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 8; m < 16; ++m)
                    acc16[m >> 2][m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][m >> 2], xf[0][m & 3], acc16[m >> 2][m & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    acc16[m >> 2][m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][m >> 2], xf[1][m & 3], acc16[m >> 2][m & 3], 0, 0, 0);
            } else {
            // fragment reads one k-step ahead of the MFMAs that consume them; the interleave is pinned
            // below (hipcc otherwise sinks each read group down to its MFMAs behind an lgkmcnt(0))
            f16x8 wf[4][2], xf[4][2];
            auto ldfrag = [&](int ks) {
                const int chunk = ks * 2 + lh;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    wf[ks][i] = *reinterpret_cast<const f16x8 *>(b + wrow[i] * PIXB + ((chunk ^ swz64(wrow[i])) << 4));
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int hp = hp_base[j] + tap_off;
                    xf[ks][j] = *reinterpret_cast<const f16x8 *>(a + hp * PIXB + ((chunk ^ swz64(hx_base[j] + tap % 3)) << 4));
                }
            };
            ldfrag(0);
#pragma unroll
            for (int ks = 0; ks < CT / 16; ++ks) {
                if (ks + 1 < CT / 16) ldfrag(ks + 1);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][i], xf[ks][j], acc[i][j], 0, 0, 0);
            }
            if (GLDS_PIN) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);           // reads of k-steps 0 and 1
#pragma unroll
                for (int g = 0; g < 8; ++g) {                                 // MFMAs of k-steps 0,1 with reads of 2,3
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
            }
            // The next iteration reads weights(it+1) (issued one iteration ago) and, at a chunk
            // boundary, halo(cc+1) (issued at tap 6).  Allow exactly the younger DMAs in flight.
            if (!pf) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if ((tap == 6 || tap == 7) && next_chunk) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // halo pieces (6) + weights(it+2) (2)
            } else {
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // weights(it+2) only
            }
            __builtin_amdgcn_s_barrier();
        }
    }

    }

    // ---------------------------------------------------------------- epilogue (LDS staged)
    const float aslope = act_slope(p.act);
    char *so = smem;
    if constexpr (M16) {
        const int l15 = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cl = wc * 64 + i * 16 + 4 * kg;
            const float4 sc = *reinterpret_cast<const float4 *>(p.scale + n0 + cl);
            const float4 sh = *reinterpret_cast<const float4 *>(p.shift + n0 + cl);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = (wp * 4 + j) * 16 + l15;
                f16x4 o;
                o[0] = (f16)act_fast(acc16[i][j][0] * sc.x + sh.x, aslope);
                o[1] = (f16)act_fast(acc16[i][j][1] * sc.y + sh.y, aslope);
                o[2] = (f16)act_fast(acc16[i][j][2] * sc.z + sh.z, aslope);
                o[3] = (f16)act_fast(acc16[i][j][3] * sc.w + sh.w, aslope);
                *reinterpret_cast<f16x4 *>(so + q * OUT_ROWB + cl * 2) = o;
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int cl = (wc * 2 + i) * 32 + 8 * qd + 4 * lh;
            const float4 sc = *reinterpret_cast<const float4 *>(p.scale + n0 + cl);
            const float4 sh = *reinterpret_cast<const float4 *>(p.shift + n0 + cl);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = (wp * 2 + j) * 32 + l31;
                f16x4 o;
                o[0] = (f16)act_fast(acc[i][j][4 * qd + 0] * sc.x + sh.x, aslope);
                o[1] = (f16)act_fast(acc[i][j][4 * qd + 1] * sc.y + sh.y, aslope);
                o[2] = (f16)act_fast(acc[i][j][4 * qd + 2] * sc.z + sh.z, aslope);
                o[3] = (f16)act_fast(acc[i][j][4 * qd + 3] * sc.w + sh.w, aslope);
                *reinterpret_cast<f16x4 *>(so + q * OUT_ROWB + cl * 2) = o;
            }
        }
    }
    }
    __syncthreads();

    constexpr int CPP = BN / 8;
    if (p.mode == ST_NHWC) {
        for (int e = tid; e < TH * TW * CPP; e += 512) {
            const int q = e / CPP, c8 = e % CPP;
            const int oy = oy0 + q / TW, ox = ox0 + q % TW;
            const int ch = n0 + c8 * 8;
            if (oy < p.Ho && ox < p.Wo && ch < p.Cout)
                *reinterpret_cast<f16x8 *>(p.dst + ((size_t)oy * p.Wo + ox) * p.dstC + ch) =
                    *reinterpret_cast<const f16x8 *>(so + q * OUT_ROWB + c8 * 16);
        }
    } else if (p.mode == ST_PS) {
        const int cps = p.dstC;
        for (int e = tid; e < TH * TW * CPP; e += 512) {
            const int q = e / CPP, c8 = e % CPP;
            const int oy = oy0 + q / TW, ox = ox0 + q % TW;
            const int ch = n0 + c8 * 8;
            if (oy < p.Ho && ox < p.Wo && ch < p.Cout) {
                const int sub = ch / cps, c = ch % cps;
                const int Y = 2 * oy + (sub >> 1), X = 2 * ox + (sub & 1);
                if (Y < p.Hd && X < p.Wd)
                    *reinterpret_cast<f16x8 *>(p.dst + ((size_t)Y * p.Wd + X) * cps + c) =
                        *reinterpret_cast<const f16x8 *>(so + q * OUT_ROWB + c8 * 16);
            }
        }
    } else if (p.mode == ST_PS_DOT3) {
        // Pixel shuffle followed by a 1x1 conv to 3 channels (HG Up_conv5 -> conv10, first half of
        // the concat): the 64 shuffled channels of an output pixel are one 128-byte run of the
        // LDS row, so the dot products are taken here and only 3 partial sums per pixel leave the CU.
        float *s_w = reinterpret_cast<float *>(smem + TH * TW * OUT_ROWB);
        for (int e = tid; e < 3 * 64; e += 512) s_w[e] = p.dotw[e];
        __syncthreads();
        for (int e = tid; e < TH * TW * 2; e += 512) {
            const int q = e >> 1, s2 = e & 1;
            const int oy = oy0 + q / TW, ox = ox0 + q % TW;
            if (oy < p.Ho && ox < p.Wo) {
                const char *row = so + q * OUT_ROWB + s2 * 128;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const f16x8 v = *reinterpret_cast<const f16x8 *>(row + c8 * 16);
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float x = (float)v[k];
                        a0 += s_w[c8 * 8 + k] * x;
                        a1 += s_w[64 + c8 * 8 + k] * x;
                        a2 += s_w[128 + c8 * 8 + k] * x;
                    }
                }
                const int sub = n0 / 64 + s2;
                const int Y = 2 * oy + (sub >> 1), X = 2 * ox + (sub & 1);
                if (Y < p.Hd && X < p.Wd)
                    *reinterpret_cast<float4 *>(p.dst_dot + ((size_t)Y * p.Wd + X) * 4) = make_float4(a0, a1, a2, 0.f);
            }
        }
    } else {   // ST_POOL
        for (int e = tid; e < (TH / 2) * (TW / 2) * CPP; e += 512) {
            const int pq = e / CPP, c8 = e % CPP;
            const int py = pq / (TW / 2), px = pq % (TW / 2);
            const int oy = oy0 / 2 + py, ox = ox0 / 2 + px;
            const int ch = n0 + c8 * 8;
            if (oy < p.Hd && ox < p.Wd && ch < p.Cout) {
                const int q00 = (2 * py) * TW + 2 * px;
                f16x8 v = *reinterpret_cast<const f16x8 *>(so + q00 * OUT_ROWB + c8 * 16);
                const f16x8 v1 = *reinterpret_cast<const f16x8 *>(so + (q00 + 1) * OUT_ROWB + c8 * 16);
                const f16x8 v2 = *reinterpret_cast<const f16x8 *>(so + (q00 + TW) * OUT_ROWB + c8 * 16);
                const f16x8 v3 = *reinterpret_cast<const f16x8 *>(so + (q00 + TW + 1) * OUT_ROWB + c8 * 16);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f16 m = v[k] > v1[k] ? v[k] : v1[k];
                    const f16 m2 = v2[k] > v3[k] ? v2[k] : v3[k];
                    v[k] = m > m2 ? m : m2;
                }
                *reinterpret_cast<f16x8 *>(p.dst + ((size_t)oy * p.Wd + ox) * p.dstC + ch) = v;
            }
        }
    }
}

}  // namespace

// KS x KS (3 or 1), stride 1, Cin (src0 [+ src1 concat]) multiple of 64, CoutPad multiple of 128; store modes
// NHWC / PS / POOL / PS_DOT3 without residuals (what the HG head needs).  hipErrorInvalidValue otherwise.
hipError_t conv_glds_launch(ConvParams p, int ks, hipStream_t stream)
{
    if ((ks != 1 && ks != 3) || (p.c0 % CT) || (p.c1 % CT) || (p.CoutPad % BN) || p.res1 || p.res2 || p.dst_full ||
        p.mode == ST_PLANAR3 || !p.zeros || (p.mode == ST_PS_DOT3 && (p.dstC != 64 || !p.dotw || !p.dst_dot)))
        return hipErrorInvalidValue;
    static bool attr_set = false;
    static bool m16 = true;
    if (!attr_set) {
        if (const char *e = getenv("HDRTV_GLDS_M16")) m16 = atoi(e) != 0;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_glds_kernel<3, false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_glds_kernel<3, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_glds_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const int grid = p.tiles_x * p.tiles_y * (p.CoutPad / BN);
    if (ks == 3 && m16) hipLaunchKernelGGL((conv_glds_kernel<3, true>), dim3(grid), dim3(512), SMEM, stream, p);
    else if (ks == 3) hipLaunchKernelGGL((conv_glds_kernel<3, false>), dim3(grid), dim3(512), SMEM, stream, p);
    else hipLaunchKernelGGL((conv_glds_kernel<1, false>), dim3(grid), dim3(512), SMEM, stream, p);
    return hipGetLastError();
}
