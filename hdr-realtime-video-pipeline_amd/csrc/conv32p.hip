// conv32p.hip -- persistent 3x3 / stride-1 convolution for the 32-channel LE main branch, with
// the SFT layer in front of it fused in (gfx950).
//
// Reference: every `conv(sft(x, cond))` pair of HDRUNet3T1 (SFT_layer1 -> HR_conv1, SFT_layer2 ->
// HR_conv2, ResBlock_with_SFT's sft1 -> conv1 and sft2 -> conv2; HDRUNet3T1_arch.py:168-200,
// arch_util.py:60-95), the three up-convs (3x3 32->128 + PixelShuffle + ReLU + skip) and conv_last.
//
// These layers are HBM-bound on paper (32 channels: 64 B per pixel in, 64 B out, 288 MAC per output
// channel) and were instruction-issue-bound in practice (PMC + ISA count: ~1100 VALU and ~800 SALU
// instructions per tile and wave against 24 MFMAs), so the kernel is organised around bytes AND
// around not recomputing lane constants:
//   * persistent workgroups walk pixel tiles; the layer's whole weight set (18 KiB for 32->32,
//     72 KiB for 32->128) is staged into LDS ONCE per workgroup;
//   * the activation halo tile (and the 16-channel condition halo tile when SFT is fused) of tile
//     t+2 is put in flight by LDS-DMA (global_load_lds_dwordx4, swizzle on the source address) while
//     tile t is convolved and tile t+1 is SFT-transformed: double-buffered, no VGPR staging; every
//     per-lane halo coordinate, source offset and LDS address is computed once per workgroup, a tile
//     costs one scalar base plus two unsigned compares per DMA piece;
//   * fused SFT: the two 16->16->32 1x1 MLPs run as three MFMAs per 32 halo pixels on the
//     condition tile, LeakyReLU / modulation in packed f16 (the reference's fp16 model computes them
//     in f16 too), and x*(scale+1)+shift rewrites the activation tile IN PLACE in LDS before the
//     conv reads it (out-of-image halo pixels are forced to 0 = the conv's zero padding), so the
//     modulated tensor never exists in HBM (saves 128 B/pixel per SFT);
//   * LDS halo rows have a pitch of 20 pixels of 64 B: (pixel mod 4) == (column mod 4) and the
//     chunk swizzle (column >> 2) & 3 make every ds_read_b128 fragment read conflict-free;
//     fragment addresses are base ^ (k-step << 5) plus immediates for kernel row and tap;
//   * epilogue through LDS: fp32 scale/shift/activation, residual adds, PixelShuffle (one pass
//     per sub-position for the 128-channel up-convs) or the planar 3-channel head.
// W8A8 layers (template I8; W8A8Conv2d.forward, hdrtvnet_torch.py:351-364): the tile the conv reads is not the f16
// halo tile but its int8 codes c = clamp(rint((x - x_zero) / x_scale), 0, 255) - 128, written by the same per-tile
// pass that applies the SFT modulation (or, for a layer without SFT, by a pass that only quantises) into a second,
// 32-byte-per-pixel LDS tile; the conv is 9 v_mfma_i32_32x32x32_i8 (one tap = 32 input channels = one K step) and
// its integer sum is exact.  The reference pads with zeros AFTER dequantisation and x_zero is a float, so no code
// means "0.0": out-of-image halo pixels hold code 0 (they add nothing to the sum) and the epilogue adds
//     x_scale * w_scale[n] * acc + w_scale[n] * (128 * x_scale + x_zero) * sum(w_int8[n] over the IN-IMAGE taps) + b[n],
// the second term being a per-channel constant for each of the 16 border classes (top / bottom row missing x left /
// right column missing) that the host tabulates (hdrtv_api.hip pack_conv32_i8).
// Out-of-image DMA lanes read the zeroed guard that the workspace keeps behind every tensor
// (hdrtv_api.hip ws_add), so one scalar base + a 32-bit lane offset addresses every piece.
#include <cstdlib>
#include <type_traits>

#include "launchers.h"

namespace {

// Diagnostic build only (make STAMP=1): per-phase s_memtime sums, written by lane 0 of every wave to
// p.dump[(block*NW + wave)*8 + phase] as cycles.  Never compiled into the shipped library.
#ifdef HDRTV_STAMP
#define STAMP_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) do { const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(); st_acc[i] += st_t1 - st_t0; st_t0 = st_t1; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#endif


constexpr int TW = 16, HC = TW + 2, HW = 20;
constexpr int OUT_ROWB = 64 + 16;
// NW waves per workgroup, each owning 2 tile rows of 16 pixels: NW = 8 -> 16x16 tile, one workgroup
// per CU; NW = 4 -> 8x16 tile, 76 KiB of LDS, two independent workgroups per CU.
template <int NW> struct Til {
    static constexpr int TH = 2 * NW, NT = 64 * NW;
    static constexpr int NPIX = (TH + 2) * HW;                       // 360 / 200 slots (324 / 180 real)
    static constexpr int NG = (NPIX + 31) / 32;                      // 32-pixel SFT groups per tile
    static constexpr int G_PW = (NG + NW - 1) / NW;
    static constexpr int A_PW = NW == 8 ? 3 : 4;                     // 1-KiB pieces (16 px x 64 B) per wave
    static constexpr int C_PW = 2;                                   // 1-KiB pieces (32 px x 32 B) per wave
    static constexpr int A_BYTES = NW * A_PW * 1024, C_BYTES = NW * C_PW * 1024;
    static constexpr int OUT_BYTES = TH * TW * OUT_ROWB;
    static_assert(NW * A_PW * 16 >= NPIX && NW * C_PW * 32 >= NPIX, "halo buffers must hold the tile");
};

__device__ __forceinline__ int swz32(int v) { return (v >> 2) & 3; }   // weight rows: by row; halo: by column


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

// LeakyReLU(0.1) on accumulator registers 8s..8s+7, in packed f16 -> next MFMA's B fragment
__device__ __forceinline__ f16x8 lrelu_pack16(const f32x16 &a, int s)
{
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)a[8 * s + j];
    return __builtin_elementwise_max(o, o * (f16)0.1f);
}

template <int NPASS, bool SFT, int NW, bool I8, bool SQ = false>
struct Lay {
    static constexpr int A_BYTES = Til<NW>::A_BYTES, C_BYTES = Til<NW>::C_BYTES, OUT_BYTES = Til<NW>::OUT_BYTES;
    static constexpr int COUTP = 32 * NPASS;
    static constexpr int W_BYTES = 9 * COUTP * (I8 ? 32 : 64);
    static constexpr int SS_BYTES = I8 ? COUTP * 4 * 17 : COUTP * 8;    // I8: scale[COUTP] + shift[16 border classes][COUTP]
    static constexpr int Q_BYTES = I8 ? Til<NW>::NG * 32 * 32 : 0;      // int8 code tile: 32 B per halo slot, whole 32-pixel groups
    static constexpr int OFF_SS = W_BYTES;
    static constexpr int OFF_A = OFF_SS + SS_BYTES;
    static constexpr int OFF_C = OFF_A + 2 * A_BYTES;
    static constexpr int OFF_Q = OFF_C + (SFT ? 2 * C_BYTES : 0);
    static constexpr int OFF_OUT = OFF_Q + 2 * Q_BYTES;
    static constexpr int OFF_K = OFF_OUT + OUT_BYTES;                   // SQ: 6 x 32 dequantisation constants of the SFT convs
    static constexpr int SMEM = OFF_K + (SQ ? 768 : 0);
    static_assert(SMEM <= 160 * 1024, "LDS budget");
};

template <int NPASS, bool SFT, int NW, bool I8, bool SQ>
__global__ __launch_bounds__(64 * NW) void conv32p_kernel(Conv32Params p)
{
    static_assert(!SQ || (SFT && I8), "W8A8 SFT convs come with a W8A8 conv behind them");
    using L = Lay<NPASS, SFT, NW, I8, SQ>;
    constexpr bool PREP = SFT || I8;          // a per-tile pass over the landed halo tile exists
    using T = Til<NW>;
    constexpr int TH = T::TH, NT = T::NT, NPIX = T::NPIX, A_BYTES = T::A_BYTES, C_BYTES = T::C_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sW = smem;
    float *sSS = reinterpret_cast<float *>(smem + L::OFF_SS);
    char *sA = smem + L::OFF_A;
    char *sC = smem + L::OFF_C;
    char *sQ = smem + L::OFF_Q;
    char *sO = smem + L::OFF_OUT;
    const float *sK = reinterpret_cast<const float *>(smem + L::OFF_K);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int ntiles = p.tiles_x * p.tiles_y;
    const unsigned uH = (unsigned)p.H, uW = (unsigned)p.W;
    const unsigned src_guard = (unsigned)p.H * p.W * 64u;     // byte offset of the zero guard behind src (32 ch f16)
    const unsigned cond_guard = (unsigned)p.H * p.W * 32u;    // ... behind cond (16 ch f16)

    // ---- lane constants of the LDS-DMA pieces: halo row/column and byte offset from the halo origin
    int a_pos[T::A_PW], a_off[T::A_PW], c_pos[T::C_PW], c_off[T::C_PW];
#pragma unroll
    for (int it = 0; it < T::A_PW; ++it) {
        const int hp = (wave + it * NW) * 16 + (lane >> 2), slot = lane & 3;
        const int hy = hp / HW, hx = hp - hy * HW;
        a_pos[it] = (hp < NPIX && hx < HC) ? (hy | (hx << 8)) : -1;
        a_off[it] = (hy * p.W + hx) * 64 + ((slot ^ swz32(hx)) << 4);
    }
    if (SFT) {
#pragma unroll
        for (int it = 0; it < T::C_PW; ++it) {
            const int hp = (wave + it * NW) * 32 + (lane >> 1), half = lane & 1;
            const int hy = hp / HW, hx = hp - hy * HW;
            c_pos[it] = (hp < NPIX && hx < HC) ? (hy | (hx << 8)) : -1;
            c_off[it] = (hy * p.W + hx) * 32 + (half << 4);
        }
    }
    auto issue_tile = [&](int t, int buf) {
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        const int pix0 = iy0 * p.W + ix0;                      // may be negative; valid lanes land >= 0
#pragma unroll
        for (int it = 0; it < T::A_PW; ++it) {
            const bool ok = a_pos[it] >= 0 && (unsigned)(iy0 + (a_pos[it] & 255)) < uH && (unsigned)(ix0 + (a_pos[it] >> 8)) < uW;
            const unsigned off = ok ? (unsigned)(pix0 * 64 + a_off[it]) : src_guard + ((lane & 3) << 4);
            dma16(dma_rsrc(p.src), sA + buf * A_BYTES + (wave + it * NW) * 1024, off);
        }
        if (SFT) {
#pragma unroll
            for (int it = 0; it < T::C_PW; ++it) {
                const bool ok = c_pos[it] >= 0 && (unsigned)(iy0 + (c_pos[it] & 255)) < uH && (unsigned)(ix0 + (c_pos[it] >> 8)) < uW;
                const unsigned off = ok ? (unsigned)(pix0 * 32 + c_off[it]) : cond_guard + ((lane & 1) << 4);
                dma16(dma_rsrc(p.cond), sC + buf * C_BYTES + (wave + it * NW) * 1024, off);
            }
        }
    };

    // ---- once per workgroup: the whole weight set and the per-channel scale/shift into LDS
    if constexpr (I8) {
        // int8 weights [tap][COUTP][32 bytes, K order of the code tile]: rows of 32 B, the two 16-byte halves of row n
        // swapped when (n >> 3) & 1 so that a 16-lane ds_read_b128 group never meets two rows 8 apart in the same half
        for (int piece = wave; piece < 9 * L::COUTP / 32; piece += NW) {
            const int r = piece * 32 + (lane >> 1), half = lane & 1;
            const int n = r % L::COUTP;
            dma16(dma_rsrc(p.wpk8), sW + piece * 1024, (unsigned)(r * 32 + ((half ^ ((n >> 3) & 1)) << 4)));
        }
        for (int e = tid; e < 17 * L::COUTP; e += NT) sSS[e] = e < L::COUTP ? p.scale[e] : p.shift[e - L::COUTP];
    } else {
        for (int piece = wave; piece < 9 * L::COUTP / 16; piece += NW) {
            const int r = piece * 16 + (lane >> 2), slot = lane & 3;     // r = tap*COUTP + n
            const int n = r % L::COUTP;
            dma16(dma_rsrc(p.wpk), sW + piece * 1024, (unsigned)(r * 32 + ((slot ^ swz32(n)) << 3)) * 2u);
        }
        for (int e = tid; e < L::COUTP; e += NT) {
            sSS[e] = p.scale[e];
            sSS[L::COUTP + e] = p.shift[e];
        }
    }

    // ---- SFT: lane constants of this wave's 32-pixel groups, fragments and biases
    f16x8 sa0, sa1s, sa1t;
    f32x16 sbh, sbs, sbt;
    int g_pos[T::G_PW], g_c[T::G_PW], g_x[T::G_PW], g_q[T::G_PW];
    i32x4 qa0, qa1s, qa1t;                                     // SQ: int8 A fragments of the three SFT MFMAs
    float cq_inv = 0.f, cq_zoff = 0.f;                         // SQ: this lane half's condition quantiser (lh 0: scale branch, 1: shift branch)
    if constexpr (SQ) {
        const i32x4 *fr = reinterpret_cast<const i32x4 *>(p.sq_wfrag);
        qa0 = fr[lane]; qa1s = fr[64 + lane]; qa1t = fr[128 + lane];
        cq_inv = p.sq_inv[lh]; cq_zoff = p.sq_zoff[lh];
        for (int e = tid; e < 192; e += NT) reinterpret_cast<float *>(smem + L::OFF_K)[e] = p.sq_const[e];
    } else if (SFT) {
        const f16x8 *fr = reinterpret_cast<const f16x8 *>(p.sft_wfrag);
        sa0 = fr[lane]; sa1s = fr[64 + lane]; sa1t = fr[128 + lane];
        sbh = tile16(p.sft_bias, lh); sbs = tile16(p.sft_bias + 32, lh); sbt = tile16(p.sft_bias + 64, lh);
#pragma unroll
        for (int k = 0; k < 16; ++k) sbs[k] += 1.f;            // (scale + 1) enters through the accumulator init
    }
    if (PREP) {
#pragma unroll
        for (int gi = 0; gi < T::G_PW; ++gi) {
            const int hp = (wave + gi * NW) * 32 + l31;
            const int hy = hp / HW, hx = hp - hy * HW;
            g_pos[gi] = (hp < NPIX && hx < HC) ? (hy | (hx << 8)) : -1;
            g_c[gi] = hp * 32 + lh * 16;
            g_x[gi] = hp * 64 + (swz32(hx) << 4) + 8 * lh;      // channel quad qd lives at g_x ^ (qd << 4)
            // code tile: this lane's 16 channels {8qd + 4lh + k} are ONE 16-byte half of the pixel's 32-byte row (the
            // weights' K axis is packed in the same order); the halves swap on odd halo rows (bank conflicts, see ldfrag)
            g_q[gi] = hp * 32 + ((lh ^ (hy & 1)) << 4);
        }
    }
    // y = x*(scale+1)+shift in place on a landed halo tile (arch_util.py:68-72).  A wave owns one or two 32-pixel groups;
    // with only two waves per SIMD the pass is bound by LDS and MFMA latency, not issue, so it runs in three sweeps over
    // the wave's groups: every LDS read first (condition fragment and the four activation quads: all in flight together),
    // then the MLPs (the second group's MFMAs fill the first's result latency), then modulate / quantise and write.
    // Reads for pad slots stay inside the tile buffers (NG * 32 slots fit A_BYTES / C_BYTES); only the writes are masked.
    static_assert(T::NG * 32 * 64 <= A_BYTES && T::NG * 32 * 32 <= C_BYTES, "pad-slot reads stay inside the halo buffers");
    auto sft_groups = [&](auto ngc, int tt, int buf) __attribute__((always_inline)) {
        constexpr int N = decltype(ngc)::value;
        const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        char *a = sA + buf * A_BYTES;
        const char *cbuf = sC + buf * C_BYTES;
        bool inimg[N];
        f16x4 yv[N][4];
        f16x8 c0[N], c1[N];
        f32x16 sc[N], sh[N];
#pragma unroll
        for (int gi = 0; gi < N; ++gi) {
            inimg[gi] = g_pos[gi] >= 0 && (unsigned)(iy0 + (g_pos[gi] & 255)) < uH && (unsigned)(ix0 + (g_pos[gi] >> 8)) < uW;
            if constexpr (SQ) {
                const char *crow = cbuf + g_c[gi] - lh * 16;
                c0[gi] = *reinterpret_cast<const f16x8 *>(crow);
                c1[gi] = *reinterpret_cast<const f16x8 *>(crow + 16);
            } else if constexpr (SFT) {
                c0[gi] = *reinterpret_cast<const f16x8 *>(cbuf + g_c[gi]);
            }
        }
#pragma unroll
        for (int gi = 0; gi < N; ++gi) {           // behind the condition reads: the first MFMA waits for those only
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) yv[gi][qd] = *reinterpret_cast<const f16x4 *>(a + (g_x[gi] ^ (qd << 4)));
        }
        __builtin_amdgcn_sched_barrier(0);         // keep every read above the MLPs (hipcc otherwise sinks group 1's to its MFMA)
#pragma unroll
        for (int gi = 0; gi < N; ++gi) {
            if constexpr (SQ) {
                // W8A8 SFT convs (arch_util.py:60-72 with W8A8Conv2d layers): both first layers read the condition pixel
                // through their own quantiser -> one K = 32 MFMA whose lanes 0..31 carry the scale branch's codes of
                // all 16 channels and lanes 32..63 the shift branch's (weights block-diagonal); hidden rows 0..15 /
                // 16..31 are dequantised, LeakyReLU'd and re-quantised for the second layers in registers.
                i32x4 cb;
                cb[0] = (int)quant4((float)c0[gi][0], (float)c0[gi][1], (float)c0[gi][2], (float)c0[gi][3], cq_inv, cq_zoff);
                cb[1] = (int)quant4((float)c0[gi][4], (float)c0[gi][5], (float)c0[gi][6], (float)c0[gi][7], cq_inv, cq_zoff);
                cb[2] = (int)quant4((float)c1[gi][0], (float)c1[gi][1], (float)c1[gi][2], (float)c1[gi][3], cq_inv, cq_zoff);
                cb[3] = (int)quant4((float)c1[gi][4], (float)c1[gi][5], (float)c1[gi][6], (float)c1[gi][7], cq_inv, cq_zoff);
                i32x16 z16;
#pragma unroll
                for (int k = 0; k < 16; ++k) z16[k] = 0;
                const i32x16 hacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(qa0, cb, z16, 0, 0, 0);
                const float *K = sK + lh * 16;             // [set][lh][16]: hidden scale', shift'; scale-out scale, shift; shift-out scale, shift
                float t[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 ka = *reinterpret_cast<const float4 *>(K + 4 * g), kb = *reinterpret_cast<const float4 *>(K + 32 + 4 * g);
                    const float zo = p.sq_hzoff[g >> 1];
                    const float u0 = (float)hacc[4 * g + 0] * ka.x + kb.x, u1 = (float)hacc[4 * g + 1] * ka.y + kb.y,
                                u2 = (float)hacc[4 * g + 2] * ka.z + kb.z, u3 = (float)hacc[4 * g + 3] * ka.w + kb.w;
                    t[4 * g + 0] = fmaxf(u0, 0.1f * u0) + zo; t[4 * g + 1] = fmaxf(u1, 0.1f * u1) + zo;
                    t[4 * g + 2] = fmaxf(u2, 0.1f * u2) + zo; t[4 * g + 3] = fmaxf(u3, 0.1f * u3) + zo;
                }
                i32x4 hs = {0, 0, 0, 0}, ht = {0, 0, 0, 0};
                hs[0] = (int)quant4u(t[0], t[1], t[2], t[3]);   hs[1] = (int)quant4u(t[4], t[5], t[6], t[7]);
                ht[0] = (int)quant4u(t[8], t[9], t[10], t[11]); ht[1] = (int)quant4u(t[12], t[13], t[14], t[15]);
                const i32x16 a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(qa1s, hs, z16, 0, 0, 0);
                const i32x16 a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(qa1t, ht, z16, 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 k2 = *reinterpret_cast<const float4 *>(K + 64 + 4 * g), k3 = *reinterpret_cast<const float4 *>(K + 96 + 4 * g);
                    const float4 k4 = *reinterpret_cast<const float4 *>(K + 128 + 4 * g), k5 = *reinterpret_cast<const float4 *>(K + 160 + 4 * g);
                    sc[gi][4 * g + 0] = (float)a1[4 * g + 0] * k2.x + k3.x; sc[gi][4 * g + 1] = (float)a1[4 * g + 1] * k2.y + k3.y;
                    sc[gi][4 * g + 2] = (float)a1[4 * g + 2] * k2.z + k3.z; sc[gi][4 * g + 3] = (float)a1[4 * g + 3] * k2.w + k3.w;
                    sh[gi][4 * g + 0] = (float)a2[4 * g + 0] * k4.x + k5.x; sh[gi][4 * g + 1] = (float)a2[4 * g + 1] * k4.y + k5.y;
                    sh[gi][4 * g + 2] = (float)a2[4 * g + 2] * k4.z + k5.z; sh[gi][4 * g + 3] = (float)a2[4 * g + 3] * k4.w + k5.w;
                }
            } else if constexpr (SFT) {
                const f32x16 h = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa0, c0[gi], sbh, 0, 0, 0);
                const f16x8 hs = lrelu_pack16(h, 0), ht = lrelu_pack16(h, 1);
                sc[gi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa1s, hs, sbs, 0, 0, 0);
                sh[gi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sa1t, ht, sbt, 0, 0, 0);
            }
        }
#pragma unroll
        for (int gi = 0; gi < N; ++gi) {
            i32x4 codes;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                f16x4 y = yv[gi][qd];
                if constexpr (SFT) {
                    const f16x4 s1 = cvt_h4(sc[gi][4 * qd], sc[gi][4 * qd + 1], sc[gi][4 * qd + 2], sc[gi][4 * qd + 3]);      // (packed converts: common.h)
                    const f16x4 s0 = cvt_h4(sh[gi][4 * qd], sh[gi][4 * qd + 1], sh[gi][4 * qd + 2], sh[gi][4 * qd + 3]);
                    y = y * s1 + s0;
                }
                if constexpr (I8) {
                    // u8 code q = clamp(rint((y - x_zero) / x_scale), 0, 255) as one FMA + rint + saturating pack
                    const unsigned w = quant4((float)y[0], (float)y[1], (float)y[2], (float)y[3], p.q_inv, p.q_zoff);
                    codes[qd] = inimg[gi] ? (int)w : 0;
                } else {
                    if (!inimg[gi]) { y[0] = (f16)0.f; y[1] = (f16)0.f; y[2] = (f16)0.f; y[3] = (f16)0.f; }
                    yv[gi][qd] = y;
                }
            }
            if (g_pos[gi] >= 0) {
                if constexpr (I8) {
                    *reinterpret_cast<i32x4 *>(sQ + buf * L::Q_BYTES + g_q[gi]) = codes;
                } else {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<f16x4 *>(a + (g_x[gi] ^ (qd << 4))) = yv[gi][qd];
                }
            }
        }
    };
    auto sft_tile = [&](int tt, int buf) __attribute__((always_inline)) {
        static_assert(T::G_PW == 2, "a wave owns one or two groups");
        if (wave + NW < T::NG) sft_groups(std::integral_constant<int, 2>{}, tt, buf);      // wave-uniform
        else if (wave < T::NG) sft_groups(std::integral_constant<int, 1>{}, tt, buf);
    };

    // ---- prologue: tile 0 landed (and SFT-transformed), tile 1 in flight
    int t = blockIdx.x;
    const int step = gridDim.x;
    if (t < ntiles) issue_tile(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (PREP && t < ntiles) sft_tile(t, 0);
    if (t + step < ntiles) issue_tile(t + step, 1);
    __syncthreads();

    // ---- conv fragment addresses (lane constants): activations per kernel column, weights per k-step
    const int q = wave * 32 + l31;                       // this lane's output pixel in the tile
    const int qy = q / TW, qx = q % TW;
    int xoff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xoff[kx] = (qy * HW + qx + kx) * 64 + ((lh ^ swz32(qx + kx)) << 4);
    const int woff = I8 ? l31 * 32 + ((lh ^ ((l31 >> 3) & 1)) << 4) : l31 * 64 + ((lh ^ swz32(l31)) << 4);
    // I8: code-tile fragment of kernel column kx; the 16-byte half is lh on even halo rows, swapped on odd ones, so the
    // address for kernel row ky is (qoff[kx] + ky * HW * 32) ^ ((ky & 1) << 4)
    int qoff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) qoff[kx] = (qy * HW + qx + kx) * 32 + ((lh ^ (qy & 1)) << 4);
    // this thread's two 16-byte output chunks: pixel (q2y[it], q2x) of the tile, channel chunk c8 -- pixels of the wave's OWN two
    // rows (the ones its accumulators hold), so that the staging tile is read back by the wave that wrote it: LDS operations
    // of one wave complete in order, and the store phase of a pass needs no workgroup barrier
    const int c8 = lane & 3;
    int q2y[2];
    const int q2x = lane >> 2;
#pragma unroll
    for (int it = 0; it < 2; ++it) q2y[it] = 2 * wave + it;
    const bool c8_ok = c8 * 8 < p.Cout;
    const float aslope = act_slope(p.act);

    // Steady state, two barriers per tile:
    //   conv(t) -> stage result in LDS -> [DMA(t+1) landed] -> barrier -> stores(t) -> DMA(t+2) issued into
    //   the buffer conv(t) released -> SFT of tile t+1 -> barrier.
    STAMP_DECL;
    for (int buf = 0; t < ntiles; t += step, buf ^= 1) {
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const char *a = sA + buf * A_BYTES;
        // I8: border class of this lane's output pixel = which kernel rows / columns fall outside the image there
        const int bcls = I8 ? ((((oy0 + qy == 0) | ((oy0 + qy == p.H - 1) << 1)) << 2) | ((ox0 + qx == 0) | ((ox0 + qx == p.W - 1) << 1))) & 15 : 0;
        STAMP(7);

        // output element offsets (-1: outside) and residual prefetch
        int ooff[NPASS][2];
        f16x8 rs1[NPASS][2], rs2[NPASS][2];
        if (p.mode != ST_PLANAR3) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int oy = oy0 + q2y[it], ox = ox0 + q2x;
                const bool in = c8_ok && oy < p.H && ox < p.W;
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    int off = -1;
                    if (p.mode == ST_PS) {
                        const int Y = 2 * oy + (pass >> 1), X = 2 * ox + (pass & 1);
                        if (in && Y < p.Hd && X < p.Wd) off = (Y * p.Wd + X) * 32 + c8 * 8;
                    } else if (in) {
                        off = (oy * p.W + ox) * p.dstC + c8 * 8;
                    }
                    ooff[pass][it] = off;
                    f16x8 z;
#pragma unroll
                    for (int k = 0; k < 8; ++k) z[k] = (f16)0.f;
                    rs1[pass][it] = (p.res1 && off >= 0) ? *reinterpret_cast<const f16x8 *>(p.res1 + off) : z;
                    rs2[pass][it] = (p.res2 && off >= 0) ? *reinterpret_cast<const f16x8 *>(p.res2 + off) : z;
                }
            }
        }

        // planar head (conv_last): element offsets and the residual plane values, fetched before the conv
        // so the store phase does not sit behind two dependent global loads
        constexpr int PR = (TH * TW * 3 + NT - 1) / NT;
        long pl_off[PR];
        unsigned pl_res[PR];       // raw f16 bits
        if (p.mode == ST_PLANAR3) {
#pragma unroll
            for (int it = 0; it < PR; ++it) {
                const int e = tid + it * NT;
                const int ch = e / (TH * TW), qq = e % (TH * TW);
                const int oy = oy0 + qq / TW, ox = ox0 + qq % TW;
                const bool ok = e < TH * TW * 3 && oy < p.H && ox < p.W;
                pl_off[it] = ok ? (long)((size_t)ch * p.H * p.W + (size_t)oy * p.W + ox) : -1;
                // unconditional load (offset 0 when masked): a load under a branch is waited for at the branch's end
                // issued as inline asm: hipcc touches (masks / packs) a plain load's result at once, which
                // parks the wave on the load before the conv; the s_waitcnt vmcnt(0) in front of barrier 1
                // covers it, and the value is first read after that barrier
                const f16 *rp = (p.res_planar ? p.res_planar : p.src) + (ok ? pl_off[it] : 0);
                asm volatile("global_load_ushort %0, %1, off" : "=v"(pl_res[it]) : "v"(rp) : "memory");
            }
        }
        STAMP(0);      // offsets + residual prefetch
        // fp16 multi-pass layers (the up-convs) run their passes two at a time: an activation fragment read from LDS feeds
        // both passes' MFMAs (54 LDS reads per pass pair instead of 72)
        constexpr int PB = (!I8 && NPASS % 2 == 0) ? 2 : 1;
        f32x16 accb[PB];
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            f32x16 acc;
            if constexpr (PB == 2) {
                if ((pass & 1) == 0) {
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int k = 0; k < 16; ++k) accb[b][k] = 0.f;
                    f16x8 wfr[2][18], xfr[18];
                    auto ldfrag = [&](int st) {
                        const int tap = st >> 1, ks = st & 1;
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            wfr[b][st] = *reinterpret_cast<const f16x8 *>(sW + (woff ^ (ks << 5)) + (tap * L::COUTP + (pass + b) * 32) * 64);
                        xfr[st] = *reinterpret_cast<const f16x8 *>(a + (xoff[tap % 3] ^ (ks << 5)) + (tap / 3) * HW * 64);
                    };
                    ldfrag(0); ldfrag(1); ldfrag(2);
#pragma unroll
                    for (int st = 0; st < 18; ++st) {
                        if (st + 3 < 18) ldfrag(st + 3);
#pragma unroll
                        for (int b = 0; b < 2; ++b) accb[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfr[b][st], xfr[st], accb[b], 0, 0, 0);
                    }
                    // pin the interleave: 9 reads up front, then {2 MFMAs, 3 reads} x 15, then 6 MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
#pragma unroll
                    for (int st = 0; st < 15; ++st) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                }
                acc = accb[pass & 1];
            } else if constexpr (I8) {
                // 9 k-steps (one tap = 32 input channels); fragment reads run three steps ahead of their MFMA
                const char *qa = sQ + buf * L::Q_BYTES;
                i32x16 iacc;
#pragma unroll
                for (int k = 0; k < 16; ++k) iacc[k] = 0;
                i32x4 wq[9], xq[9];
                auto ldq = [&](int tap) {
                    wq[tap] = *reinterpret_cast<const i32x4 *>(sW + woff + (tap * L::COUTP + pass * 32) * 32);
                    xq[tap] = *reinterpret_cast<const i32x4 *>(qa + ((qoff[tap % 3] + (tap / 3) * HW * 32) ^ (((tap / 3) & 1) << 4)));
                };
                ldq(0); ldq(1); ldq(2);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if (tap + 3 < 9) ldq(tap + 3);
                    iacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wq[tap], xq[tap], iacc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = (float)iacc[k];
            } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            // 18 k-steps (9 taps x 2); fragment reads run three steps ahead of the MFMA that consumes them
            f16x8 wfr[18], xfr[18];
            auto ldfrag = [&](int st) {
                const int tap = st >> 1, ks = st & 1;
                wfr[st] = *reinterpret_cast<const f16x8 *>(sW + (woff ^ (ks << 5)) + (tap * L::COUTP + pass * 32) * 64);
                xfr[st] = *reinterpret_cast<const f16x8 *>(a + (xoff[tap % 3] ^ (ks << 5)) + (tap / 3) * HW * 64);
            };
            ldfrag(0); ldfrag(1); ldfrag(2);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                if (st + 3 < 18) ldfrag(st + 3);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfr[st], xfr[st], acc, 0, 0, 0);
            }
            // pin the interleave (hipcc otherwise sinks every read pair down to its MFMA and waits
            // lgkmcnt(0) eighteen times): 6 reads up front, then {1 MFMA, 2 reads} x 15, then 3 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int st = 0; st < 15; ++st) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
            STAMP(1);  // conv MFMAs
            // ---- this pass's 32 channels x (TH*16) pixels into the LDS staging tile
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int cl = 8 * qd + 4 * lh;
                const float4 sc = *reinterpret_cast<const float4 *>(sSS + pass * 32 + cl);
                const float4 sh = *reinterpret_cast<const float4 *>(sSS + L::COUTP + (I8 ? bcls * L::COUTP : 0) + pass * 32 + cl);
                f16x4 o;
                o[0] = (f16)act_fast(acc[4 * qd + 0] * sc.x + sh.x, aslope);
                o[1] = (f16)act_fast(acc[4 * qd + 1] * sc.y + sh.y, aslope);
                o[2] = (f16)act_fast(acc[4 * qd + 2] * sc.z + sh.z, aslope);
                o[3] = (f16)act_fast(acc[4 * qd + 3] * sc.w + sh.w, aslope);
                *reinterpret_cast<f16x4 *>(sO + q * OUT_ROWB + cl * 2) = o;
            }
            STAMP(2);  // staging write
            // The first pass makes sure the next tile's LDS-DMA and this tile's residual prefetches have landed (vmcnt counts
            // in order and the DMA is the older one, so waiting for the residuals waits for it anyway).  The builtin, not
            // inline asm: hipcc's waitcnt pass then KNOWS nothing is pending and puts no vmcnt wait into any store phase --
            // where a vmcnt(0) in front of the second chunk would also wait for the first chunk's store to complete.
            if (pass == 0) __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0), expcnt / lgkmcnt untouched
            STAMP(3);  // wait for DMA(t+1) / outstanding memory ops
            // one barrier per tile here (tile t+1 has landed for everyone); the planar head reads other waves' pixels from the
            // staging tile and needs it for that, too
            if (pass == 0) __syncthreads();
            STAMP(4);  // barrier 1
            if (p.mode == ST_PLANAR3) {
#pragma unroll
                for (int it = 0; it < PR; ++it) {
                    if (pl_off[it] >= 0) {
                        const int e = tid + it * NT;
                        const int ch = e / (TH * TW), qq = e % (TH * TW);
                        const float v = (float)*reinterpret_cast<const f16 *>(sO + qq * OUT_ROWB + ch * 2) + (p.res_planar ? (float)__builtin_bit_cast(f16, (unsigned short)pl_res[it]) : 0.f);
                        p.dst_planar[pl_off[it]] = (f16)v;
                    }
                }
            } else {
                // both staging chunks first (one LDS latency, not two), then the adds and the stores
                f16x8 v[2];
#pragma unroll
                for (int it = 0; it < 2; ++it) v[it] = *reinterpret_cast<const f16x8 *>(sO + (q2y[it] * TW + q2x) * OUT_ROWB + c8 * 16);
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    // residual adds in packed f16, one rounding per add as the reference's fp16 model does
                    // (x + conv2(..) then + skip, arch_util.py:95, HDRUNet3T1_arch.py:186-198); 6 VALU per 8
                    // values instead of ~40 through fp32 -- this kernel is instruction-issue-bound
                    v[it] = (v[it] + rs1[pass][it]) + rs2[pass][it];
                    if (ooff[pass][it] >= 0) *reinterpret_cast<f16x8 *>(p.dst + ooff[pass][it]) = v[it];
                }
            }
            STAMP(5);  // DMA issue + stores
        }
        __syncthreads();                                // every wave is done with this tile's halo buffer (and, planar head, the staging tile)
        // conv(t) released its halo buffer at the barrier: tile t+2 goes in flight AFTER the stores (an
        // LDS-DMA in flight makes hipcc wait vmcnt(0) at the next use of any plain load result -- the
        // prefetched residuals -- which would park the store phase on the DMA) and BEFORE the SFT of
        // tile t+1, so it has a whole tile period to land
        if (t + 2 * step < ntiles) issue_tile(t + 2 * step, buf);
        if (PREP && t + step < ntiles) sft_tile(t + step, buf ^ 1);
        STAMP(6);      // SFT of the next tile
        __syncthreads();
    }
#ifdef HDRTV_STAMP
    if (p.dump && lane == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long *>(p.dump)[((size_t)blockIdx.x * NW + wave) * 8 + i] = st_acc[i];
#endif
}

template <int NPASS, bool SFT, int NW, bool I8 = false, bool SQ = false>
hipError_t launch_t(const Conv32Params &p, int n_cu, hipStream_t s)
{
    using L = Lay<NPASS, SFT, NW, I8, SQ>;
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv32p_kernel<NPASS, SFT, NW, I8, SQ>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const int ntiles = p.tiles_x * p.tiles_y;
    const long cap = (long)n_cu * (160 * 1024 / L::SMEM);  // persistent: as many workgroups as fit the chip
    const int grid = ntiles < cap ? ntiles : (int)cap;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), L::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

// src (and cond) must be followed by >= 64 zero bytes (the workspace guard): out-of-image halo lanes read them.
// nw: the caller's A/B switch -- 0 = per layer (below), 8 = 16x16 tiles, 4 = 8x16 tiles x 2 workgroups per CU
hipError_t conv32p_launch(Conv32Params p, int n_cu, hipStream_t s, int nw)
{
    if ((size_t)p.H * p.W * 64 >= 0xf0000000ull) return hipErrorInvalidValue;     // 32-bit byte offsets
    const bool sft = p.cond != nullptr;
    p.tiles_x = (p.W + TW - 1) / TW;
#ifndef HDRTV_AB
    // the shipped library keeps the up-convs (32 -> 128 + PixelShuffle) here; the single-pass layers run on conv32s.hip and
    // le_rows.hip, and their conv32p forms (the bit-identity yardstick of both) exist in the A/B library only (make AB=1)
    (void)nw;
    if (p.CoutPad != 128 || sft || p.sq_wfrag) return hipErrorNotSupported;
    p.tiles_y = (p.H + 15) / 16;
    return p.wpk8 ? launch_t<4, false, 8, true>(p, n_cu, s) : launch_t<4, false, 8>(p, n_cu, s);
#else
    if (p.wpk8) {                                        // W8A8 layer: int8 MFMA on the quantised tile, 16x16 tiles only
        p.tiles_y = (p.H + 15) / 16;
        if (p.CoutPad == 32 && sft && p.sq_wfrag) return launch_t<1, true, 8, true, true>(p, n_cu, s);
        if (p.CoutPad == 32) return sft ? launch_t<1, true, 8, true>(p, n_cu, s) : launch_t<1, false, 8, true>(p, n_cu, s);
        if (p.CoutPad == 128 && !sft) return launch_t<4, false, 8, true>(p, n_cu, s);
        return hipErrorInvalidValue;
    }
    if (p.sq_wfrag) return hipErrorInvalidValue;         // W8A8 SFT convs in front of an fp16 conv: no kernel (no shipped recipe has it)
    // conv_last (32 -> 3, no SFT, planar store) is all per-tile latency: two 4-wave workgroups per CU hide it better
    // (0.286 -> 0.251 ms at 4K); every other layer is faster with the 16x16 tile
    const bool small_tile = nw == 4 || (p.CoutPad == 32 && !sft && p.mode == ST_PLANAR3 && nw == 0);
    if (!small_tile || p.CoutPad == 128) {               // the 72 KiB weight set of the up-convs leaves room for one workgroup only
        p.tiles_y = (p.H + 15) / 16;
        if (p.CoutPad == 32) return sft ? launch_t<1, true, 8>(p, n_cu, s) : launch_t<1, false, 8>(p, n_cu, s);
        if (p.CoutPad == 128 && !sft) return launch_t<4, false, 8>(p, n_cu, s);
        return hipErrorInvalidValue;
    }
    p.tiles_y = (p.H + 7) / 8;
    if (p.CoutPad == 32) return sft ? launch_t<1, true, 4>(p, n_cu, s) : launch_t<1, false, 4>(p, n_cu, s);
    return hipErrorInvalidValue;
#endif
}
