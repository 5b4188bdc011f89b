// conv32s.hip -- the 32 -> 32 (and 32 -> 3) 3x3 / stride-1 convolutions of the LE main branch as ONE-barrier-per-tile
// persistent kernel with the SFT layer in front fused in (gfx950).  Same layers, operands, arithmetic and results (bit for
// bit) as conv32p.hip's single-pass variants, which it replaces; conv32p keeps the 32 -> 128 up-convs.
//
// Reference: every `conv(sft(x, cond))` pair of HDRUNet3T1 (SFT_layer1 -> HR_conv1, SFT_layer2 -> HR_conv2,
// ResBlock_with_SFT's sft1 -> conv1 and sft2 -> conv2; HDRUNet3T1_arch.py:168-200, arch_util.py:60-95) and conv_last;
// W8A8 layers as W8A8Conv2d.forward (hdrtvnet_torch.py:351-364), scheme in conv32p.hip's header.
//
// Why a second schedule.  conv32p runs a tile as  conv | barrier | stores, DMA issue, SFT of the next tile | barrier:
// all eight waves are in the MFMA phase together and in the VALU / LDS-latency phase (SFT) together, two waves per
// SIMD, so the matrix pipe idles through one phase and the vector pipe through the other (stamps: conv 14 %, SFT 39 %,
// barriers 15 % of the wave cycles).  Here the two phases of a tile period overlap:
//   * the per-tile work is  R: residual loads   X: LDS-DMA of tile t+2   M: the conv MFMAs of tile t   P: SFT (or quantise)
//     pass over tile t+1's halo   E: epilogue + stores of tile t, split by ROLE (template SPLIT; every layer with a P pass
//     except those with W8A8 SFT convs): waves 0-3 are conv waves -- R M E over 64 output pixels each, two 32-pixel groups
//     that share every weight fragment -- and waves 4-7 prep waves -- X and P, three 32-slot halo groups each.  A SIMD holds
//     one of each: the conv wave feeds the MFMA pipe while the prep wave is in the VALU / LDS-latency part, and neither
//     waits for the other's memory operations (a conv wave issues no DMA, a prep wave no loads or stores).  Without the
//     split (layers without a P pass; W8A8 SFT convs, whose P pass is too heavy for four waves; nosplit) every
//     wave does both: waves 0-3 run R X M P E, waves 4-7 R P M E, one conv group each, two halo groups on waves 4-7 and one
//     on waves 0-3;
//   * ONE barrier per tile.  That takes (a) three halo buffers for fp16 layers: the conv reads A[t], P rewrites A[t+1]
//     in place, the DMA lands in A[t+2] (W8A8 layers: the conv reads the code tile, two of each suffice); (b) an epilogue
//     that needs no workgroup synchronisation: a wave transposes ITS pixels x 32 channels through a wave-private
//     strip (LDS operations of one wave complete in order) and stores them itself, 16 B per lane, 64 B runs per pixel;
//   * vmcnt: hipcc does not count across LDS-DMA -- it waits vmcnt(0) in front of the first use of a plain load issued
//     around a DMA and in front of the first LDS read it can name while a DMA is in flight (here: the scale / shift
//     tables).  Both are therefore kept in E, the last thing in the tile; a wave that has issued DMA closes the tile with
//     vmcnt(NSTORE) (every wave issues exactly NSTORE stores per tile, unconditionally -- masked-off lanes store to a trash
//     line, their residual loads read a zero line -- and the DMA is older than them, so stores are never waited for), a
//     prep wave of the split with vmcnt(0), a conv wave of the split with no vector-memory wait at all.  (Inline-asm
//     loads with hand-counted waits were tried for the residuals: hipcc copied the destination registers in front of the
//     wait -- a read of data that had not landed, visible as sporadic wrong half-waves at 1080p; tests/test_isa_contracts.py
//     pins the structure.)
// Stamps (make STAMP=1, profiles/r02_conv32s_stamps.txt) drove the steps: first form (every wave issues DMA, does M and P) ->
// the tile's critical path sat in waves 4-7 and every wave queued in the address unit behind the barrier -> DMA issue on
// waves 0-3 -> role split (the prep waves are now the critical path: DMA issue 26 %, P 64 % of their cycles; the conv waves
// wait 20 % of theirs at the barrier).
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

constexpr int TW = 16, TH = 16, HC = TW + 2, HW = 20, NW = 8, NT = 64 * NW;
constexpr int NPIX = (TH + 2) * HW;                    // 360 halo slots (324 real)
constexpr int NG = (NPIX + 31) / 32;                   // 12 groups of 32 slots
// LDS-DMA is issued by waves 0-3 only (they have one SFT group to the two of waves 4-7: stamps put the tile's critical path
// in waves 4-7, and all eight waves issuing at once after the barrier queue behind each other in the address unit)
constexpr int NWI = 4, A_PW = 6, C_PW = 3;             // 1-KiB pieces per issuing wave: 16 px x 64 B / 32 px x 32 B
constexpr int A_BYTES = NWI * A_PW * 1024, C_BYTES = NWI * C_PW * 1024;
constexpr int OUT_ROWB = 64 + 16, STRIP = 32 * OUT_ROWB;
static_assert(NWI * A_PW * 16 >= NPIX && NWI * C_PW * 32 >= NPIX && NG * 32 * 64 <= A_BYTES && NG * 32 * 32 <= C_BYTES, "halo buffers");

__device__ __forceinline__ int swz32(int v) { return (v >> 2) & 3; }


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

__device__ __forceinline__ f16x8 lrelu_pack16(const f32x16 &a, int s)
{
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)a[8 * s + j];
    return __builtin_elementwise_max(o, o * (f16)0.1f);
}

// C3 (HR_conv1 with conv_first fused in front): the 3-plane image patch of a tile, 20 x 20 pixels of 4 channels (8 bytes;
// pitch 24: columns 20..23 are the kx4 = 3 / pad-slot dummy reads and stay zero), double-buffered
constexpr int P3_H = TH + 4, P3_W = TW + 4, P3_PW = 24, P3_BYTES = P3_H * P3_PW * 8;

template <bool SFT, bool I8, bool SQ, bool C3 = false>
struct Lay {
    static constexpr int NA = (I8 || C3) ? 2 : 3;                        // halo buffers (see header; C3: nothing is DMA'd into them)
    static constexpr int W_BYTES = 9 * 32 * (I8 ? 32 : 64);
    static constexpr int SS_BYTES = I8 ? 32 * 4 * 17 : 32 * 8;          // I8: scale[32] + shift[16 border classes][32]
    static constexpr int Q_BYTES = I8 ? NG * 32 * 32 : 0;               // int8 code tile: 32 B per halo slot
    static constexpr int OFF_A = W_BYTES;
    static constexpr int OFF_C = OFF_A + NA * A_BYTES;
    static constexpr int OFF_Q = OFF_C + (SFT ? 2 * C_BYTES : 0);
    static constexpr int OFF_P3 = OFF_Q + 2 * Q_BYTES;
    static constexpr int OFF_OUT = OFF_P3 + (C3 ? 2 * P3_BYTES : 0);
    static constexpr int SMEM = OFF_OUT + NW * STRIP;                   // dynamic part; the constant tables are static arrays
    static_assert(SMEM + SS_BYTES + (SQ ? 768 : 0) <= 160 * 1024, "LDS budget");
};

// s_waitcnt immediate of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

template <bool SFT, bool I8, bool SQ, bool PLANAR, bool C3 = false, bool SPLIT = false>
__global__ __launch_bounds__(NT) void conv32s_kernel(Conv32Params p)
{
    static_assert(!SQ || (SFT && I8), "W8A8 SFT convs come with a W8A8 conv behind them");
    static_assert(!C3 || (SFT && !I8 && !PLANAR), "conv_first is fused in front of SFT_layer1 + HR_conv1 (fp16) only");
    static_assert(!SPLIT || SFT || I8, "the role split needs a P pass to give to waves 4-7");
    using L = Lay<SFT, I8, SQ, C3>;
    constexpr bool PREP = SFT || I8;
    constexpr int NA = L::NA;
    // SPLIT: waves 0-3 are conv waves (R M E over 64 output pixels = two 32-pixel groups each), waves 4-7 prep waves (the
    // LDS-DMA and the P pass, three halo groups each).  Otherwise every wave does both (one conv group; two or one halo groups).
    constexpr int NQ = SPLIT ? 2 : 1;                  // 32-pixel conv groups of a conv wave
    constexpr int NSTORE = (PLANAR ? 3 : 2) * NQ;      // stores per conv wave and tile, issued unconditionally
        extern __shared__ __attribute__((aligned(16))) char smem[];
    // scale / shift (and the W8A8 SFT constants) live in STATIC LDS arrays: hipcc's waitcnt pass makes every LDS read
    // that may alias an LDS-DMA destination wait vmcnt(0) -- a read of these tables in the epilogue would wait for the
    // next tile's DMA; distinct objects cannot alias the dynamic buffer the DMA writes
    __shared__ __attribute__((aligned(16))) float sSS[L::SS_BYTES / 4];
    __shared__ __attribute__((aligned(16))) float sKw[SQ ? 192 : 4];
    char *sW = smem;
    char *sA = smem + L::OFF_A;
    char *sC = smem + L::OFF_C;
    char *sQ = smem + L::OFF_Q;
    f16x4 *sP3 = reinterpret_cast<f16x4 *>(smem + L::OFF_P3);
    const float *sK = sKw;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    char *strip = smem + L::OFF_OUT + wave * (NQ * STRIP);           // SPLIT: waves 0-3 only, 64 pixels each
    const int iw = wave & 3;                                          // index among the four DMA-issuing waves
    // who issues the LDS-DMA: without the split waves 0-3.  With it the prep waves -- a conv wave that issues DMA pays for it
    // in its epilogue, where hipcc's vmcnt(0) in front of the residuals then also waits for the pieces it has just issued
    // (measured: condition tile on the conv waves = the 17 SFT launches 0.98 -> 1.02 ms) -- except with conv_first fused in,
    // where the prep waves carry the extra conv and only the three condition pieces exist (0.385 -> 0.37 ms on the conv waves)
    const bool issue_a = SPLIT ? wave >= 4 : wave < NWI, issue_c = (SPLIT && !C3) ? wave >= 4 : wave < NWI;
    const int ntiles = p.tiles_x * p.tiles_y;
    const unsigned uH = (unsigned)p.H, uW = (unsigned)p.W;
    const unsigned src_guard = (unsigned)p.H * p.W * 64u;     // byte offset of the zero guard behind src (32 ch f16)
    const unsigned cond_guard = (unsigned)p.H * p.W * 32u;    // ... behind cond (16 ch f16)

    // ---- lane constants of the LDS-DMA pieces: halo row/column and byte offset from the halo origin
    int a_pos[A_PW], a_off[A_PW], c_pos[C_PW], c_off[C_PW];
#pragma unroll
    for (int it = 0; it < A_PW; ++it) {
        const int hp = (iw + it * NWI) * 16 + (lane >> 2), slot = lane & 3;
        const int hy = hp / HW, hx = hp - hy * HW;
        a_pos[it] = (hp < NPIX && hx < HC) ? (hy | (hx << 8)) : -1;
        a_off[it] = (hy * p.W + hx) * 64 + ((slot ^ swz32(hx)) << 4);
    }
    if (SFT) {
#pragma unroll
        for (int it = 0; it < C_PW; ++it) {
            const int hp = (iw + it * NWI) * 32 + (lane >> 1), half = lane & 1;
            const int hy = hp / HW, hx = hp - hy * HW;
            c_pos[it] = (hp < NPIX && hx < HC) ? (hy | (hx << 8)) : -1;
            c_off[it] = (hy * p.W + hx) * 32 + (half << 4);
        }
    }
    // no branch inside: past the end the last tile is fetched again and never used
    auto issue_tile = [&](int tq, char *abuf, char *cbuf) __attribute__((always_inline)) {
        const int t = tq < ntiles ? tq : ntiles - 1;
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        const int pix0 = iy0 * p.W + ix0;                      // may be negative; valid lanes land >= 0
        // a tile whose halo lies inside the image (all but the border tiles) needs no per-lane image test: two instructions
        // per piece instead of eight (stamps: the issue was a quarter of a prep wave's tile time)
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + TH + 2 <= p.H && ix0 + HC <= p.W;     // wave-uniform
        if (!C3 && issue_a) {
            if (interior) {
#pragma unroll
                for (int it = 0; it < A_PW; ++it) {
                    const unsigned off = a_pos[it] >= 0 ? (unsigned)(pix0 * 64 + a_off[it]) : src_guard + ((lane & 3) << 4);
                    dma16(dma_rsrc(p.src), abuf + (iw + it * NWI) * 1024, off);
                }
            } else {
#pragma unroll
                for (int it = 0; it < A_PW; ++it) {
                    const bool ok = a_pos[it] >= 0 && (unsigned)(iy0 + (a_pos[it] & 255)) < uH && (unsigned)(ix0 + (a_pos[it] >> 8)) < uW;
                    const unsigned off = ok ? (unsigned)(pix0 * 64 + a_off[it]) : src_guard + ((lane & 3) << 4);
                    dma16(dma_rsrc(p.src), abuf + (iw + it * NWI) * 1024, off);
                }
            }
        }
        if (SFT && issue_c) {
            if (interior) {
#pragma unroll
                for (int it = 0; it < C_PW; ++it) {
                    const unsigned off = c_pos[it] >= 0 ? (unsigned)(pix0 * 32 + c_off[it]) : cond_guard + ((lane & 1) << 4);
                    dma16(dma_rsrc(p.cond), cbuf + (iw + it * NWI) * 1024, off);
                }
            } else {
#pragma unroll
                for (int it = 0; it < C_PW; ++it) {
                    const bool ok = c_pos[it] >= 0 && (unsigned)(iy0 + (c_pos[it] & 255)) < uH && (unsigned)(ix0 + (c_pos[it] >> 8)) < uW;
                    const unsigned off = ok ? (unsigned)(pix0 * 32 + c_off[it]) : cond_guard + ((lane & 1) << 4);
                    dma16(dma_rsrc(p.cond), cbuf + (iw + it * NWI) * 1024, off);
                }
            }
        }
    };

    // ---- once per workgroup: the whole weight set and the per-channel scale/shift into LDS
    if constexpr (I8) {
        // int8 weights [tap][32][32 bytes, K order of the code tile]: the two 16-byte halves of row n swapped when
        // (n >> 3) & 1 so that a 16-lane ds_read_b128 group never meets two rows 8 apart in the same half
        for (int piece = wave; piece < 9; piece += NW) {
            const int r = piece * 32 + (lane >> 1), half = lane & 1;
            const int n = r % 32;
            dma16(dma_rsrc(p.wpk8), sW + piece * 1024, (unsigned)(r * 32 + ((half ^ ((n >> 3) & 1)) << 4)));
        }
        for (int e = tid; e < 17 * 32; e += NT) sSS[e] = e < 32 ? p.scale[e] : p.shift[e - 32];
    } else {
        for (int piece = wave; piece < 18; piece += NW) {
            const int r = piece * 16 + (lane >> 2), slot = lane & 3;     // r = tap*32 + n
            const int n = r % 32;
            dma16(dma_rsrc(p.wpk), sW + piece * 1024, (unsigned)(r * 32 + ((slot ^ swz32(n)) << 3)) * 2u);
        }
        for (int e = tid; e < 32; e += NT) {
            sSS[e] = p.scale[e];
            sSS[32 + e] = p.shift[e];
        }
    }

    // ---- C3: this thread's pixel of a tile's 20 x 20 image patch (threads 0..399), fetched into registers in R and
    // written to LDS in E, two tiles ahead of the pass that reads it
    const int p3r = tid / P3_W, p3c = tid - p3r * P3_W;
    f16 p3v[3];
    auto p3_fetch = [&](int tq) __attribute__((always_inline)) {
        const int t = tq < ntiles ? tq : ntiles - 1;
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int iy = ty * TH - 2 + p3r, ix = tx * TW - 2 + p3c;
        const bool ok = tid < P3_H * P3_W && (unsigned)iy < uH && (unsigned)ix < uW;
        const size_t o = ok ? (size_t)iy * p.W + ix : 0;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            p3v[ch] = p.c3_img[(size_t)ch * p.H * p.W + o];
            if (!ok) p3v[ch] = (f16)0.f;
        }
    };
    auto p3_stage = [&](int buf) __attribute__((always_inline)) {
        if (tid < P3_H * P3_W) sP3[buf * (P3_BYTES / 8) + p3r * P3_PW + p3c] = f16x4{p3v[0], p3v[1], p3v[2], (f16)1.f};   // 1: the bias slot
    };
    f16x8 c3w[3];                                            // conv_first's A fragments (kernel rows), le_hg_misc.hip's K order
    if constexpr (C3) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) c3w[ky] = reinterpret_cast<const f16x8 *>(p.c3_wfrag)[ky * 64 + lane];
        for (int e = tid; e < 2 * P3_H * (P3_PW - P3_W); e += NT) {         // the dummy columns stay zero
            const int b = e / (P3_H * (P3_PW - P3_W)), r = (e / (P3_PW - P3_W)) % P3_H, cc = e % (P3_PW - P3_W);
            sP3[b * (P3_BYTES / 8) + r * P3_PW + P3_W + cc] = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        }
    }

    // ---- SFT / quantise pass: lane constants of this wave's 32-slot groups, fragments and biases
    constexpr int G_PW = SPLIT ? 3 : 2;
    // SPLIT: prep wave w takes groups w-4, w, w+4.  Otherwise waves 4-7: groups w-4 and w; waves 0-3: group 8+w
    const int gid[3] = {SPLIT ? iw : (wave >= 4 ? wave - 4 : 8 + wave), SPLIT ? iw + 4 : wave, iw + 8};
    f16x8 sa0, sa1s, sa1t;
    f32x16 sbh, sbs, sbt;
    int g_pos[3], g_c[3], g_x[3], g_q[3], g_p3[3];
    i32x4 qa0, qa1s, qa1t;
    float cq_inv = 0.f, cq_zoff = 0.f;
    if constexpr (SQ) {
        const i32x4 *fr = reinterpret_cast<const i32x4 *>(p.sq_wfrag);
        qa0 = fr[lane]; qa1s = fr[64 + lane]; qa1t = fr[128 + lane];
        cq_inv = p.sq_inv[lh]; cq_zoff = p.sq_zoff[lh];
        for (int e = tid; e < 192; e += NT) sKw[e] = p.sq_const[e];
    } else if (SFT) {
        const f16x8 *fr = reinterpret_cast<const f16x8 *>(p.sft_wfrag);
        sa0 = fr[lane]; sa1s = fr[64 + lane]; sa1t = fr[128 + lane];
        sbh = tile16(p.sft_bias, lh); sbs = tile16(p.sft_bias + 32, lh); sbt = tile16(p.sft_bias + 64, lh);
#pragma unroll
        for (int k = 0; k < 16; ++k) sbs[k] += 1.f;            // (scale + 1) enters through the accumulator init
    }
    if (PREP) {
#pragma unroll
        for (int gi = 0; gi < G_PW; ++gi) {
            const int hp = gid[gi] * 32 + l31;
            const int hy = hp / HW, hx = hp - hy * HW;
            g_pos[gi] = (hp < NPIX && hx < HC) ? (hy | (hx << 8)) : -1;
            g_c[gi] = hp * 32 + lh * 16;
            g_x[gi] = hp * 64 + (swz32(hx) << 4) + 8 * lh;      // channel quad qd lives at g_x ^ (qd << 4)
            g_q[gi] = hp * 32 + ((lh ^ (hy & 1)) << 4);         // code tile: halves swap on odd halo rows
            // C3: halo pixel (hy, hx) is patch pixel (hy + 1, hx + 1); the fragment of kernel row ky starts one up / left
            g_p3[gi] = g_pos[gi] >= 0 ? hy * P3_PW + hx + 2 * lh : 2 * lh;
        }
    }
    // y = x*(scale+1)+shift in place on a landed halo tile (arch_util.py:68-72), three sweeps over the wave's groups:
    // every LDS read, the MLPs, modulate / quantise and write (see conv32p.hip)
    auto sft_groups = [&](auto ngc, auto g0c, int tt, char *a, const char *cbuf, char *qbuf, const f16x4 *p3) __attribute__((always_inline)) {
        constexpr int N = decltype(ngc)::value, G0 = decltype(g0c)::value;    // this call: groups G0 .. G0 + N - 1 of the wave
        const int ty = tt / p.tiles_x, tx = tt - ty * p.tiles_x;
        const int iy0 = ty * TH - 1, ix0 = tx * TW - 1;
        bool inimg[N];
        f16x4 yv[N][4];
        f16x8 c0[N], c1[N];
        f32x16 sc[N], sh[N];
        f16x4 s1p[N][4], s0p[N][4];                        // scale + 1 and shift rounded to f16 as soon as their MFMAs are done: a
                                                           // third of the registers of the fp32 tiles, so three groups fit at once
        auto pack_ss = [&](int gi) __attribute__((always_inline)) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                {       // (packed converts: common.h cvt_h4)
                    s1p[gi][qd] = cvt_h4(sc[gi][4 * qd], sc[gi][4 * qd + 1], sc[gi][4 * qd + 2], sc[gi][4 * qd + 3]);
                    s0p[gi][qd] = cvt_h4(sh[gi][4 * qd], sh[gi][4 * qd + 1], sh[gi][4 * qd + 2], sh[gi][4 * qd + 3]);
                }
        };
#pragma unroll
        for (int gi = 0; gi < N; ++gi) {
            inimg[gi] = g_pos[G0 + gi] >= 0 && (unsigned)(iy0 + (g_pos[G0 + gi] & 255)) < uH && (unsigned)(ix0 + (g_pos[G0 + gi] >> 8)) < uW;
            if constexpr (SQ) {
                const char *crow = cbuf + g_c[G0 + gi] - lh * 16;
                c0[gi] = *reinterpret_cast<const f16x8 *>(crow);
                c1[gi] = *reinterpret_cast<const f16x8 *>(crow + 16);
            } else if constexpr (SFT) {
                c0[gi] = *reinterpret_cast<const f16x8 *>(cbuf + g_c[G0 + gi]);
            }
        }
        f16x8 xc3[N][3];
#pragma unroll
        for (int gi = 0; gi < N; ++gi) {
            if constexpr (C3) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const f16x4 *q = p3 + g_p3[G0 + gi] + ky * P3_PW;
                    const f16x4 u = q[0], v = q[1];
                    xc3[gi][ky] = f16x8{u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
                }
            } else {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) yv[gi][qd] = *reinterpret_cast<const f16x4 *>(a + (g_x[G0 + gi] ^ (qd << 4)));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (C3) {
            // conv_first (3x3, 3 -> 32, bias, ReLU; HDRUNet3T1_arch.py:168) on the halo pixels, exactly as conv_c3<32> computes
            // it (same K order, the bias in the centre tap's 4th-channel slot, one rounding to f16): the tile SFT_layer1 modulates
#pragma unroll
            for (int gi = 0; gi < N; ++gi) {
                f32x16 h;
#pragma unroll
                for (int k = 0; k < 16; ++k) h[k] = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) h = __builtin_amdgcn_mfma_f32_32x32x16_f16(c3w[ky], xc3[gi][ky], h, 0, 0, 0);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd)
#pragma unroll
                    for (int k = 0; k < 4; ++k) yv[gi][qd][k] = (f16)act_fast(h[4 * qd + k], 0.f);     // bias: inside the sum (K slot)
            }
        }
#pragma unroll
        for (int gi = 0; gi < N; ++gi) {
            if constexpr (SQ) {
                // W8A8 SFT convs: one K = 32 MFMA for both first layers (block-diagonal), hidden rows dequantised,
                // LeakyReLU'd and re-quantised in registers for the second layers (conv32p.hip)
                i32x4 cb;
                cb[0] = (int)quant4((float)c0[gi][0], (float)c0[gi][1], (float)c0[gi][2], (float)c0[gi][3], cq_inv, cq_zoff);
                cb[1] = (int)quant4((float)c0[gi][4], (float)c0[gi][5], (float)c0[gi][6], (float)c0[gi][7], cq_inv, cq_zoff);
                cb[2] = (int)quant4((float)c1[gi][0], (float)c1[gi][1], (float)c1[gi][2], (float)c1[gi][3], cq_inv, cq_zoff);
                cb[3] = (int)quant4((float)c1[gi][4], (float)c1[gi][5], (float)c1[gi][6], (float)c1[gi][7], cq_inv, cq_zoff);
                i32x16 z16;
#pragma unroll
                for (int k = 0; k < 16; ++k) z16[k] = 0;
                const i32x16 hacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(qa0, cb, z16, 0, 0, 0);
                const float *K = sK + lh * 16;
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
            if constexpr (SFT) { if (gi > 0) pack_ss(gi - 1); }     // behind the next group's MFMAs: its results have landed
        }
        if constexpr (SFT) pack_ss(N - 1);
#pragma unroll
        for (int gi = 0; gi < N; ++gi) {
            i32x4 codes;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                f16x4 y = yv[gi][qd];
                if constexpr (SFT) {
                    y = y * s1p[gi][qd] + s0p[gi][qd];
                }
                if constexpr (I8) {
                    const unsigned w = quant4((float)y[0], (float)y[1], (float)y[2], (float)y[3], p.q_inv, p.q_zoff);
                    codes[qd] = inimg[gi] ? (int)w : 0;
                } else {
                    if (!inimg[gi]) { y[0] = (f16)0.f; y[1] = (f16)0.f; y[2] = (f16)0.f; y[3] = (f16)0.f; }
                    yv[gi][qd] = y;
                }
            }
            if (g_pos[G0 + gi] >= 0) {
                if constexpr (I8) {
                    *reinterpret_cast<i32x4 *>(qbuf + g_q[G0 + gi]) = codes;
                } else {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<f16x4 *>(a + (g_x[G0 + gi] ^ (qd << 4))) = yv[gi][qd];
                }
            }
        }
    };
    auto prep_tile = [&](int tt, char *a, const char *cbuf, char *qbuf, const f16x4 *p3) __attribute__((always_inline)) {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if constexpr (SPLIT) {
            if (wave >= 4) {          // three groups as 2 + 1 (W8A8 layers, conv_first fused: one by one): more at once do not fit the register file
                if constexpr (SQ || C3 || I8) {
                    sft_groups(I1{}, I0{}, tt, a, cbuf, qbuf, p3);
                    sft_groups(I1{}, I1{}, tt, a, cbuf, qbuf, p3);
                    sft_groups(I1{}, I2{}, tt, a, cbuf, qbuf, p3);
                } else {
                    sft_groups(I2{}, I0{}, tt, a, cbuf, qbuf, p3);
                    sft_groups(I1{}, I2{}, tt, a, cbuf, qbuf, p3);
                }
            }
        } else {
            if (wave >= 4) sft_groups(I2{}, I0{}, tt, a, cbuf, qbuf, p3);  // wave-uniform
            else sft_groups(I1{}, I0{}, tt, a, cbuf, qbuf, p3);
        }
    };

    // ---- prologue: tiles 0 and 1 in flight, tile 0 landed and transformed
    const int t0 = blockIdx.x, step = gridDim.x;
    issue_tile(t0, sA, sC);
    issue_tile(t0 + step, sA + A_BYTES, sC + C_BYTES);
    if constexpr (C3) {
        p3_fetch(t0); p3_stage(0);
        p3_fetch(t0 + step); p3_stage(1);
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
    __builtin_amdgcn_s_barrier();
    if (PREP) {
        prep_tile(t0, sA, sC, sQ, sP3);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));      // LDS writes of the pass done; vmcnt untouched
        __builtin_amdgcn_s_barrier();
    }

    // ---- conv fragment addresses (lane constants)
    // this lane's output pixels in the tile: q = (conv wave) * 32 NQ + 32 g + l31, i.e. rows 2 NQ w + 2 g + (l31 >> 4)
    int qy[NQ], xoff[NQ][3], qoff[NQ][3];
    const int qx = l31 % TW;
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
        qy[g] = (wave * NQ + g) * 2 + l31 / TW;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            xoff[g][kx] = (qy[g] * HW + qx + kx) * 64 + ((lh ^ swz32(qx + kx)) << 4);
            qoff[g][kx] = (qy[g] * HW + qx + kx) * 32 + ((lh ^ (qy[g] & 1)) << 4);
        }
    }
    const int woff = I8 ? l31 * 32 + ((lh ^ ((l31 >> 3) & 1)) << 4) : l31 * 64 + ((lh ^ swz32(l31)) << 4);
    // this lane's 2 NQ 16-byte output chunks: pixel (2 NQ wave + it, lane >> 2) of the tile, channel chunk c8
    const int c8 = lane & 3, spx = lane >> 2;
    const float aslope = act_slope(p.act);
    char *trash = reinterpret_cast<char *>(p.trash) + tid * 16;

    int ab = 0, cb = 0;                                  // conv reads A[ab] (Q[cb]); P works on the next of each; the DMA lands behind that
    STAMP_DECL;
    for (int t = t0; t < ntiles; t += step) {
        STAMP(7);
        const int nb = ab + 1 == NA ? 0 : ab + 1, db = NA == 2 ? ab : (nb + 1 == NA ? 0 : nb + 1);
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const bool conv_wave = !SPLIT || wave < 4;           // wave-uniform
        int bcls[NQ];
#pragma unroll
        for (int g = 0; g < NQ; ++g)
            bcls[g] = I8 ? ((((oy0 + qy[g] == 0) | ((oy0 + qy[g] == p.H - 1) << 1)) << 2) | ((ox0 + qx == 0) | ((ox0 + qx == p.W - 1) << 1))) & 15 : 0;

        // R: destination addresses and residuals (masked lanes: trash line / zero line), loads issued unconditionally
        f16 *dptr[2 * NQ];
        i32x4 rs1[2 * NQ], rs2[2 * NQ];                  // raw bits of 8 f16 each
        f16 *pl_dst[NQ][3];
        f16 pl_res[NQ][3];
        if (conv_wave) {
            if constexpr (PLANAR) {
#pragma unroll
                for (int g = 0; g < NQ; ++g) {
                    const int oy = oy0 + qy[g], ox = ox0 + qx;
                    const bool in = lh == 0 && oy < p.H && ox < p.W;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const size_t e = (size_t)ch * p.H * p.W + (size_t)oy * p.W + ox;
                        pl_dst[g][ch] = in ? p.dst_planar + e : reinterpret_cast<f16 *>(trash);
                        pl_res[g][ch] = *((in && p.res_planar) ? p.res_planar + e : p.zeros);
                    }
                }
            } else {
#pragma unroll
                for (int it = 0; it < 2 * NQ; ++it) {
                    const int oy = oy0 + 2 * NQ * wave + it, ox = ox0 + spx;
                    const bool in = c8 * 8 < p.Cout && oy < p.H && ox < p.W;
                    const size_t off = ((size_t)oy * p.W + ox) * p.dstC + c8 * 8;
                    dptr[it] = in ? p.dst + off : reinterpret_cast<f16 *>(trash);
                    rs1[it] = *reinterpret_cast<const i32x4 *>((in && p.res1) ? p.res1 + off : p.zeros);
                    rs2[it] = *reinterpret_cast<const i32x4 *>((in && p.res2) ? p.res2 + off : p.zeros);
                }
            }
        }
        if constexpr (C3) p3_fetch(t + 2 * step);              // R: this thread's pixel of the patch of tile t+2
        // X: tile t+2 in flight
        issue_tile(t + 2 * step, sA + db * A_BYTES, sC + cb * C_BYTES);
        __builtin_amdgcn_sched_barrier(0);
        STAMP(0);      // addresses, residual loads, DMA issue

        f32x16 acc[NQ];
        // the wave's NQ pixel groups share every weight fragment: 1 + NQ LDS reads per NQ MFMAs
        auto conv_mfma = [&]() __attribute__((always_inline)) {
            if constexpr (I8) {
                const char *qa = sQ + cb * L::Q_BYTES;
                i32x16 iacc[NQ];
#pragma unroll
                for (int g = 0; g < NQ; ++g)
#pragma unroll
                    for (int k = 0; k < 16; ++k) iacc[g][k] = 0;
                i32x4 wq[9], xq[NQ][9];
                auto ldq = [&](int tap) {
                    wq[tap] = *reinterpret_cast<const i32x4 *>(sW + woff + tap * 32 * 32);
#pragma unroll
                    for (int g = 0; g < NQ; ++g)
                        xq[g][tap] = *reinterpret_cast<const i32x4 *>(qa + ((qoff[g][tap % 3] + (tap / 3) * HW * 32) ^ (((tap / 3) & 1) << 4)));
                };
                ldq(0); ldq(1); ldq(2);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if (tap + 3 < 9) ldq(tap + 3);
#pragma unroll
                    for (int g = 0; g < NQ; ++g) iacc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wq[tap], xq[g][tap], iacc[g], 0, 0, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 3 * (1 + NQ), 0);
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NQ, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1 + NQ, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NQ, 0);
#pragma unroll
                for (int g = 0; g < NQ; ++g)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[g][k] = (float)iacc[g][k];
            } else {
                const char *a = sA + ab * A_BYTES;
#pragma unroll
                for (int g = 0; g < NQ; ++g)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[g][k] = 0.f;
                f16x8 wfr[18], xfr[NQ][18];
                auto ldfrag = [&](int st) {
                    const int tap = st >> 1, ks = st & 1;
                    wfr[st] = *reinterpret_cast<const f16x8 *>(sW + (woff ^ (ks << 5)) + tap * 32 * 64);
#pragma unroll
                    for (int g = 0; g < NQ; ++g)
                        xfr[g][st] = *reinterpret_cast<const f16x8 *>(a + (xoff[g][tap % 3] ^ (ks << 5)) + (tap / 3) * HW * 64);
                };
                ldfrag(0); ldfrag(1); ldfrag(2);
#pragma unroll
                for (int st = 0; st < 18; ++st) {
                    if (st + 3 < 18) ldfrag(st + 3);
#pragma unroll
                    for (int g = 0; g < NQ; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wfr[st], xfr[g][st], acc[g], 0, 0, 0);
                }
                // pin the interleave: three steps of reads up front, then {NQ MFMAs, 1 + NQ reads} x 15, then 3 NQ MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 3 * (1 + NQ), 0);
#pragma unroll
                for (int st = 0; st < 15; ++st) {
                    __builtin_amdgcn_sched_group_barrier(0x008, NQ, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1 + NQ, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NQ, 0);
            }
        };
        // Epilogue, last thing in the tile on both phase orders.  hipcc waits vmcnt(0) in front of the residuals' first use
        // (it does not count across LDS-DMA) and in front of the first LDS read it can name while a DMA is in flight (the
        // scale / shift tables): both land HERE, a few instructions in front of the tile's closing wait for the same DMA,
        // instead of in the middle of the MFMA or SFT phase.
        auto epilogue = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PLANAR) {
                // conv_last: channels 0..2 of pixel l31 sit in registers 0..2 of the lanes with lh == 0; the conv result
                // is rounded to f16, the residual added in fp32 and the sum rounded again (as the staged path did)
#pragma unroll
                for (int g = 0; g < NQ; ++g) {
                    const float4 sc4 = *reinterpret_cast<const float4 *>(sSS), sh4 = *reinterpret_cast<const float4 *>(sSS + 32 + (I8 ? bcls[g] * 32 : 0));
                    const float sc[3] = {sc4.x, sc4.y, sc4.z}, sh[3] = {sh4.x, sh4.y, sh4.z};
                    const float o[3] = {act_fast(acc[g][0] * sc[0] + sh[0], aslope), act_fast(acc[g][1] * sc[1] + sh[1], aslope),
                                        act_fast(acc[g][2] * sc[2] + sh[2], aslope)};
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        const float v = (float)(f16)o[ch] + (p.res_planar ? (float)pl_res[g][ch] : 0.f);
                        *pl_dst[g][ch] = (f16)v;
                    }
                }
            } else {
#pragma unroll
                for (int g = 0; g < NQ; ++g)
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const int cl = 8 * qd + 4 * lh;
                        const float4 s4 = *reinterpret_cast<const float4 *>(sSS + cl);
                        const float4 h4 = *reinterpret_cast<const float4 *>(sSS + 32 + (I8 ? bcls[g] * 32 : 0) + cl);
                        const float sc[4] = {s4.x, s4.y, s4.z, s4.w}, sh[4] = {h4.x, h4.y, h4.z, h4.w};
                        f16x4 o;
                        o[0] = (f16)act_fast(acc[g][4 * qd + 0] * sc[0] + sh[0], aslope);
                        o[1] = (f16)act_fast(acc[g][4 * qd + 1] * sc[1] + sh[1], aslope);
                        o[2] = (f16)act_fast(acc[g][4 * qd + 2] * sc[2] + sh[2], aslope);
                        o[3] = (f16)act_fast(acc[g][4 * qd + 3] * sc[3] + sh[3], aslope);
                        *reinterpret_cast<f16x4 *>(strip + (g * 32 + l31) * OUT_ROWB + cl * 2) = o;
                    }
                // this wave's 32 NQ pixels back as 16-byte channel chunks (LDS operations of one wave complete in order)
                f16x8 v[2 * NQ];
#pragma unroll
                for (int it = 0; it < 2 * NQ; ++it) v[it] = *reinterpret_cast<const f16x8 *>(strip + (it * 16 + spx) * OUT_ROWB + c8 * 16);
#pragma unroll
                for (int it = 0; it < 2 * NQ; ++it) {
                    // residual adds in packed f16, one rounding per add as the reference's fp16 model does
                    // (x + conv2(..) then + skip, arch_util.py:95, HDRUNet3T1_arch.py:186-198)
                    v[it] = (v[it] + __builtin_bit_cast(f16x8, rs1[it])) + __builtin_bit_cast(f16x8, rs2[it]);
                    *reinterpret_cast<f16x8 *>(dptr[it]) = v[it];
                }
            }
        };
        const int cn = cb ^ 1;
        if constexpr (SPLIT) {
            if (wave < 4) {                                    // conv wave: 64 output pixels
                conv_mfma();
                STAMP(1);
                epilogue();
                STAMP(3);
            } else {                                           // prep wave: three halo groups of the next tile
                if (t + step < ntiles) prep_tile(t + step, sA + nb * A_BYTES, sC + cn * C_BYTES, sQ + cn * L::Q_BYTES, sP3 + cn * (P3_BYTES / 8));
                STAMP(2);
            }
            if constexpr (C3) p3_stage(cb);
            // prep waves: their DMA of tile t+2 has landed and their LDS writes are done; conv waves: their LDS reads are done
            // (and with conv_first fused their condition-tile pieces have landed: older than the NSTORE stores, which are
            // never waited for)
            if (wave < 4) __builtin_amdgcn_s_waitcnt(waitcnt_imm(C3 ? NSTORE : 63, 0));     // C3: their condition-tile pieces are older than the stores
            else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        } else {
            if (wave < 4) {
                conv_mfma();
                STAMP(1);  // conv MFMAs
                if (PREP && t + step < ntiles) prep_tile(t + step, sA + nb * A_BYTES, sC + cn * C_BYTES, sQ + cn * L::Q_BYTES, sP3 + cn * (P3_BYTES / 8));
                STAMP(2);  // SFT / quantise pass of the next tile
            } else {
                if (PREP && t + step < ntiles) prep_tile(t + step, sA + nb * A_BYTES, sC + cn * C_BYTES, sQ + cn * L::Q_BYTES, sP3 + cn * (P3_BYTES / 8));
                STAMP(2);
                conv_mfma();
                STAMP(1);
            }
            epilogue();
            if constexpr (C3) p3_stage(cb);                    // patch of tile t+2 (its parity is this tile's)
            STAMP(3);      // epilogue incl. hipcc's vmcnt(0) (residuals + the DMA of tile t+2)
            // the DMA of tile t+2 is older than this tile's NSTORE stores: landed once at most NSTORE operations are pending
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(NSTORE, 0));
        }
        STAMP(4);      // closing wait
        __builtin_amdgcn_s_barrier();
        STAMP(5);      // barrier
        ab = nb;
        cb = cn;
    }
#ifdef HDRTV_STAMP
    if (p.dump && lane == 0)
        for (int i = 0; i < 8; ++i) reinterpret_cast<unsigned long long *>(p.dump)[((size_t)blockIdx.x * NW + wave) * 8 + i] = st_acc[i];
#endif
}

template <bool SFT, bool I8, bool SQ, bool PLANAR, bool C3 = false, bool SPLIT = false>
hipError_t launch_k(const Conv32Params &p, int n_cu, hipStream_t s)
{
    using L = Lay<SFT, I8, SQ, C3>;
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv32s_kernel<SFT, I8, SQ, PLANAR, C3, SPLIT>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, L::SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const int ntiles = p.tiles_x * p.tiles_y;
    const int grid = ntiles < n_cu ? ntiles : n_cu;        // persistent: one workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), L::SMEM, s, p);
    return hipGetLastError();
}

// layers with a P pass run the role split (waves 0-3 convolve, waves 4-7 prepare the next tile) unless the caller's A/B switch
// `nosplit` asks for the first form
template <bool SFT, bool I8, bool SQ, bool PLANAR, bool C3 = false>
hipError_t launch_t(const Conv32Params &p, int n_cu, hipStream_t s, bool nosplit)
{
    if constexpr ((SFT || I8) && !SQ) {        // W8A8 SFT convs: their P pass is too heavy for four waves (1.52 -> 1.82 ms split)
#ifdef HDRTV_AB
        if (nosplit) return launch_k<SFT, I8, SQ, PLANAR, C3, false>(p, n_cu, s);
#else
        if (nosplit) return hipErrorNotSupported;              // the first form exists in the A/B library only (make AB=1)
#endif
        return launch_k<SFT, I8, SQ, PLANAR, C3, true>(p, n_cu, s);
    } else {
        return launch_k<SFT, I8, SQ, PLANAR, C3, false>(p, n_cu, s);
    }
}

}  // namespace

// Single-pass (CoutPad == 32) layers only; src (and cond) must be followed by >= 64 zero bytes (the workspace guard),
// p.zeros must hold >= 16 zero bytes and p.trash >= 8 KiB of write-only scratch.
hipError_t conv32s_launch(Conv32Params p, int n_cu, hipStream_t s, bool nosplit)
{
    if ((size_t)p.H * p.W * 64 >= 0xf0000000ull || p.CoutPad != 32 || !p.zeros || !p.trash) return hipErrorInvalidValue;
    const bool sft = p.cond != nullptr, planar = p.mode == ST_PLANAR3;
    if (!planar && (p.mode != ST_NHWC || p.dstC < 32)) return hipErrorInvalidValue;
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    if (p.wpk8) {
        if (p.c3_img) return hipErrorInvalidValue;
        if (planar) return sft ? hipErrorInvalidValue : launch_t<false, true, false, true>(p, n_cu, s, nosplit);
        if (sft && p.sq_wfrag) return launch_t<true, true, true, false>(p, n_cu, s, nosplit);
        return sft ? launch_t<true, true, false, false>(p, n_cu, s, nosplit) : launch_t<false, true, false, false>(p, n_cu, s, nosplit);
    }
    if (p.sq_wfrag) return hipErrorInvalidValue;         // W8A8 SFT convs in front of an fp16 conv: no kernel (no shipped recipe has it)
    if (planar) return sft ? hipErrorInvalidValue : launch_t<false, false, false, true>(p, n_cu, s, nosplit);
    if (p.c3_img) return (sft && p.c3_wfrag) ? launch_t<true, false, false, false, true>(p, n_cu, s, nosplit) : hipErrorInvalidValue;
    return sft ? launch_t<true, false, false, false>(p, n_cu, s, nosplit) : launch_t<false, false, false, false>(p, n_cu, s, nosplit);
}
