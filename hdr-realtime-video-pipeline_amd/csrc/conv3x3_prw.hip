// conv3x3_prw.hip -- HG 3x3 convolutions (Hallucination_arch.py:97-137), "private weights" schedule.
//
// GEMM view as conv3x3_pglds.hip (M = output channels, N = a 16x16-pixel tile, K = 64-channel chunk x tap, halo tile
// with its 18x18 pixels x 128 B staged once per chunk by LDS-DMA and re-read for all nine taps), but the work of a tile
// is cut the other way: the block covers 256 output channels and wave w owns channels 32w .. 32w+31 for ALL 256 pixels
// (2 x 16 accumulator tiles of 16x16 = 128 VGPRs).  A wave therefore needs only ITS 32 weight rows of a tap (4 KiB),
// which it DMAs into a wave-private two-slot ring one tap ahead and waits for with its own vmcnt:
//   * no weight traffic is shared between waves, so there is no barrier per tap -- one s_barrier per 64-channel chunk
//     (576 MFMAs per wave) orders the shared halo buffers, where conv_pglds has nine;
//   * a wave's waits are for DMAs it issued a whole tap (~2000 cycles) earlier, so they are free in steady state, and the
//     eight waves drift apart instead of reaching the DMA issue / wait / barrier / MFMA phases together: the two waves
//     of a SIMD fill each other's gaps on the matrix pipe;
//   * per MFMA: 0.56 ds_read_b128 (conv_pglds: 0.5), 0.06 LDS-DMA pieces (0.08), the same L2 -> CU weight bytes.
// Accumulation order per output element is conv_pglds's (chunk, tap, k-step), so results are bit-identical to it.
//
// LDS: 2 x 41 KiB halo + 8 x 2 x 4 KiB weight rings + 2 x 2 KiB scale/shift + 1 KiB DMA trash + 1 KiB dot weights = 152 KiB.
#include "launchers.h"

namespace {

constexpr int TW = 16, HW = 18;
constexpr int CT = 64, PIXB = CT * 2;                    // 64-channel chunk = 128 B per pixel
constexpr int BN = 256, WCH = 32;                        // block / wave output channels
// Tile height TH = 16 (324 halo px = 40.5 KiB: 41 pieces, 6 per wave, pieces 41..47 go to the trash KiB) or, for layers with
// too few 16x16 tiles to fill the chip's last round, TH = 8 (180 halo px: 23 pieces, 3 per wave; twice the weight bytes per MAC)
template <int TH> struct Geo {
    static constexpr int NPIX = (TH + 2) * HW, A_PIECES = (NPIX * PIXB + 1023) / 1024, A_PIECES_PER_WAVE = (A_PIECES + 7) / 8;
};
constexpr int A_BYTES = Geo<16>::A_PIECES * 1024;        // LDS layout is the TH = 16 one for both
constexpr int W_SLOT = WCH * PIXB;                       // 4 KiB: 32 rows x 64 K
constexpr int W_OFF = 2 * A_BYTES;
constexpr int SS_OFF = W_OFF + 8 * 2 * W_SLOT;           // two slots of {scale[256], shift[256]}
constexpr int TRASH_OFF = SS_OFF + 2 * 2048;
constexpr int DOTW_OFF = TRASH_OFF + 1024;
constexpr int SMEM = DOTW_OFF + 1024;                    // 155 648 B

// LDS-DMA as a BUFFER load (buffer_load_dwordx4 ... lds), not global_load_lds: the global form is a FLAT-encoded
// instruction that hipcc's waitcnt pass treats as "may touch LDS and memory", after which it never counts again -- every
// later wait becomes lgkmcnt(0) / vmcnt(0) (tools/lds_dma_oob_probe.hip and the ISA of this file show the difference).
// Lanes whose byte offset lies outside the resource's num_records write zeros to LDS: the image border needs no zero line.
// OFF is an immediate added to both the memory and the LDS address.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)bytes, 0x00020000);
}
template <int OFF> __device__ __forceinline__ void bdma16(rsrc_t r, void *lds, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, OFF, 0);
}
constexpr unsigned OOB = 0x80000000u;                    // beyond any tensor here (all < 2 GiB)

template <int N> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Tile { int n0, oy0, ox0; };

template <int MODE, int TH>
__global__ __launch_bounds__(512) void conv_prw_kernel(ConvParams p)
{
    constexpr int NPIX = Geo<TH>::NPIX, A_PIECES = Geo<TH>::A_PIECES, A_PIECES_PER_WAVE = Geo<TH>::A_PIECES_PER_WAVE;
    constexpr int NG = TH / 2;                               // groups of two pixel rows per tap
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char *sW = smem + W_OFF + wave * (2 * W_SLOT);

    // ---- this block's run of tiles: XCD x owns a contiguous range, its blocks interleave in it (as conv_pglds) --
    const int ntn = p.CoutPad / BN;
    const int total = p.tiles_x * p.tiles_y * ntn;
    int t_first, t_step, ntile;
    {
        const int G = gridDim.x, b = blockIdx.x, xcd = b & 7, slot = b >> 3;
        const int nslots = (G - xcd + 7) >> 3;
        const int q = total >> 3, r = total & 7;
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int len = q + (xcd < r ? 1 : 0);
        t_first = base + slot;
        t_step = nslots;
        ntile = slot < len ? (len - slot + nslots - 1) / nslots : 0;
    }
    if (ntile == 0) return;
    auto decode = [&](int t) {
        Tile o;
        const int nsp = p.tiles_x * p.tiles_y;
        const int nt_i = p.nt_slow ? t / nsp : t % ntn, sp = p.nt_slow ? t - nt_i * nsp : t / ntn;
        const int ty = sp / p.tiles_x, tx = sp - ty * p.tiles_x;
        o.n0 = nt_i * BN; o.oy0 = ty * TH; o.ox0 = tx * TW;
        return o;
    };

    const int nchunk = (p.c0 + p.c1) / CT, nchunk0 = p.c0 / CT;

    // ---- LDS-DMA issue helpers (wave-uniform LDS base, per-lane swizzled source) ------------
    // (the per-lane address arithmetic of the rare DMA issues and of the epilogue is recomputed from an opaque copy of the
    // lane id each time: hoisted out of the tile loop it would sit in ~40 VGPRs across the MFMA stream, which has none to spare)
    auto opaque_lane = [&]() { int v = lane; asm volatile("" : "+v"(v)); return v; };
    auto issue_A = [&](int cc, int buf, const Tile &T) {
        const int ln = opaque_lane();
        const int l_row = ln >> 3, l_slot = ln & 7;
        const f16 *src;
        int cs, coff;
        if (cc < nchunk0) { src = p.src0; cs = p.s0_stride; coff = cc * CT; }
        else { src = p.src1; cs = p.s1_stride; coff = (cc - nchunk0) * CT; }
        const rsrc_t rs = make_rsrc(src, (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)cs * 2u);
#pragma unroll
        for (int it = 0; it < A_PIECES_PER_WAVE; ++it) {
            const int piece = wave + it * 8;
            const int hp = piece * 8 + l_row;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = T.oy0 - 1 + hy, ix = T.ox0 - 1 + hx;
            const bool ok = (hp < NPIX) & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
            const unsigned off = ((unsigned)(iy * p.Wi + ix) * (unsigned)cs + (unsigned)(coff + ((l_slot ^ (hx & 7)) << 3))) * 2u;
            bdma16<0>(rs, piece < A_PIECES ? sA + buf * A_BYTES + piece * 1024 : smem + TRASH_OFF, ok ? off : OOB, 0);
        }
    };
    // this wave's 32 weight rows of (chunk cc, tap): four 1-KiB pieces = one scalar offset + four immediates
    const unsigned w_lane = (unsigned)((lane >> 3) * CT + (((lane & 7) ^ (lane >> 3)) << 3)) * 2u;
    auto issue_W = [&](int cc, int tap, int n0, int slot) {
        const rsrc_t rs = make_rsrc(p.wpk, 9u * (unsigned)nchunk * (unsigned)p.CoutPad * (unsigned)PIXB);
        const unsigned so = (unsigned)((tap * nchunk + cc) * p.CoutPad + n0 + wave * WCH) * (unsigned)PIXB;
        char *d = sW + slot * W_SLOT;
        bdma16<0>(rs, d, w_lane, so); bdma16<1024>(rs, d, w_lane, so); bdma16<2048>(rs, d, w_lane, so); bdma16<3072>(rs, d, w_lane, so);
    };
    auto issue_SS = [&](int n0, int slot) {      // wave 0: scale[256], wave 1: shift[256]; the others keep the piece count equal
        const rsrc_t rs = make_rsrc(wave == 1 ? p.shift : p.scale, (unsigned)p.CoutPad * 4u);
        bdma16<0>(rs, wave < 2 ? smem + SS_OFF + slot * 2048 + wave * 1024 : smem + TRASH_OFF, (unsigned)opaque_lane() * 16u, (unsigned)n0 * 4u);
    };

    // ST_PS_DOT3 (Up_conv5: Cout = 256 = four pixel-shuffle positions x 64 channels): the 64 -> 3 dot products behind the
    // shuffle run over TWO waves' channels (wave 2s: channels 0-31 of position s, wave 2s+1: 32-63).  The odd wave leaves its
    // partial sums in its strip and raises flag[s] = tile count; the even wave adds them to its own and stores; ack[s] tells
    // the odd wave that the strip may be overwritten (by its weight DMA at tap 1 of the next tile).  Both counters live in LDS.
    volatile int *s_flag = reinterpret_cast<volatile int *>(smem + DOTW_OFF + 768);     // [4] flag, [4] ack
    if constexpr (MODE == ST_PS_DOT3) {
        // The dot products run on the matrix pipe (epilogue below).  Table entry ((half * 3 + o) * 4 + kg): the A-operand
        // fragment of output o for the lanes of k-group kg of a wave that owns channel half `half` -- K slot e of the group is
        // channel 4 kg + e (e < 4, accumulator block 0) or 16 + 4 kg + e - 4 (block 1), i.e. exactly the order in which a
        // lane's two f16x4 results of a pixel row form a B fragment without any data movement.  An fp32 weight is carried as
        // hi + lo * 2^-10 (two f16 values, lo scaled into the normal range): products exact, 22 bits of the weight kept.
        if (tid < 24) {
            const int kgq = tid & 3, o = (tid >> 2) % 3, half = tid / 12;
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float w = p.dotw[o * 64 + half * 32 + (e < 4 ? 4 * kgq + e : 16 + 4 * kgq + e - 4)];
                hi[e] = (f16)w;
                lo[e] = (f16)((w - (float)hi[e]) * 1024.f);
            }
            *reinterpret_cast<f16x8 *>(smem + DOTW_OFF + tid * 32) = hi;
            *reinterpret_cast<f16x8 *>(smem + DOTW_OFF + tid * 32 + 16) = lo;
        }
        if (tid < 8) s_flag[tid] = 0;
    }

    f32x4 acc[2][TH];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TH; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float act_lb = p.act == ACT_RELU ? 0.f : -__builtin_inff();

    // ---- the MFMA stream -------------------------------------------------------------------------
    // A tap is TH/2 groups of two pixel rows: group g's four fragment reads are issued in front of group g-1's eight MFMAs
    // (the compiler's counted lgkmcnt waits leave them in flight).  The pipeline runs across taps, chunks and tiles: group 7 of a tap reads the next
    // tap's weight fragments and its first group, behind the vmcnt wait for those weights (issued at the top of this tap)
    // and, in front of a new chunk, the barrier that says every wave's halo pieces have landed.
    f16x8 wf[2][2], xa[2][2], xb[2][2];
    // (fragment addresses are rebuilt from an opaque lane id at every tap: kept loop-invariant they cost a dozen VGPRs)
    auto rd_w = [&](f16x8 (&w)[2][2], int slot, int ln) {
        const int q15 = ln & 15, qg = ln >> 4;
        const char *bw = sW + slot * W_SLOT + q15 * PIXB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                w[ks][i] = *reinterpret_cast<const f16x8 *>(bw + i * 16 * PIXB + (((ks * 4 + qg) ^ (q15 & 7)) << 4));
    };
    // ax: the tap's halo origin (buffer + tap offset, wave-uniform); dx = tap % 3
    auto rd_x = [&](f16x8 (&x)[2][2], const char *ax, int dx, int row0, int ln) {
        const int q15 = ln & 15, qg = ln >> 4, kx = (q15 + dx) & 7;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                x[r][ks] = *reinterpret_cast<const f16x8 *>(ax + q15 * PIXB + (row0 + r) * HW * PIXB + (((ks * 4 + qg) ^ kx) << 4));
    };
    auto mm = [&](const f16x8 (&x)[2][2], int row0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = row0 + r;
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][0], x[r][0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][1], x[r][0], acc[1][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][0], x[r][1], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][1], x[r][1], acc[1][j], 0, 0, 0);
        }
    };

    // ---- prologue ------------------------------------------------------------------------------
    Tile cur = decode(t_first), nxt = cur;
    issue_A(0, 0, cur);
    issue_SS(cur.n0, 0);
    issue_W(0, 0, cur.n0, 0);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    rd_w(wf, 0, lane);
    rd_x(xa, sA, 0, 0, lane);

    int gch = 0;                                  // chunks done so far: halo buffer parity
    int ws = 0;                                   // taps done so far: weight slot parity
    for (int k = 0; k < ntile; ++k) {
        const bool has_next = k + 1 < ntile;
        if (has_next) nxt = decode(t_first + (k + 1) * t_step);
        for (int cc = 0; cc < nchunk; ++cc, ++gch) {
            const char *a = sA + (gch & 1) * A_BYTES;
            const char *a_nc = sA + ((gch + 1) & 1) * A_BYTES;
            const bool last_chunk = cc + 1 == nchunk;
            const bool pfA = !last_chunk || has_next;    // a halo tile is staged during this chunk's tap 1
#pragma unroll
            for (int tap = 0; tap < 9; ++tap, ++ws) {
                if constexpr (MODE == ST_PS_DOT3) {
                    // this tap's DMA goes into the slot that was the last tile's strip: the partner must have read it
                    if (tap == 1 && cc == 0 && k > 0 && (wave & 1))
                    {
                        while (s_flag[4 + (wave >> 1)] < k) __builtin_amdgcn_s_sleep(2);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // nothing below (the DMA into the strip) moves above the poll
                    }
                }
                // weights of the next tap into the slot whose fragments have been in registers since the end of the last tap
                if (tap < 8) issue_W(cc, tap + 1, cur.n0, (ws + 1) & 1);
                else if (!last_chunk) issue_W(cc + 1, 0, cur.n0, (ws + 1) & 1);
                else if (has_next) issue_W(0, 0, nxt.n0, (ws + 1) & 1);
                if (tap == 1 && pfA) {
                    if (!last_chunk) issue_A(cc + 1, (gch + 1) & 1, cur);
                    else { issue_A(0, (gch + 1) & 1, nxt); issue_SS(nxt.n0, (k + 1) & 1); }
                }
                const char *ax = a + ((tap / 3) * HW + tap % 3) * PIXB;
                const int tl = opaque_lane();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NG - 1; ++g) {
                    if (g & 1) rd_x(xa, ax, tap % 3, 2 * g + 2, tl); else rd_x(xb, ax, tap % 3, 2 * g + 2, tl);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g & 1) mm(xb, 2 * g); else mm(xa, 2 * g);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the next tap's weights, issued at the top of this one, have had seven groups to land; only this tap's halo
                // pieces are younger.  vmcnt retires in issue order: a halo staged at tap 1 has landed long before tap 8.
                if (tap == 1 && pfA) {
                    if (last_chunk) wait_vm<A_PIECES_PER_WAVE + 1>(); else wait_vm<A_PIECES_PER_WAVE>();
                } else {
                    wait_vm<0>();
                }
                if (tap == 8) __builtin_amdgcn_s_barrier();   // every wave's pieces of the next halo are in; this chunk's buffer is done with
                {
                    const int tnx = (tap + 1) % 9;
                    const char *axn = (tap == 8 ? a_nc : a) + ((tnx / 3) * HW + tnx % 3) * PIXB;
                    f16x8 wn[2][2];
                    rd_x(xa, axn, tnx % 3, 0, tl);
                    rd_w(wn, (ws + 1) & 1, tl);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(xb, TH - 2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int i = 0; i < 2; ++i) wf[ks][i] = wn[ks][i];
                }
            }
        }

        // ------------------------------------------------------------ epilogue, from registers
        // lane: pixel (row j, column l15), channels n0 + 32 wave + i*16 + 4*kg + {0..3}
        const float *ss = reinterpret_cast<const float *>(smem + SS_OFF + (k & 1) * 2048);
        const int eln = opaque_lane();
        const int l15 = eln & 15, kg = eln >> 4;
        char *trash = reinterpret_cast<char *>(p.trash) + eln * 16;
        const int cw = wave * WCH + 4 * kg;
        // The strip is the weight slot whose fragments (the next tile's first tap) are already in registers; the next DMA into
        // it is a whole tap away (the other slot receives weights at the top of the next tap): wave-private, no synchronisation.  A pixel's 32 channels are 64 B = four 16-B slots, xor-swizzled by
        // pixel so that the 8-byte quad writes of 16 pixels spread over the banks.
        char *stg = sW + (ws & 1) * W_SLOT;
        const int s_px = eln >> 2, s_slot = eln & 3;
        float4 sc[2], sh[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sc[i] = *reinterpret_cast<const float4 *>(ss + cw + i * 16);
            sh[i] = *reinterpret_cast<const float4 *>(ss + 256 + cw + i * 16);
        }
        auto outv = [&](int i, int j) {
            f16x4 o;
            o[0] = (f16)fmaxf(acc[i][j][0] * sc[i].x + sh[i].x, act_lb);
            o[1] = (f16)fmaxf(acc[i][j][1] * sc[i].y + sh[i].y, act_lb);
            o[2] = (f16)fmaxf(acc[i][j][2] * sc[i].z + sh[i].z, act_lb);
            o[3] = (f16)fmaxf(acc[i][j][3] * sc[i].w + sh[i].w, act_lb);
            acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            return o;
        };
        auto st_off = [&](int px, int slot16, int half) {       // byte offset of an 8-byte quad in the strip
            return px * 64 + ((slot16 ^ ((px >> 1) & 3)) << 4) + half * 8;
        };
        if constexpr (MODE == ST_NHWC || MODE == ST_PS) {
            const int cps = p.dstC;
            const int chw = cur.n0 + wave * WCH;
            const int sub = MODE == ST_PS ? chw / cps : 0;
            const int cbase = (MODE == ST_PS ? chw - sub * cps : chw) + s_slot * 8;
#pragma unroll
            for (int pass = 0; pass < TH / 4; ++pass) {         // four pixel rows per pass = the 4 KiB strip
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        *reinterpret_cast<f16x4 *>(stg + st_off(jj * 16 + l15, i * 2 + (kg >> 1), kg & 1)) = outv(i, pass * 4 + jj);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int px = rr * 16 + s_px;
                    const f16x8 v = *reinterpret_cast<const f16x8 *>(stg + px * 64 + ((s_slot ^ ((px >> 1) & 3)) << 4));
                    const int oy = cur.oy0 + pass * 4 + rr;
                    const int oxx = cur.ox0 + s_px;
                    f16 *d;
                    if constexpr (MODE == ST_NHWC) {
                        const bool ok = oy < p.Ho && oxx < p.Wo;
                        d = ok ? p.dst + ((size_t)oy * p.Wo + oxx) * p.dstC + cbase : reinterpret_cast<f16 *>(trash);
                    } else {
                        const int Y = 2 * oy + (sub >> 1), X = 2 * oxx + (sub & 1);
                        const bool ok = oy < p.Ho && oxx < p.Wo && Y < p.Hd && X < p.Wd;
                        d = ok ? p.dst + ((size_t)Y * p.Wd + X) * cps + cbase : reinterpret_cast<f16 *>(trash);
                    }
                    *reinterpret_cast<f16x8 *>(d) = v;
                }
            }
        } else if constexpr (MODE == ST_PS_DOT3) {
            // pixel shuffle, ReLU, then this wave's half of the 64 -> 3 dot products: only 3 partial sums per pixel leave the CU.
            // The f16 results of a pixel row (as the reference's fp16 graph holds Up_conv5's output) are the B operand of one
            // v_mfma_f32_16x16x32_f16 per weight half (hi, lo): K = the wave's 32 channels, output rows 4 q + o carry output o of
            // pixel row 4 pass + q, so after four rows lane (kg, l15) holds the three sums of pixel (4 pass + kg, l15) in
            // registers 0..2 -- the layout the pair hand-off below expects.  fp32 accumulation as conv_pglds's, associated
            // differently (and the weights to 22 bits): the results agree with it to rounding, not bit for bit.
            const int orow = l15 & 3, qrow = l15 >> 2;
            const char *tb = smem + DOTW_OFF + ((((wave & 1) * 3 + (orow < 3 ? orow : 0)) * 4 + kg) << 5);
            const f16x8 zero8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            f16x8 fh = *reinterpret_cast<const f16x8 *>(tb), fl = *reinterpret_cast<const f16x8 *>(tb + 16);
            if (orow == 3) { fh = zero8; fl = zero8; }
            float4 R[TH / 4];                                    // this lane's pixels: (row 4 * pass + kg, column l15)
#pragma unroll
            for (int pass = 0; pass < TH / 4; ++pass) {
                f32x4 ah = {0.f, 0.f, 0.f, 0.f}, al = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f16x4 o0 = outv(0, pass * 4 + jj), o1 = outv(1, pass * 4 + jj);
                    const f16x8 b = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
                    const bool mine = qrow == jj;
                    ah = __builtin_amdgcn_mfma_f32_16x16x32_f16(mine ? fh : zero8, b, ah, 0, 0, 0);
                    al = __builtin_amdgcn_mfma_f32_16x16x32_f16(mine ? fl : zero8, b, al, 0, 0, 0);
                }
                constexpr float k = 1.f / 1024.f;
                R[pass] = make_float4(ah[0] + al[0] * k, ah[1] + al[1] * k, ah[2] + al[2] * k, 0.f);
            }
            const int pair = wave >> 1;
            if (wave & 1) {
#pragma unroll
                for (int pass = 0; pass < TH / 4; ++pass) *reinterpret_cast<float4 *>(stg + pass * 1024 + eln * 16) = R[pass];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // the partial sums are in LDS before the flag is
                s_flag[pair] = k + 1;
            } else {
                const char *pst = stg + 2 * W_SLOT;                          // the partner's strip: same slot parity, next wave's ring
                while (s_flag[pair] < k + 1) __builtin_amdgcn_s_sleep(2);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");        // the strip reads below stay behind the poll (compiler order, too)
#pragma unroll
                for (int pass = 0; pass < TH / 4; ++pass) {
                    const float4 v = *reinterpret_cast<const float4 *>(pst + pass * 1024 + eln * 16);
                    R[pass].x += v.x; R[pass].y += v.y; R[pass].z += v.z;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // (the reads have returned)
                s_flag[4 + pair] = k + 1;
#pragma unroll
                for (int pass = 0; pass < TH / 4; ++pass) {
                    const int oy = cur.oy0 + pass * 4 + kg, ox = cur.ox0 + l15;
                    const int Y = 2 * oy + (pair >> 1), X = 2 * ox + (pair & 1);
                    if (oy < p.Ho && ox < p.Wo && Y < p.Hd && X < p.Wd)
                        *reinterpret_cast<float4 *>(p.dst_dot + ((size_t)Y * p.Wd + X) * 4) = make_float4(R[pass].x, R[pass].y, R[pass].z, 0.f);
                }
            }
        } else {   // ST_POOL: 2x2 max; rows j, j+1 are in this lane, columns 2c, 2c+1 meet in the strip
#pragma unroll
            for (int pass = 0; pass < TH / 8; ++pass) {         // four pooled rows per pass
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const f16x4 u = outv(i, pass * 8 + 2 * jj), v = outv(i, pass * 8 + 2 * jj + 1);
                        f16x4 m;
#pragma unroll
                        for (int r = 0; r < 4; ++r) m[r] = u[r] > v[r] ? u[r] : v[r];
                        *reinterpret_cast<f16x4 *>(stg + st_off(jj * 16 + l15, i * 2 + (kg >> 1), kg & 1)) = m;
                    }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {                // 16 pooled pixels (2 rows x 8) per store
                    const int prow = rr * 2 + (s_px >> 3), pcol = s_px & 7;
                    const int px0 = prow * 16 + 2 * pcol, px1 = px0 + 1;
                    f16x8 v = *reinterpret_cast<const f16x8 *>(stg + px0 * 64 + ((s_slot ^ ((px0 >> 1) & 3)) << 4));
                    const f16x8 v1 = *reinterpret_cast<const f16x8 *>(stg + px1 * 64 + ((s_slot ^ ((px1 >> 1) & 3)) << 4));
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = v[r] > v1[r] ? v[r] : v1[r];
                    const int py = (cur.oy0 >> 1) + pass * 4 + prow, pxx = (cur.ox0 >> 1) + pcol;
                    const bool ok = py < p.Hd && pxx < p.Wd;
                    f16 *d = ok ? p.dst + ((size_t)py * p.Wd + pxx) * p.dstC + cur.n0 + wave * WCH + s_slot * 8 : reinterpret_cast<f16 *>(trash);
                    *reinterpret_cast<f16x8 *>(d) = v;
                }
            }
        }
        cur = nxt;
    }
}

template <int MODE, int TH>
hipError_t launch_mode(const ConvParams &p, int grid, hipStream_t stream)
{
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv_prw_kernel<MODE, TH>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SMEM, stream, p);
    return hipGetLastError();
}

}  // namespace

// 3x3, stride 1, pad 1, Cin (src0 [+ src1 concat]) multiple of 64, Cout == CoutPad multiple of 256, no residuals;
// store modes NHWC / PS / POOL (th = pixel rows per tile: 16 or 8) and PS_DOT3 (Cout = 256, th = 16).  One block per CU (n_cu), each walking tiles.
// hipErrorInvalidValue otherwise.
hipError_t conv_prw_launch(ConvParams p, int th, int n_cu, hipStream_t stream)
{
    if ((p.c0 % CT) || (p.c1 % CT) || p.c0 + p.c1 < CT || (p.CoutPad % BN) || p.Cout != p.CoutPad || p.res1 || p.res2 ||
        p.dst_full || !p.zeros || !p.trash || n_cu < 8 || (p.act != ACT_RELU && p.act != ACT_NONE) || (th != 8 && th != 16) ||
        (p.mode != ST_NHWC && p.mode != ST_PS && p.mode != ST_POOL && p.mode != ST_PS_DOT3) || (p.mode == ST_PS && (p.dstC % 64)) ||
        (p.mode == ST_PS_DOT3 && (p.CoutPad != 256 || p.dstC != 64 || !p.dotw || !p.dst_dot || th != 16)))
        return hipErrorInvalidValue;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + th - 1) / th;
    const int total = p.tiles_x * p.tiles_y * (p.CoutPad / BN);
    const int grid = total < n_cu ? total : n_cu;
    if (th == 16) {
        switch (p.mode) {
        case ST_NHWC: return launch_mode<ST_NHWC, 16>(p, grid, stream);
        case ST_PS: return launch_mode<ST_PS, 16>(p, grid, stream);
        case ST_PS_DOT3: return launch_mode<ST_PS_DOT3, 16>(p, grid, stream);
        default: return launch_mode<ST_POOL, 16>(p, grid, stream);
        }
    }
    switch (p.mode) {
    case ST_NHWC: return launch_mode<ST_NHWC, 8>(p, grid, stream);
    case ST_PS: return launch_mode<ST_PS, 8>(p, grid, stream);
    default: return launch_mode<ST_POOL, 8>(p, grid, stream);
    }
}
