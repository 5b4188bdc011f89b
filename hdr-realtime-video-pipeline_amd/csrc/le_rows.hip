// le_rows.hip -- row-streaming fused kernels of the LE main branch (gfx950): whole layer chains in ONE launch, every
// intermediate tensor in LDS rings of a few pixel rows, nothing but the chain's inputs and outputs in HBM.
//
//   le_rb_rows    ResBlock_with_SFT.forward (arch_util.py:89-95)   y = x + conv2(sft2(relu(conv1(sft1(x, c))), c))
//   le_head_rows  HDRUNet3T1_arch.py:168-172   fea0 = relu(HR_conv1(SFT_layer1(relu(conv_first(img)), cond1))), fea1 = relu(down_conv1(fea0))
//   le_tail_rows  HDRUNet3T1_arch.py:196-206   out = img + conv_last(relu(HR_conv2(SFT_layer2(relu(shuffle(up_conv3(u))) + fea0, cond1))))
// with SFTLayer.forward (arch_util.py:66-72).  The 16x16-tile kernels they replace (conv32s.hip, conv32p.hip,
// conv_tile_f16.hip: one launch per conv) write every 32-channel intermediate to HBM, read it back, and fetch each input halo 1.27x.
//
// Schedule (all three).  A workgroup owns a STRIP of 60 output columns and a SEGMENT of rows and walks down it two rows per
// step.  The stages of a chain run skewed against each other, each on rows the previous stage finished a step earlier; for the
// ResBlock:
//     step s:   LDS-DMA of rows 2s+6, 2s+7 (x: 64 px x 64 B, cond: 64 px x 32 B; three steps ahead)
//               sft1           -> Y1 rows 2s,   2s+1          (64 columns: the strip + 2 halo columns each side)
//               conv1 + sft2   -> Y2 rows 2s-3, 2s-2          (62 columns)
//               conv2 + x      -> out rows 2s-6, 2s-5         (60 columns; x from the ring the DMA filled)
// so a row is fetched ONCE (plus 4 of 64 columns shared with the neighbour strips and 4 rows per segment), there is no
// vertical recompute, and ONE s_barrier per step orders all rings (every ring slot is written and read in different steps).
// Waves have ROLES, so that a wave's 3x3 filter bank lives in its registers for the whole launch (18 A fragments = 72
// VGPRs; conv32s re-reads it from LDS for every 32 pixels); one 32-pixel group (row g >> 1, column half g & 1) per stage,
// step and wave.  Each SIMD holds one wave of either role; they meet at the barrier.
// What bounds these kernels is neither HBM (their byte floors are 0.07 / 0.20 / 0.20 ms at 3840x2160) nor the matrix pipe
// (30-45 % busy) but the instruction rate of a wave: with two waves per SIMD a wave issues one instruction per four cycles,
// whatever its kind, so every scalar and address instruction counts as much as an MFMA's issue slot (measured by leaving parts
// out: tools/abl_rows.sh).  Hence: ring positions are byte-offset cursors that advance by adds (no division per step), LDS
// addresses are integers with the buffer base folded into lane constants, the Y rings keep their first two rows a second time
// behind the last (a 3-row conv window never wraps: one address per fragment column, kernel rows as immediates), an SFT
// pass (a chain of dependent MFMA -> VALU -> MFMA steps) runs between the MFMAs of an independent conv, and ReLU / LeakyReLU /
// the residual add work on packed f16.
// Arithmetic, operand order and rounding points are those of the per-layer kernels (K order (tap, k-step) on
// v_mfma_f32_32x32x16_f16, SFT in packed f16, bias added in fp32 behind the sum, residual adds in f16): results are
// bit-identical to them (tests/test_gpu_le_rows.py).
#include "le_rows.h"

namespace {

// A 3x3 32 -> 32 filter bank as 18 A fragments of v_mfma_f32_32x32x16_f16 (wpk = [tap][coutp][32 in]; fragment (tap, ks):
// lane = out channel n0 + l31, input channels 16 ks + 8 lh ..)
struct Bank { f16x8 f[18]; };
__device__ __forceinline__ void load_bank(Bank &b, const f16 *wpk, int l31, int lh, int coutp = 32, int n0 = 0)
{
#pragma unroll
    for (int st = 0; st < 18; ++st)
        b.f[st] = *reinterpret_cast<const f16x8 *>(wpk + ((st >> 1) * coutp + n0 + l31) * 32 + 16 * (st & 1) + 8 * lh);
}
__device__ __forceinline__ void load_bias(f32x4 (&b)[4], const float *bias, int lh)
{
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) b[qd] = *reinterpret_cast<const f32x4 *>(bias + 8 * qd + 4 * lh);
}

// The SFT layer's operands (pack_sft, hdrtv_api.hip): three A fragments (hidden stack, scale head, shift head) and the hidden
// stack's bias tile (the accumulator init) in registers; the two heads' bias tiles ((scale + 1) folded into the first) in LDS,
// 2 lane halves x 16 floats each, read per pass
struct SftW { f16x8 a0, a1s, a1t; f32x16 bh; };
__device__ __forceinline__ void load_sft(SftW &s, const f16 *wfrag, const float *bias, int lane, int lh)
{
    const f16x8 *fr = reinterpret_cast<const f16x8 *>(wfrag);
    s.a0 = fr[lane]; s.a1s = fr[64 + lane]; s.a1t = fr[128 + lane];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) s.bh[4 * g + k] = bias[8 * g + 4 * lh + k];
}
__device__ __forceinline__ void sft_tiles_to_lds(char *dst, const float *bias, int tid)
{
    if (tid < 64) {
        const int t = tid >> 5, lh = (tid >> 4) & 1, j = tid & 15;
        reinterpret_cast<float *>(dst)[tid] = bias[32 * (t + 1) + 8 * (j >> 2) + 4 * lh + (j & 3)] + (t == 0 ? 1.f : 0.f);
    }
}
__device__ __forceinline__ f32x16 ld_tile(unsigned a)
{
    f32x16 t;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = lds_rd<f32x4>(a + 16 * g);
        t[4 * g + 0] = v[0]; t[4 * g + 1] = v[1]; t[4 * g + 2] = v[2]; t[4 * g + 3] = v[3];
    }
    return t;
}
// The reference's activation fake-quantiser (W8A8Conv2d.forward, hdrtvnet_torch.py:351-358) on four / eight f16 values in registers:
// u8 code by one FMA + v_cvt_pk_u8_f32 (rint, saturating; common.h quant4), back by v_cvt_f32_ubyteN + one FMA, rounded to f16
__device__ __forceinline__ f16x4 fq4(const f16x4 &x, const FqParam &q)
{
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)x[0], q.inv, q.zoff), 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)x[1], q.inv, q.zoff), 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)x[2], q.inv, q.zoff), 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)x[3], q.inv, q.zoff), 3, w);
    return cvt4(__builtin_fmaf((float)(w & 255u), q.scale, q.zero), __builtin_fmaf((float)((w >> 8) & 255u), q.scale, q.zero),
                __builtin_fmaf((float)((w >> 16) & 255u), q.scale, q.zero), __builtin_fmaf((float)(w >> 24), q.scale, q.zero));
}
__device__ __forceinline__ f16x8 fq8(const f16x8 &x, const FqParam &q)
{
    const f16x4 lo = fq4(__builtin_shufflevector(x, x, 0, 1, 2, 3), q), hi = fq4(__builtin_shufflevector(x, x, 4, 5, 6, 7), q);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ void fq_row(f16x4 (&y)[4], const FqParam &q)
{
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) y[qd] = fq4(y[qd], q);
}

// y = x * (scale + 1) + shift on one pixel's 16 channels of this lane (channel quads qd: channels 8 qd + 4 lh ..), in three
// stages so that a caller can put independent work between the dependent MFMAs (conv32s's arithmetic, bit for bit).
// tiles = LDS address of this lane half's head tiles (sft_tiles_to_lds + 64 lh)
// fq (W8A8 SFT convs): its four quantisers {cond -> scale branch, cond -> shift branch, scale hidden, shift hidden}, or null.  The
// two first layers then see differently quantised copies of the condition pixel: two MFMAs, each with the other branch's
// rows of the stacked A fragment zeroed (l31 < 16: scale branch).
__device__ __forceinline__ f32x16 sft_hidden(const SftW &s, const f16x8 &c0, const FqParam *fq = nullptr, int l31 = 0)
{
    if (fq) {
        const f16x8 z = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        const f32x16 h = __builtin_amdgcn_mfma_f32_32x32x16_f16(l31 < 16 ? s.a0 : z, fq8(c0, fq[0]), s.bh, 0, 0, 0);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(l31 < 16 ? z : s.a0, fq8(c0, fq[1]), h, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(s.a0, c0, s.bh, 0, 0, 0);
}
__device__ __forceinline__ void sft_heads(const SftW &s, const f32x16 &h, unsigned tiles, f32x16 &sc, f32x16 &sh, const FqParam *fq = nullptr)
{
    f16x8 hs = lrelu_pack16(h, 0), ht = lrelu_pack16(h, 1);
    if (fq) { hs = fq8(hs, fq[2]); ht = fq8(ht, fq[3]); }
    sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(s.a1s, hs, ld_tile(tiles), 0, 0, 0);
    sh = __builtin_amdgcn_mfma_f32_32x32x16_f16(s.a1t, ht, ld_tile(tiles + 128), 0, 0, 0);
}
__device__ __forceinline__ void sft_modulate(const f32x16 &sc, const f32x16 &sh, f16x4 (&y)[4])
{
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
        y[qd] = y[qd] * cvt4(sc[4 * qd], sc[4 * qd + 1], sc[4 * qd + 2], sc[4 * qd + 3]) + cvt4(sh[4 * qd], sh[4 * qd + 1], sh[4 * qd + 2], sh[4 * qd + 3]);
}

// 3x3 conv of one 32-pixel group out of a mirrored ring: va[kx][ks] = this lane's fragment address (kernel column kx, k-step
// ks; buffer base included) in ring row 0, `win` the byte offset of the window's first row; the window's rows are ROWB apart.
// K order (tap, k-step); reads run AHEAD steps in front of the MFMAs; hook(st) runs behind MFMA st.
// PIN: left alone, hipcc sinks the fragment reads down to their MFMAs to save registers (the last six became read -> lgkmcnt(0)
// -> MFMA pairs, a full LDS round trip each).  1: source order is the schedule (sched_barrier per step; for convs without hooks:
// a hook's VALU would sit between two MFMAs as one block).  2: reads and MFMAs alternate as written, everything else floats
// (sched_group_barrier; for convs with hooks).  0: hipcc's order.
#ifndef ROWS_PRIO_MASK
#define ROWS_PRIO_MASK 4   // which convs raise their wave's priority (s_setprio 1) for the length of their MFMA stream: bit 0 / 1 ResBlock conv1 / conv2,
#endif                     // 2 / 3 tail up_conv / HR_conv2 + conv_last, 4 / 5 head down_conv1 duty / HR_conv1
template <int AHEAD, int ROWB, int PIN, int PRIO = 0, class Hook>
__device__ __forceinline__ f32x16 conv18(const Bank &w, const unsigned (&va)[3][2], int win, Hook hook)
{
    unsigned a[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a[kx][ks] = va[kx][ks] + (unsigned)win;
    f32x16 acc;
    f16x8 x[18];
    auto ld = [&](int st) __attribute__((always_inline)) {
        const int tap = st >> 1, ks = st & 1, ky = tap / 3, kx = tap % 3;
        x[st] = lds_rd<f16x8>(a[kx][ks] + ky * ROWB);
    };
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    if (PIN == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int st = 0; st < AHEAD; ++st) ld(st);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
        if (PIN == 1) __builtin_amdgcn_sched_barrier(0);
        if (st + AHEAD < 18) ld(st + AHEAD);
        if (RB_ABL & 4) {
            if (st == 0) acc = zero16();
            acc[st & 15] += (float)x[st][0] * (float)w.f[st][0];
        } else if (st == 0) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.f[0], x[0], zero16(), 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.f[st], x[st], acc, 0, 0, 0);
        }
        hook(st);
    }
    if (PIN == 2) {
        // AHEAD reads, then {MFMA, read} pairs, then the remaining MFMAs; a hook's MFMAs and LDS reads take slots of the same
        // sequence (the pattern is longer than the conv alone needs)
        __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
        for (int st = 0; st < 18 - AHEAD + 12; ++st) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, AHEAD, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    return acc;
}
// ------------------------------------------------------------------------------------------------------------------------
// Fused ResBlock_with_SFT.  Roles: B (waves 0-3) conv1 + sft2 -- LDS to LDS, not one memory operation; C (waves 4-7) the
// LDS-DMA, sft1, conv2, the residual and the stores.  Both SFT passes run inside a conv's MFMA stream.
template <int DPF> struct RbGeo {
    static constexpr int LAG = 6;                            // output rows trail the sft1 rows by LAG
    static constexpr int XR = 2 * DPF + LAG + 2;             // x ring: fetched 2 DPF rows ahead, read again LAG rows later (the residual)
    static constexpr int CR = 2 * DPF + LAG;                 // condition ring: last read by sft2, 3 rows behind sft1
    static constexpr int OFF_X = 0, OFF_C = OFF_X + XR * X_ROWB, OFF_Y1 = OFF_C + CR * C_ROWB, OFF_Y2 = OFF_Y1 + YPH * Y_ROWB;
    static constexpr int OFF_B = OFF_Y2 + YPH * Y_ROWB;      // the head tiles of sft2, then of sft1, then conv2's bias
    static constexpr int SMEM = OFF_B + 2 * SFT_TILE_B + 128;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(BIG % XR == 0 && BIG % CR == 0 && BIG % YN == 0, "BIG");
};

template <int DPF, bool FQ>
__global__ __launch_bounds__(512) void le_rb_rows_kernel(RowsRbParams p)
{
    using G = RbGeo<DPF>;
    const FqParam *const fqs1 = (FQ && (p.fq & 4)) ? p.fq_s1 : nullptr, *const fqs2 = (FQ && (p.fq & 8)) ? p.fq_s2 : nullptr;
    constexpr int LAG = G::LAG, XR = G::XR, CR = G::CR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned sm = lds_off(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS;
    const int y0 = seg * p.rows_per_seg, y1 = min(y0 + p.rows_per_seg, p.H);
    const int ya = y0 - 2;                                             // image row of ring row 0
    const int nsteps = (y1 - ya + LAG - 1) / 2 + 1;
    const int H = p.H, W = p.W;
    sft_tiles_to_lds(smem + G::OFF_B, p.sft2_bias, tid);
    sft_tiles_to_lds(smem + G::OFF_B + SFT_TILE_B, p.sft1_bias, tid);
    if (tid < 32) reinterpret_cast<float *>(smem + G::OFF_B + 2 * SFT_TILE_B)[tid] = p.b2[tid];

    const int g = wave & 3, gr = g >> 1, gh = g & 1;                   // this wave's 32-pixel group: row gr of the step's pair, column half gh
    const int cx = 32 * gh + l31;                                      // this lane's pixel slot in its group's ring rows
    unsigned va[3][2];                                                 // conv fragments: output slot cx reads input slots cx .. cx + 2
    frag_addr<LStd>(va, sm, cx, lh);
    const bool colfull = x0 >= 2 && x0 + 62 <= W;                      // no column of the strip's halo lies outside the image

    if (wave < 4) {
        // ------------------------------------------------------------------ role B: conv1 + sft2 (its cond MLPs between conv1's MFMAs)
        Bank w1;
        load_bank(w1, p.w1, l31, lh);
        SftW s2;
        load_sft(s2, p.sft2_wfrag, p.sft2_bias, lane, lh);
        f32x4 bq[4];
        load_bias(bq, p.b1, lh);
        const unsigned t2 = sm + G::OFF_B + 64 * lh;
        unsigned vq[2];                                                // Y2 write: this lane's two chunks of slot cx
        chunk_addr<LStd>(vq, sm, cx, lh);
        const unsigned vc2 = sm + LCond::at(cx + 1, lh);               // Y2 slot cx = image column x0 - 1 + cx = condition slot cx + 1
        const bool col2 = (unsigned)(x0 - 1 + cx) < (unsigned)W;
        // ring rows of step s: conv1 + sft2 on rb = 2 s - 3 + gr
        Cur<G::OFF_C, CR, C_ROWB> cb(gr - 3);
        Cur<G::OFF_Y1, YN, Y_ROWB> wn(gr - 4);
        Cur<G::OFF_Y2, YN, Y_ROWB> yb2(gr - 3);
        int rb_img = ya + gr - 3;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            if (RB_ABL & 32) { __builtin_amdgcn_s_barrier(); continue; }
            const f16x8 c2 = lds_rd<f16x8>(vc2 + cb.o);
            const bool row2 = (unsigned)rb_img < (unsigned)H;          // outside the image: conv2's zero padding
            // the SFT pass is a chain of dependent MFMA -> VALU -> MFMA steps, the conv an independent stream that covers its
            // latencies: the pass's two cond MLPs run between the conv's MFMAs, only the modulation waits for the conv
            f32x16 h2, sc2, sh2;
            const f32x16 acc = conv18<ROWS_AHEAD_H - (FQ ? 1 : 0), Y_ROWB, ROWS_PIN_H, ROWS_PRIO_MASK & 1>(w1, va, wn.o, [&](int st) __attribute__((always_inline)) {
                if (RB_ABL & 8) return;
                if (st == 2) h2 = sft_hidden(s2, c2, fqs2, l31);
                if (st == 10) sft_heads(s2, h2, t2, sc2, sh2, fqs2);
            });
            STAMP(1);
            f16x4 y[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) y[qd] = __builtin_elementwise_max(bias_cvt4(acc, qd, bq[qd]), zero4());
            if (!(RB_ABL & 8)) {
                sft_modulate(sc2, sh2, y);
                if (FQ && (p.fq & 2)) fq_row(y, p.fq_c2);                 // conv2 is W8A8: its input quantiser
                if (!(colfull && row2)) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) if (!(col2 && row2)) y[qd] = zero4();
                }
            }
            put_row(vq, yb2.o, yb2.mirrored(), y);
            cb.step(); wn.step(); yb2.step();
            rb_img += 2;
            STAMP(2);
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
            STAMP(4);
            if (!(RB_ABL & 16)) __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    } else {
        // ------------------------------------------------------------------ role C: the DMA, sft1 (between conv2's MFMAs), conv2 + x, stores
        Bank w2;
        load_bank(w2, p.w2, l31, lh);
        f32x4 bq[4];
        load_bias(bq, p.b2, lh);
        SftW s1;
        load_sft(s1, p.sft1_wfrag, p.sft1_bias, lane, lh);
        const unsigned t1 = sm + G::OFF_B + SFT_TILE_B + 64 * lh;
        unsigned vq[2];                                                // x read, Y1 write: this lane's two chunks of slot cx
        chunk_addr<LStd>(vq, sm, cx, lh);
        const unsigned vc1 = sm + LCond::at(cx, lh);                   // slot cx = image column x0 - 2 + cx
        const bool col1 = (unsigned)(x0 - 2 + cx) < (unsigned)W;
        const dma_rsrc_t rx = dma_rsrc(p.x), rc = dma_rsrc(p.cond);
        // per step: pieces 2 gh, 2 gh + 1 of x row gr (16 pixels x 64 B each) and piece gh of condition row gr (32 pixels x 32 B)
        const unsigned xl0 = LStd::src_off(2 * gh, lane), xl1 = LStd::src_off(2 * gh + 1, lane), cl = LCond::src_off(gh, lane);
        const bool xok0 = (unsigned)(x0 - 2 + LStd::src_px(2 * gh, lane)) < (unsigned)W, xok1 = (unsigned)(x0 - 2 + LStd::src_px(2 * gh + 1, lane)) < (unsigned)W;
        const bool cok = (unsigned)(x0 - 2 + LCond::src_px(gh, lane)) < (unsigned)W;
        auto issue = [&](int r, int xo_, int co_) __attribute__((always_inline)) {        // image row r into the ring rows at xo_ / co_
            if (RB_ABL & 1) return;
            const bool rok = (unsigned)r < (unsigned)H && r <= y1 + 1;
            const unsigned pix = (unsigned)(r * W + x0 - 2);
            dma16_at(rx, sm + xo_ + (2 * gh) * 1024, (rok && xok0) ? pix * 64u + xl0 : DMA_OOB);
            dma16_at(rx, sm + xo_ + (2 * gh + 1) * 1024, (rok && xok1) ? pix * 64u + xl1 : DMA_OOB);
            dma16_at(rc, sm + co_ + gh * 1024, (rok && cok) ? pix * 32u + cl : DMA_OOB);
        };
        // epilogue: the result's quads become this lane's two chunks of its pixel (quads_to_chunks: no LDS round trip), the
        // residual is added in that form (x of output column x0 + cx: ring slot cx + 2, read as the same two chunks) and the lane
        // stores 2 x 16 bytes: the two lanes of a pixel write bytes 0-31 with one store instruction, 32-63 with the next
        char *trash = p.trash + tid * 16;
        unsigned vr[2];
        chunk_addr<LStd>(vr, sm, cx + 2, lh);
        const bool ocol = cx < WS && x0 + cx < W;
        f16 *const dst0 = p.dst + (size_t)(x0 + cx) * 32 + 8 * lh;
        const int yend = y1;                                           // (the step loop has a local y1: the sft1 row)
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq)
            issue(ya + 2 * sq + gr, G::OFF_X + ((2 * sq + gr + BIG) % XR) * X_ROWB, G::OFF_C + ((2 * sq + gr + BIG) % CR) * C_ROWB);
        // ring rows of step s: DMA into 2 (s + DPF) + gr; sft1 on ra = 2 s + gr; conv2 + residual on ro = ra - LAG
        Cur<G::OFF_X, XR, X_ROWB> xd(2 * DPF + gr), xa(gr), xres(gr - LAG);
        Cur<G::OFF_C, CR, C_ROWB> cd(2 * DPF + gr), ca(gr);
        Cur<G::OFF_Y1, YN, Y_ROWB> ya1(gr);
        Cur<G::OFF_Y2, YN, Y_ROWB> wn(gr - LAG - 1);
        int ro_img = ya + gr - LAG;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            issue(ro_img + LAG + 2 * DPF, xd.o, cd.o);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            if (RB_ABL & 64) {
                __builtin_amdgcn_s_waitcnt(waitcnt_imm(3 * (DPF - 1), 0));
                __builtin_amdgcn_s_barrier();
                xd.step(); cd.step(); ro_img += 2;
                continue;
            }
            const f16x8 c1 = lds_rd<f16x8>(vc1 + ca.o);
            f16x4 y1[4];
            get_row(vq, xa.o, y1);
            const f16x8 res0 = lds_rd<f16x8>(vr[0] + (unsigned)xres.o), res1 = lds_rd<f16x8>(vr[1] + (unsigned)xres.o);
            const bool row1 = (unsigned)(ro_img + LAG) < (unsigned)H;  // outside the image: conv1's zero padding
            // conv2 on row ro with row ra's whole SFT pass (independent of it) between its MFMAs
            f32x16 h1, sc1, sh1;
            const f32x16 acc = conv18<ROWS_AHEAD_H + (FQ ? 0 : 2), Y_ROWB, ROWS_PIN_H, (ROWS_PRIO_MASK >> 1) & 1>(w2, va, wn.o, [&](int st) __attribute__((always_inline)) {
                if (RB_ABL & 8) { if (st == 13) put_row(vq, ya1.o, ya1.mirrored(), y1); return; }
                if (st == 1) h1 = sft_hidden(s1, c1, fqs1, l31);
                if (st == 7) sft_heads(s1, h1, t1, sc1, sh1, fqs1);
                if (st == 13) {
                    sft_modulate(sc1, sh1, y1);
                    if (FQ && (p.fq & 1)) fq_row(y1, p.fq_c1);            // conv1 is W8A8: its input quantiser
                    if (!(colfull && row1)) {
#pragma unroll
                        for (int qd = 0; qd < 4; ++qd) if (!(col1 && row1)) y1[qd] = zero4();
                    }
                    put_row(vq, ya1.o, ya1.mirrored(), y1);
                }
            });
            STAMP(1);
            if (FQ) {       // (the fake-quant variant has no registers to spare for the bias: from LDS, behind the conv)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) bq[qd] = lds_rd<f32x4>(sm + G::OFF_B + 2 * SFT_TILE_B + 32 * qd + 16 * lh);
            }
            {
                f16x4 o[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) o[qd] = bias_cvt4(acc, qd, bq[qd]);
                f16x8 o0, o1;
                quads_to_chunks(o, o0, o1);
                o0 += res0; o1 += res1;                                // x + conv2(..): one f16 rounding per element, as in the per-layer kernel
                f16 *d = (ocol && ro_img >= y0 && ro_img < yend) ? dst0 + (size_t)ro_img * W * 32 : reinterpret_cast<f16 *>(trash);
                if (RB_ABL & 2) d = reinterpret_cast<f16 *>(trash);
                *reinterpret_cast<f16x8 *>(d) = o0;
                *reinterpret_cast<f16x8 *>(d == reinterpret_cast<f16 *>(trash) ? d : d + 16) = o1;
            }
            xd.step(); xa.step(); xres.step(); cd.step(); ca.step(); ya1.step(); wn.step();
            ro_img += 2;
            STAMP(3);
            // per step and wave: three DMA pieces, then (behind the conv) two stores, all always issued: the pieces that step s + 1
            // reads were issued at the top of step s + 1 - DPF, in front of 5 (DPF - 1) + 2 younger operations
            __builtin_amdgcn_s_waitcnt(waitcnt_imm((RB_ABL & 3) ? 0 : 5 * (DPF - 1) + 2, 0));
            STAMP(4);
            if (!(RB_ABL & 16)) __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The full-resolution tail of the LE net: u is the half-resolution trunk output; the per-layer form is conv32p<4> (up-conv +
// PixelShuffle + ReLU + skip, 0.53 GB written), conv32s<sft> (read back, 0.53 GB written) and conv32s<plain, planar> (read
// back).  Here the two 32-channel full-resolution tensors live in LDS rings (Y: the modulated up-conv output, Z:
// relu(HR_conv2)); HBM sees u, fea0, cond1, the residual planes and the three output planes.
//     step s:  role T1 (waves 0-3): LDS-DMA of fea0 / cond rows 2s+6, 2s+7, of u row s+4 (half resolution: 34 of 48 slots used;
//                   rows 0 and 1 of a ring lap a second time behind the last) and of the residual planes of output rows 2s, 2s+1
//                   (4-byte DMA; a plain load's first use would wait, vmcnt retiring in order, for every DMA issued before it);
//                   wave b = PixelShuffle position (b >> 1, b & 1): up_conv3 bank b on u rows s-1 .. s+1 -> 32 half-resolution
//                   pixels = every other pixel of full-resolution row 2s + (b >> 1); ReLU, + fea0, SFT_layer2 (its cond MLPs inside
//                   the conv's MFMA stream) -> Y rows 2s, 2s+1
//              role T2 (waves 4-7, group (g >> 1, g & 1)): HR_conv2 + ReLU -> Z rows 2s-3, 2s-2; conv_last + residual -> output
//                   rows 2s-6, 2s-5 (planar)
constexpr int U_SLOTS = 48, U_ROWB = U_SLOTS * 64, UN = 8, UPH = UN + 2;
template <int DPF> struct TailGeo {
    static constexpr int LAG = 6;
    static constexpr int FR = 2 * DPF + 2;                   // fea0 / cond rings: fetched 2 DPF rows ahead of their one use
    static_assert(DPF + 4 <= UN, "u ring: the window's three rows, the row being quantised in place (W8A8 up_conv), DPF rows in flight");
    static constexpr int OFF_U = 0, OFF_F = OFF_U + UPH * U_ROWB, OFF_C = OFF_F + FR * X_ROWB, OFF_Y = OFF_C + FR * C_ROWB;
    static constexpr int OFF_Z = OFF_Y + YPH * Y_ROWB, OFF_TR = OFF_Z + YPH * Y_ROWB;       // TR: 4 x 1 KiB the unused mirror DMAs land in
    static constexpr int R_SLOTB = 256, RN = DPF + 1;        // residual planes: per group RN slots of [3 planes][32 px] f16
    static constexpr int OFF_R = OFF_TR + 4096, OFF_B = OFF_R + 4 * RN * R_SLOTB;           // B: SFT_layer2's head tiles
    static constexpr int SMEM = OFF_B + SFT_TILE_B;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(BIG % FR == 0 && BIG % UN == 0, "BIG");
};

template <int DPF, bool FQ>
__global__ __launch_bounds__(512) void le_tail_rows_kernel(RowsTailParams p)
{
    using G = TailGeo<DPF>;
    const FqParam *const fqs = (FQ && (p.fq & 8)) ? p.fq_s : nullptr;
    constexpr int FR = G::FR, LAG = G::LAG, RN = G::RN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned sm = lds_off(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS, hx0 = x0 >> 1;
    const int y0 = seg * p.rows_per_seg, y1 = min(y0 + p.rows_per_seg, p.H);      // rows_per_seg is even
    const int ya = y0 - 2, hya = ya >> 1;                              // image row of ring row 0 (even); its half-resolution row
    const int nsteps = (y1 - ya + LAG - 1) / 2 + 1;
    const int H = p.H, W = p.W, H1 = H >> 1, W1 = W >> 1;
    sft_tiles_to_lds(smem + G::OFF_B, p.sft_bias, tid);
    const bool colfull = x0 >= 2 && x0 + 62 <= W;
    const int g = wave & 3, gr = g >> 1, gh = g & 1;

    if (wave < 4) {
        // ------------------------------------------------------------------ role T1
        Bank wu;
        load_bank(wu, p.w_up, l31, lh, 128, 32 * g);
        SftW s2;
        load_sft(s2, p.sft_wfrag, p.sft_bias, lane, lh);
        const unsigned t2 = sm + G::OFF_B + 64 * lh;
        f32x4 bq[4];
        load_bias(bq, p.b_up + 32 * g, lh);
        unsigned va[3][2];                                             // half-resolution pixel l31 reads u slots l31 .. l31 + 2
        frag_addr<LStd>(va, sm, l31, lh);
        const int cx = 2 * l31 + gh;                                   // full-resolution slot (image column x0 - 2 + cx) of this lane's pixel
        unsigned vf[2], vq[2];                                         // fea0 read, Y write: this lane's two chunks of slot cx (stride 2 across the lanes)
        chunk_addr<LTailF>(vf, sm, cx, lh);
        chunk_addr<LTailY>(vq, sm, cx, lh);
        const unsigned vc = sm + LCondT::at(cx, lh);
        const bool col = (unsigned)(x0 - 2 + cx) < (unsigned)W;
        // the DMA: pieces 2 gh, 2 gh + 1 of fea0 row gr, piece gh of condition row gr, (waves 0-2) piece g of the u row and of
        // its second copy, the residual planes of 32 pixels of an output row (lane = plane * 16 + pixel pair) -- always six
        // DMA instructions per step (the unused ones fetch nothing into a trash KiB)
        const dma_rsrc_t rf = dma_rsrc(p.fea0), rc = dma_rsrc(p.cond), ru = dma_rsrc(p.u), rres = dma_rsrc(p.res_planar);
        const unsigned fl0 = LTailF::src_off(2 * gh, lane), fl1 = LTailF::src_off(2 * gh + 1, lane), cl = LCondT::src_off(gh, lane), ul = LStd::src_off(g, lane);
        const bool fok0 = (unsigned)(x0 - 2 + LTailF::src_px(2 * gh, lane)) < (unsigned)W, fok1 = (unsigned)(x0 - 2 + LTailF::src_px(2 * gh + 1, lane)) < (unsigned)W;
        const bool cok = (unsigned)(x0 - 2 + LCondT::src_px(gh, lane)) < (unsigned)W;
        const bool uok = g < 3 && LStd::src_px(g, lane) < 34 && (unsigned)(hx0 - 2 + LStd::src_px(g, lane)) < (unsigned)W1;
        const size_t plane = (size_t)H * W;
        const unsigned rl = (unsigned)((lane >> 4) * plane * 2 + (32 * gh + 2 * (lane & 15)) * 2);     // plane, pixel pair
        const bool rlok = lane < 48 && x0 + 32 * gh + 2 * (lane & 15) < W;
        const unsigned tr = sm + G::OFF_TR + g * 1024, rbuf = sm + G::OFF_R + g * RN * G::R_SLOTB;
        auto issue_fc = [&](int r, int fo, int co) __attribute__((always_inline)) {
            const bool rok = (unsigned)r < (unsigned)H && r <= y1 + 1;
            const unsigned pix = (unsigned)(r * W + x0 - 2);
            dma16_at(rf, sm + fo + (2 * gh) * 1024, (rok && fok0) ? pix * 64u + fl0 : DMA_OOB);
            dma16_at(rf, sm + fo + (2 * gh + 1) * 1024, (rok && fok1) ? pix * 64u + fl1 : DMA_OOB);
            dma16_at(rc, sm + co + gh * 1024, (rok && cok) ? pix * 32u + cl : DMA_OOB);
        };
        auto issue_u = [&](int hr, int uo, bool mir) __attribute__((always_inline)) {     // half-resolution image row hr into the ring row at uo
            const bool rok = (unsigned)hr < (unsigned)H1 && hr <= ((y1 + 1) >> 1) + 1;
            const unsigned off = (rok && uok) ? (unsigned)((hr * W1 + hx0 - 2) * 64) + ul : DMA_OOB;
            dma16_at(ru, g < 3 ? sm + uo + g * 1024 : tr, off);
            dma16_at(ru, (g < 3 && mir) ? sm + uo + UN * U_ROWB + g * 1024 : tr, mir ? off : DMA_OOB);
        };
        auto issue_r = [&](int r, int slot) __attribute__((always_inline)) {              // residual of output row r (32 pixels of column half gh)
            const bool rok = r >= y0 && r < y1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rres, (__attribute__((address_space(3))) void *)(uintptr_t)(rbuf + slot * G::R_SLOTB), 4,
                                                     (rok && rlok) ? (unsigned)((r * W + x0) * 2) + rl : DMA_OOB, 0, 0, 0);
        };
        {
            const int m = (-1 + BIG) % UN;
            issue_u(hya - 1, G::OFF_U + m * U_ROWB, m < 2);
        }
        // (UA: with a W8A8 up_conv the u rows are fetched one step further ahead -- a row is quantised in place one step before its
        // first window, and must have LANDED by then: the pieces a step may rely on are those issued DPF or more steps earlier)
        constexpr int UA = FQ ? 1 : 0;
#pragma unroll
        for (int sq = 0; sq <= DPF + UA; ++sq) {
            const int m = (sq + BIG) % UN;
            issue_u(hya + sq, G::OFF_U + m * U_ROWB, m < 2);
            if (sq < DPF) {
                issue_fc(ya + 2 * sq + gr, G::OFF_F + ((2 * sq + gr + BIG) % FR) * X_ROWB, G::OFF_C + ((2 * sq + gr + BIG) % FR) * C_ROWB);
                issue_r(ya + 2 * sq - LAG + gr, sq % RN);
            }
        }
        // ring rows of step s: this wave produces Y row ra = 2 s + gr (from fea0 / cond row ra, u rows s - 1 .. s + 1); DMA into
        // fea0 / cond row 2 (s + DPF) + gr, u row s + DPF + 1, residual slot (s + DPF) % RN
        Cur<G::OFF_F, FR, X_ROWB> fa(gr), fd(2 * DPF + gr);
        Cur<G::OFF_C, FR, C_ROWB> ca(gr), cd(2 * DPF + gr);
        Cur<G::OFF_Y, YN, Y_ROWB> yw(gr);
        int uw = G::OFF_U + ((-1 + BIG) % UN) * U_ROWB, ud = G::OFF_U + ((DPF + 1 + UA + BIG) % UN) * U_ROWB, rs = DPF % RN;
        int ra_img = ya + gr;
        // W8A8 up_conv: its input quantiser runs ONCE per element over a landed u row, in place (a quarter of the row per wave: 34 of
        // the row's 136 16-byte chunks), one step before the row's first window (row s + 2 at step s: issued at step s - DPF);
        // rows 0 / 1 of a lap also into their second copy
        auto fq_u_row = [&](int uo, int hr) __attribute__((always_inline)) {       // hr: the row's half-resolution image row
            const int upx = (34 * g + lane) >> 2;
            if (lane < 34 && (unsigned)hr < (unsigned)H1 && (unsigned)(hx0 - 2 + upx) < (unsigned)W1) {
                const unsigned a = sm + uo + (34 * g + lane) * 16;
                const f16x8 v = fq8(lds_rd<f16x8>(a), p.fq_u);
                lds_wr(a, v);
                if (uo < G::OFF_U + 2 * U_ROWB) lds_wr(a + UN * U_ROWB, v);
            }
        };
        int uq = G::OFF_U + ((2 + BIG) % UN) * U_ROWB;                 // the row the pass of step 0 takes: u row 2
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        if (FQ && (p.fq & 1)) {
            __builtin_amdgcn_s_barrier();                              // (every wave's prologue pieces have landed)
#pragma unroll
            for (int ur = -1; ur <= 1; ++ur) fq_u_row(G::OFF_U + ((ur + BIG) % UN) * U_ROWB, hya + ur);
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
        }
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            issue_fc(ra_img + 2 * DPF, fd.o, cd.o);
            issue_u(hya + s + DPF + 1 + UA, ud, ud < G::OFF_U + 2 * U_ROWB);
            issue_r(ra_img + 2 * DPF - LAG, rs);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            if (RB_ABL & 32) {
                __builtin_amdgcn_s_waitcnt(waitcnt_imm(6 * (DPF - 1), 0));
                __builtin_amdgcn_s_barrier();
                fd.step(); cd.step(); ud += U_ROWB; if (ud >= G::OFF_U + UN * U_ROWB) ud -= UN * U_ROWB; rs = rs + 1 == RN ? 0 : rs + 1; ra_img += 2;
                continue;
            }
            if (FQ && (p.fq & 1)) fq_u_row(uq, hya + s + 2);
            const f16x8 c2 = lds_rd<f16x8>(vc + ca.o);
            f16x4 sk[4];
            get_row(vf, fa.o, sk);
            const bool row = (unsigned)ra_img < (unsigned)H;
            f32x16 h2, sc2, sh2;
            const f32x16 acc = conv18<ROWS_AHEAD_H, U_ROWB, ROWS_PIN_H, (ROWS_PRIO_MASK >> 2) & 1>(wu, va, uw, [&](int st) __attribute__((always_inline)) {
                if (st == 2) h2 = sft_hidden(s2, c2, fqs, l31);
                if (st == 9) sft_heads(s2, h2, t2, sc2, sh2, fqs);
            });
            STAMP(1);
            f16x4 y[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) y[qd] = __builtin_elementwise_max(bias_cvt4(acc, qd, bq[qd]), zero4()) + sk[qd];
            sft_modulate(sc2, sh2, y);
            if (FQ && (p.fq & 2)) fq_row(y, p.fq_y);                      // HR_conv2 is W8A8: its input quantiser
            if (!(colfull && row)) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) if (!(col && row)) y[qd] = zero4();
            }
            put_row(vq, yw.o, yw.mirrored(), y);
            fa.step(); fd.step(); ca.step(); cd.step(); yw.step();
            uw += U_ROWB; if (uw >= G::OFF_U + UN * U_ROWB) uw -= UN * U_ROWB;
            ud += U_ROWB; if (ud >= G::OFF_U + UN * U_ROWB) ud -= UN * U_ROWB;
            uq += U_ROWB; if (uq >= G::OFF_U + UN * U_ROWB) uq -= UN * U_ROWB;
            rs = rs + 1 == RN ? 0 : rs + 1;
            ra_img += 2;
            STAMP(2);
            // per step and wave six DMA instructions and nothing else: those of step s + 1 are older than the 6 (DPF - 1) since
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(6 * (DPF - 1), 0));
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    } else {
        // ------------------------------------------------------------------ role T2
        Bank wh, wl;
        load_bank(wh, p.w_hr, l31, lh);
        load_bank(wl, p.w_last, l31, lh);
        f32x4 bh[4];
        load_bias(bh, p.b_hr, lh);
        const float bl0 = p.b_last[0], bl1 = p.b_last[1], bl2 = p.b_last[2];
        const int cx = 32 * gh + l31;
        unsigned va[3][2], vy[3][2], vq[2];
        frag_addr<LStd>(va, sm, cx, lh);                               // conv_last's fragments of the Z ring
        frag_addr<LTailY>(vy, sm, cx, lh);                             // HR_conv2's of the Y ring
        chunk_addr<LStd>(vq, sm, cx, lh);                              // Z write
        const bool colz = (unsigned)(x0 - 1 + cx) < (unsigned)W;       // Z slot cx = image column x0 - 1 + cx
        const unsigned rbuf = sm + G::OFF_R + g * RN * G::R_SLOTB + l31 * 2;
        // output: channels 0..2 of pixel l31 sit in accumulator registers 0..2 of the lanes with lh == 0
        char *trash = p.trash + tid * 16;
        const size_t plane = (size_t)H * W;
        const bool cok = lh == 0 && cx < WS && x0 + cx < W;
        // ring rows of step s: HR_conv2 on rb = 2 s - 3 + gr (window rb - 1 ..), conv_last on ro = rb - 3
        Cur<G::OFF_Y, YN, Y_ROWB> wy(gr - 4);
        Cur<G::OFF_Z, YN, Y_ROWB> zw(gr - 3), wz(gr - LAG - 1);
        int rb_img = ya + gr - 3, rs = 0;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        if (FQ && (p.fq & 1)) __builtin_amdgcn_s_barrier();            // (role T1's in-place pass over the first u rows)
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            if (RB_ABL & 64) { for (int ch = 0; ch < 3; ++ch) *reinterpret_cast<f16 *>(trash) = (f16)0.f; __builtin_amdgcn_s_barrier(); continue; }
            const int r = rb_img - 3;                                  // the output row
            f16 res[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) res[ch] = lds_rd<f16>(rbuf + rs * G::R_SLOTB + ch * 64);
            STAMP(0);
            {   // conv_last + residual -> the three output planes
                const f32x16 acc = conv18<ROWS_AHEAD_F, Y_ROWB, ROWS_PIN_F, (ROWS_PRIO_MASK >> 3) & 1>(wl, va, wz.o, [](int) {});
                const float o[3] = {acc[0] + bl0, acc[1] + bl1, acc[2] + bl2};
                const bool ok = cok && r >= y0 && r < y1;
                f16 *d = p.dst_planar + (size_t)r * W + x0 + cx;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    // the conv result is rounded to f16, the residual added in fp32 and the sum rounded again (conv32s PLANAR)
                    const float v = (float)(f16)o[ch] + (float)res[ch];
                    *(ok ? d + ch * plane : reinterpret_cast<f16 *>(trash)) = (f16)v;
                }
            }
            STAMP(1);
            {   // HR_conv2 + ReLU -> Z
                const f32x16 acc = conv18<ROWS_AHEAD_F, Y_ROWB, ROWS_PIN_F, (ROWS_PRIO_MASK >> 3) & 1>(wh, vy, wy.o, [](int) {});
                f16x4 z[4];
                const bool in = colz && (unsigned)rb_img < (unsigned)H;     // outside the image: conv_last's zero padding
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    z[qd] = __builtin_elementwise_max(bias_cvt4(acc, qd, bh[qd]), zero4());
                    if (FQ && (p.fq & 4)) z[qd] = fq4(z[qd], p.fq_z);     // conv_last is W8A8: its input quantiser
                    if (!in) z[qd] = zero4();
                }
                put_row(vq, zw.o, zw.mirrored(), z);
            }
            wy.step(); zw.step(); wz.step();
            rb_img += 2;
            rs = rs + 1 == RN ? 0 : rs + 1;
            STAMP(3);
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));            // stores are never waited for
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The full-resolution head of the LE net.  The per-layer form is conv32s<c3+sft> (conv_first recomputed per 16x16 tile on the
// prep waves: 2.4 TB/s) and conv_t16 (fea0 read back).  Here HBM sees the three image planes, cond1, and the two outputs.
//     step s:  role H1 (waves 0-3, group (g >> 1, g & 1)): global loads of image rows 2s+3, 2s+4 into registers, staged as
//                   {R, G, B, 1} pixels into the patch ring at the end of the step; conv_first + ReLU + SFT_layer1 -> Y rows 2s, 2s+1;
//                   wave (s & 3): down_conv1 on half-resolution row s-3 (fea0 rows 2s-7 .. 2s-5 of the F ring) -> fea1
//              role H2 (waves 4-7): LDS-DMA of cond rows 2s+6, 2s+7; HR_conv1 + ReLU -> F rows 2s-3, 2s-2 and (one step
//                   later, through a strip) fea0
// conv_first is conv32s's C3 form (K = (kernel column | pad, channel | bias slot) per kernel row, three MFMAs).
constexpr int P_SLOTS = 72, P_ROWB = P_SLOTS * 8;            // patch ring: 68 of 72 pixel slots used (image columns x0 - 3 .. x0 + 64)
template <int DPF> struct HeadGeo {
    static constexpr int CR = 2 * DPF + 2;
    static constexpr int OFF_P = 0, OFF_C = OFF_P + YPH * P_ROWB, OFF_Y = OFF_C + CR * C_ROWB, OFF_F = OFF_Y + YPH * Y_ROWB;
    static constexpr int OFF_B = OFF_F + YPH * Y_ROWB;       // SFT_layer1's head tiles
    static constexpr int SMEM = OFF_B + SFT_TILE_B;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(BIG % CR == 0, "BIG");
};

template <int DPF, bool FQ>
__global__ __launch_bounds__(512) void le_head_rows_kernel(RowsHeadParams p)
{
    using G = HeadGeo<DPF>;
    const FqParam *const fqs = (FQ && (p.fq & 8)) ? p.fq_s : nullptr;
    constexpr int CR = G::CR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned sm = lds_off(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS, hx0 = x0 >> 1;
    const int y0 = seg * p.rows_per_seg, y1 = min(y0 + p.rows_per_seg, p.H);      // rows_per_seg is even
    const int ya = y0 - 2, hya = ya >> 1;
    const int nsteps = (y1 - ya + 1) / 2 + 3;
    const int H = p.H, W = p.W, W1 = (W + 1) >> 1;
    sft_tiles_to_lds(smem + G::OFF_B, p.sft_bias, tid);
    const bool colfull = x0 >= 2 && x0 + 62 <= W;
    const int g = wave & 3, gr = g >> 1, gh = g & 1;
    const int cx = 32 * gh + l31;
    unsigned va[3][2];
    frag_addr<LStd>(va, sm, cx, lh);                                   // HR_conv1's fragments of the Y ring

    if (wave < 4) {
        // ------------------------------------------------------------------ role H1
        f16x8 c3w[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) c3w[ky] = reinterpret_cast<const f16x8 *>(p.c3_wfrag)[ky * 64 + lane];
        SftW s1;
        load_sft(s1, p.sft_wfrag, p.sft_bias, lane, lh);
        const unsigned t1 = sm + G::OFF_B + 64 * lh;
        Bank wd;
        load_bank(wd, p.w_down, l31, lh);
        f32x4 bd[4];
        load_bias(bd, p.b_down, lh);
        unsigned vq[2];                                                // Y write
        chunk_addr<LStd>(vq, sm, cx, lh);
        const unsigned vc = sm + LCond::at(cx, lh);
        const bool col = (unsigned)(x0 - 2 + cx) < (unsigned)W;
        const unsigned vp = sm + (cx + 2 * lh) * 8;                    // patch pixels cx + 2 lh, + 1 of a kernel row = K slots 8 lh .. 8 lh + 7
        // patch staging: thread t < 136 owns pixel (t / 68, t % 68) of the two new rows
        const int pr = tid / 68, pc = tid - pr * 68;
        const bool pth = tid < 136, pcol = pth && (unsigned)(x0 - 3 + pc) < (unsigned)W;
        const size_t plane = (size_t)H * W;
        const f16 *pimg = p.img + (x0 - 3 + pc);
        f16 pv[3] = {(f16)0.f, (f16)0.f, (f16)0.f};
        bool pin_img = false;
        auto patch_fetch = [&](int r0) __attribute__((always_inline)) {      // image rows r0, r0 + 1
            const int r = r0 + pr;
            const bool ok = pcol && (unsigned)r < (unsigned)H && r <= y1 + 1;
            pin_img = ok;
            const f16 *src = ok ? pimg + (size_t)r * W : p.img;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                pv[ch] = src[ch * plane];
                if (!ok) pv[ch] = (f16)0.f;
            }
        };
        auto patch_stage = [&](int m0) __attribute__((always_inline)) {       // ring slots m0 (this thread's pr == 0) and m0 + 1
            if (pth) {
                int ms = m0 + pr;
                if (ms >= YN) ms -= YN;
                const unsigned a = sm + G::OFF_P + ms * P_ROWB + pc * 8;
                f16x4 v = f16x4{pv[0], pv[1], pv[2], (f16)0.f};
                if (FQ && (p.fq & 1) && pin_img) v = fq4(v, p.fq_img);          // conv_first is W8A8: its input quantiser (zero padding stays zero)
                v[3] = (f16)1.f;                                                // 1: the bias slot
                lds_wr(a, v);
                if (ms < 2) lds_wr(a + YN * P_ROWB, v);                         // the second copy of a lap's rows 0 and 1
            }
        };
        // down_conv1: half-resolution pixel l31 (column hx0 + l31, 30 used) reads F slots 2 l31 + kx (slot c = image column x0 - 1 + c)
        unsigned vd[3][2];
        frag_addr<LHeadF>(vd, sm, l31, lh, 2);
        const bool dcol = l31 < WS / 2 && hx0 + l31 < W1;
        f16 *const d1 = p.fea1 + (size_t)(hx0 + l31) * 32 + 4 * lh;
        for (int e = tid; e < YPH * 4; e += 256)                        // the four pad slots of every patch row stay finite (zero weights read them)
            lds_wr(sm + G::OFF_P + (e >> 2) * P_ROWB + (68 + (e & 3)) * 8, zero4());
        patch_fetch(ya - 1); patch_stage(YN - 1);                      // ring rows -1, 0
        patch_fetch(ya + 1); patch_stage(1);                           // ring rows 1, 2
        // ring rows of step s: Y row ra = 2 s + gr from patch rows ra - 1 .. ra + 1 and cond row ra; staging of patch rows 2 s + 3, + 4;
        // down_conv1 on F rows 2 s - 7 .. 2 s - 5
        Cur<G::OFF_C, CR, C_ROWB> ca(gr);
        Cur<G::OFF_P, YN, P_ROWB> pw(gr - 1);
        Cur<G::OFF_Y, YN, Y_ROWB> yw(gr);
        Cur<G::OFF_F, YN, Y_ROWB> wf(-7);
        int ps = 3;                                                    // staging slot of ring row 2 s + 3 (3, 5, 1, ..)
        int ra_img = ya + gr;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            if (RB_ABL & 32) { __builtin_amdgcn_s_barrier(); continue; }
            patch_fetch(ya + 2 * s + 3);
            const f16x8 c1 = lds_rd<f16x8>(vc + ca.o);
            f32x16 h;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const f16x4 u = lds_rd<f16x4>(vp + pw.o + ky * P_ROWB), v = lds_rd<f16x4>(vp + pw.o + ky * P_ROWB + 8);
                const f16x8 xf = __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7);
                h = __builtin_amdgcn_mfma_f32_32x32x16_f16(c3w[ky], xf, ky == 0 ? zero16() : h, 0, 0, 0);
            }
            f32x16 sc1, sh1;
            sft_heads(s1, sft_hidden(s1, c1, fqs, l31), t1, sc1, sh1, fqs);
            f16x4 y[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) y[qd] = __builtin_elementwise_max(cvt4(h[4 * qd], h[4 * qd + 1], h[4 * qd + 2], h[4 * qd + 3]), zero4());
            sft_modulate(sc1, sh1, y);
            if (FQ && (p.fq & 2)) fq_row(y, p.fq_y);                      // HR_conv1 is W8A8: its input quantiser
            const bool row = (unsigned)ra_img < (unsigned)H;
            if (!(colfull && row)) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) if (!(col && row)) y[qd] = zero4();
            }
            put_row(vq, yw.o, yw.mirrored(), y);
            STAMP(1);
            if (wave == (s & 3)) {
                const int hr = hya + s - 3;                            // down_conv1 on half-resolution row s - 3
                const f32x16 acc = conv18<6, Y_ROWB, ROWS_PIN_F, (ROWS_PRIO_MASK >> 4) & 1>(wd, vd, wf.o, [](int) {});
                if (dcol && hr >= (y0 >> 1) && hr < ((y1 + 1) >> 1)) {
                    f16 *d = d1 + (size_t)hr * W1 * 32;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<f16x4 *>(d + 8 * qd) = __builtin_elementwise_max(bias_cvt4(acc, qd, bd[qd]), zero4());
                }
            }
            STAMP(2);
            patch_stage(ps);                                           // rows 2 s + 3, 2 s + 4 (fetched at the top of the step)
            ca.step(); pw.step(); yw.step(); wf.step();
            ps = ps + 2 >= YN ? ps + 2 - YN : ps + 2;
            ra_img += 2;
            STAMP(3);
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    } else {
        // ------------------------------------------------------------------ role H2
        Bank wh;
        load_bank(wh, p.w_hr, l31, lh);
        f32x4 bh[4];
        load_bias(bh, p.b_hr, lh);
        const bool colf = (unsigned)(x0 - 1 + cx) < (unsigned)W;       // F slot cx = image column x0 - 1 + cx
        const dma_rsrc_t rc = dma_rsrc(p.cond);
        const unsigned cl = LCond::src_off(gh, lane);
        const bool cok = (unsigned)(x0 - 2 + LCond::src_px(gh, lane)) < (unsigned)W;
        auto issue_c = [&](int r, int co) __attribute__((always_inline)) {
            const bool rok = (unsigned)r < (unsigned)H && r <= y1;
            dma16_at(rc, sm + co + gh * 1024, (rok && cok) ? (unsigned)((r * W + x0 - 2) * 32) + cl : DMA_OOB);
        };
        unsigned vq[2];                                                // F write
        chunk_addr<LHeadF>(vq, sm, cx, lh);
        char *trash = p.trash + tid * 16;
        // fea0 leaves straight from the registers, as this lane's two chunks of its pixel: F slot cx = image column x0 - 1 + cx
        const int ox = cx - 1;
        const bool ocol = ox >= 0 && ox < WS && x0 + ox < W;
        f16 *const dst0 = p.fea0 + ((ptrdiff_t)x0 + ox) * 32 + 8 * lh;
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq) issue_c(ya + 2 * sq + gr, G::OFF_C + ((2 * sq + gr + BIG) % CR) * C_ROWB);
        Cur<G::OFF_C, CR, C_ROWB> cd(2 * DPF + gr);
        Cur<G::OFF_Y, YN, Y_ROWB> wy(gr - 4);
        Cur<G::OFF_F, YN, Y_ROWB> fw(gr - 3);
        int rb_img = ya + gr - 3;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            issue_c(rb_img + 3 + 2 * DPF, cd.o);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            if (RB_ABL & 64) { __builtin_amdgcn_s_waitcnt(waitcnt_imm(DPF - 1, 0)); __builtin_amdgcn_s_barrier(); cd.step(); rb_img += 2; continue; }
            const f32x16 acc = conv18<ROWS_AHEAD_F, Y_ROWB, ROWS_PIN_F, (ROWS_PRIO_MASK >> 5) & 1>(wh, va, wy.o, [](int) {});
            STAMP(1);
            f16x4 z[4];
            const bool in = colf && (unsigned)rb_img < (unsigned)H;        // outside the image: down_conv1's zero padding
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) z[qd] = __builtin_elementwise_max(bias_cvt4(acc, qd, bh[qd]), zero4());
            f16x8 z0, z1;
            quads_to_chunks(z, z0, z1);
            {   // fea0 as the tail's skip wants it: not quantised
                f16 *d = (ocol && rb_img >= y0 && rb_img < y1) ? dst0 + (size_t)rb_img * W * 32 : reinterpret_cast<f16 *>(trash);
                *reinterpret_cast<f16x8 *>(d) = z0;
                *reinterpret_cast<f16x8 *>(d == reinterpret_cast<f16 *>(trash) ? d : d + 16) = z1;
            }
            if (FQ && (p.fq & 4)) { z0 = fq8(z0, p.fq_f); z1 = fq8(z1, p.fq_f); }     // down_conv1 is W8A8: its input quantiser
            if (!in) { z0 = f16x8{}; z1 = f16x8{}; }
            lds_wr(vq[0] + (unsigned)fw.o, z0); lds_wr(vq[1] + (unsigned)fw.o, z1);
            if (fw.mirrored()) { lds_wr(vq[0] + (unsigned)(fw.o + YN * Y_ROWB), z0); lds_wr(vq[1] + (unsigned)(fw.o + YN * Y_ROWB), z1); }
            cd.step(); wy.step(); fw.step();
            rb_img += 2;
            STAMP(3);
            // per step and wave: one DMA piece, then (behind the conv) two stores: the piece step s + 1 reads is older than
            // 3 (DPF - 1) + 2 operations
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(3 * (DPF - 1) + 2, 0));
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    }
}

}  // namespace

hipError_t le_rb_rows_launch(RowsRbParams p, int n_cu, hipStream_t s)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash) return hipErrorInvalidValue;
    static DevOnce once;
    static DevOnce once_q;
    int nseg;
    strips(p, n_cu, false, nseg);
    if (p.fq) {
        if (hipError_t e = set_lds(le_rb_rows_kernel<3, true>, RbGeo<3>::SMEM, once_q)) return e;
        hipLaunchKernelGGL((le_rb_rows_kernel<3, true>), dim3(p.nstrips * nseg), dim3(512), RbGeo<3>::SMEM, s, p);
    } else {
        if (hipError_t e = set_lds(le_rb_rows_kernel<3, false>, RbGeo<3>::SMEM, once)) return e;
        hipLaunchKernelGGL((le_rb_rows_kernel<3, false>), dim3(p.nstrips * nseg), dim3(512), RbGeo<3>::SMEM, s, p);
    }
    return hipGetLastError();
}

// H, W even; u is [H/2][W/2][32]
hipError_t le_tail_rows_launch(RowsTailParams p, int n_cu, hipStream_t s)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash || (p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;
    static DevOnce once;
    static DevOnce once_q;
    int nseg;
    strips(p, n_cu, true, nseg);
    if (p.fq) {
        if (hipError_t e = set_lds(le_tail_rows_kernel<3, true>, TailGeo<3>::SMEM, once_q)) return e;
        hipLaunchKernelGGL((le_tail_rows_kernel<3, true>), dim3(p.nstrips * nseg), dim3(512), TailGeo<3>::SMEM, s, p);
    } else {
        if (hipError_t e = set_lds(le_tail_rows_kernel<3, false>, TailGeo<3>::SMEM, once)) return e;
        hipLaunchKernelGGL((le_tail_rows_kernel<3, false>), dim3(p.nstrips * nseg), dim3(512), TailGeo<3>::SMEM, s, p);
    }
    return hipGetLastError();
}

// H, W even (strips start on even columns: the half-resolution map is cut at x0 / 2); fea1 is [H/2][W/2][32]
hipError_t le_head_rows_launch(RowsHeadParams p, int n_cu, hipStream_t s)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash || (p.W & 1) || (p.H & 1)) return hipErrorInvalidValue;
    static DevOnce once;
    static DevOnce once_q;
    int nseg;
    strips(p, n_cu, true, nseg);
    if (p.fq) {
        if (hipError_t e = set_lds(le_head_rows_kernel<3, true>, HeadGeo<3>::SMEM, once_q)) return e;
        hipLaunchKernelGGL((le_head_rows_kernel<3, true>), dim3(p.nstrips * nseg), dim3(512), HeadGeo<3>::SMEM, s, p);
    } else {
        if (hipError_t e = set_lds(le_head_rows_kernel<3, false>, HeadGeo<3>::SMEM, once)) return e;
        hipLaunchKernelGGL((le_head_rows_kernel<3, false>), dim3(p.nstrips * nseg), dim3(512), HeadGeo<3>::SMEM, s, p);
    }
    return hipGetLastError();
}
