// le_rows_i8.hip -- the row-streaming fused LE kernels (le_rows.hip) for chains whose layers are ALL W8A8, on int8 MFMA (gfx950).
//
//   le_rb_rows_i8   ResBlock_with_SFT.forward (arch_util.py:89-95) with conv1, conv2 and the eight 1x1 convs of its two SFT layers as
//                   W8A8Conv2d (hdrtvnet_torch.py:296-364): the full-QAT recipe's ResBlocks at 1/2 and 1/4 resolution
//   le_tail_rows_i8 HDRUNet3T1_arch.py:196-206 with up_conv3, SFT_layer2, HR_conv2 and conv_last W8A8
//   le_head_rows_i8 HDRUNet3T1_arch.py:168-172 with conv_first, SFT_layer1, HR_conv1 and down_conv1 W8A8
//
// Same schedule as le_rb_rows (strips of 60 columns x row segments, two rows per step, stages skewed across steps, one barrier
// per step, LDS-DMA three steps ahead; le_rows.hip's header), but what lives in the rings between the stages are the layers' int8
// CODES, not f16 values:
//   * a stage's epilogue applies the NEXT conv's activation quantiser once per element (one FMA + v_cvt_pk_u8_f32 per value,
//     common.h quant4) and writes the lane's 16 codes of its pixel as ONE 16-byte LDS store: a code pixel is 32 bytes, its two
//     halves hold the channels of the two lane halves in accumulator order (8 qd + 4 lh + k at byte 16 lh + 4 qd + k) -- the K order
//     pack_conv32_i8 gives the weights -- so there is no cross-lane exchange and no 8-byte access at all;
//   * a 3x3 conv is 9 v_mfma_i32_32x32x32_i8 (one tap of 32 channels = one K step) on 9 fragment reads: half the MFMAs, half the
//     LDS reads and half the ring bytes of the f16 form; the filter bank is 36 VGPRs instead of 72;
//   * the reference pads with zeros AFTER dequantisation: out-of-image ring slots hold code 0 and the conv's epilogue picks the
//     shift of the output pixel's border class (16 classes x 32 channels, pack_conv32_i8 / q_tables);
//   * the SFT layers' four 1x1 convs run as conv32s's SQ pass does: the condition pixel quantised per branch, one block-diagonal
//     K = 32 MFMA for both hidden layers, dequantise + LeakyReLU + re-quantise in registers, two MFMAs for the heads, constants
//     from LDS.
// Arithmetic, operand order and every rounding point are those of the per-layer int8 kernels conv32s<sft-i8, i8> (variant
// le_rows_i8 = 0): outputs are bit-identical to them (tests/test_gpu_le_rows.py), so the per-layer path's parity evidence against
// the reference's fake-quant arithmetic carries over, and the result no longer depends on whether a map is large enough for the
// fused kernels.
#include "le_rows.h"

namespace {

constexpr int Y8_ROWB = YP * 32;                    // a code ring row: 66 slots x 32 bytes
using L8Std = Lay32<0, 12>;                         // R1 + W1 for 32-byte pixels (tools/lds_ring_layouts.py)

struct Bank8 { i32x4 f[9]; };                       // a 3x3 32 -> 32 int8 filter bank: tap t, lane (n = l31, half lh) = bytes 16 lh .. of row n
__device__ __forceinline__ void load_bank8(Bank8 &b, const int8_t *wpk8, int l31, int lh, int coutp = 32, int n0 = 0)
{
#pragma unroll
    for (int t = 0; t < 9; ++t) b.f[t] = *reinterpret_cast<const i32x4 *>(wpk8 + (t * coutp + n0 + l31) * 32 + 16 * lh);
}
__device__ __forceinline__ i32x16 izero16() { return i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; }

// 3x3 conv of one 32-pixel group out of a mirrored code ring: va[kx] = this lane's fragment address for kernel column kx in ring
// row 0 (buffer base included), `win` the byte offset of the window's first row; reads run AHEAD taps in front of the MFMAs
template <int AHEAD, int ROWB = Y8_ROWB, class Hook>
__device__ __forceinline__ i32x16 conv9(const Bank8 &w, const unsigned (&va)[3], int win, Hook hook)
{
    i32x4 x[9];
    i32x16 acc;
    auto ld = [&](int t) __attribute__((always_inline)) { x[t] = lds_rd<i32x4>(va[t % 3] + (unsigned)(win + (t / 3) * ROWB)); };
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) ld(t);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t + AHEAD < 9) ld(t + AHEAD);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w.f[t], x[t], t == 0 ? izero16() : acc, 0, 0, 0);
        hook(t);
    }
    return acc;
}

// An SFT layer's W8A8 operands (pack_sft, SftLayer.q*): fragments in registers, the 192 constants in LDS at `kb` (this lane half's
// 16-float rows start at kb + 64 lh: ka, kb, k2, k3, k4, k5 at byte offsets 0, 128, .. 640)
struct Sft8 { i32x4 a0, a1s, a1t; float inv, zoff, hz0, hz1; unsigned kb; };
__device__ __forceinline__ void load_sft8(Sft8 &s, const RowsSftI8 &p, unsigned k_lds, int lane, int lh)
{
    const i32x4 *fr = reinterpret_cast<const i32x4 *>(p.wfrag);
    s.a0 = fr[lane]; s.a1s = fr[64 + lane]; s.a1t = fr[128 + lane];
    s.inv = p.inv[lh]; s.zoff = p.zoff[lh]; s.hz0 = p.hzoff[0]; s.hz1 = p.hzoff[1];
    s.kb = k_lds + 64 * lh;
}
// conv32s.hip's SQ pass, step by step (same expressions: the results are its bits)
__device__ __forceinline__ i32x16 sft8_hidden(const Sft8 &s, const f16x8 &c0, const f16x8 &c1)
{
    i32x4 cb;
    cb[0] = (int)quant4((float)c0[0], (float)c0[1], (float)c0[2], (float)c0[3], s.inv, s.zoff);
    cb[1] = (int)quant4((float)c0[4], (float)c0[5], (float)c0[6], (float)c0[7], s.inv, s.zoff);
    cb[2] = (int)quant4((float)c1[0], (float)c1[1], (float)c1[2], (float)c1[3], s.inv, s.zoff);
    cb[3] = (int)quant4((float)c1[4], (float)c1[5], (float)c1[6], (float)c1[7], s.inv, s.zoff);
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(s.a0, cb, izero16(), 0, 0, 0);
}
__device__ __forceinline__ void sft8_mid(const Sft8 &s, const i32x16 &hacc, i32x4 &hs, i32x4 &ht)
{
    float t[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 ka = lds_rd<f32x4>(s.kb + 16 * g), kb = lds_rd<f32x4>(s.kb + 128 + 16 * g);
        const float zo = (g >> 1) ? s.hz1 : s.hz0;
        const float u0 = (float)hacc[4 * g + 0] * ka[0] + kb[0], u1 = (float)hacc[4 * g + 1] * ka[1] + kb[1],
                    u2 = (float)hacc[4 * g + 2] * ka[2] + kb[2], u3 = (float)hacc[4 * g + 3] * ka[3] + kb[3];
        t[4 * g + 0] = fmaxf(u0, 0.1f * u0) + zo; t[4 * g + 1] = fmaxf(u1, 0.1f * u1) + zo;
        t[4 * g + 2] = fmaxf(u2, 0.1f * u2) + zo; t[4 * g + 3] = fmaxf(u3, 0.1f * u3) + zo;
    }
    hs = i32x4{(int)quant4u(t[0], t[1], t[2], t[3]), (int)quant4u(t[4], t[5], t[6], t[7]), 0, 0};
    ht = i32x4{(int)quant4u(t[8], t[9], t[10], t[11]), (int)quant4u(t[12], t[13], t[14], t[15]), 0, 0};
}
__device__ __forceinline__ void sft8_heads(const Sft8 &s, const i32x4 &hs, const i32x4 &ht, f16x4 (&s1p)[4], f16x4 (&s0p)[4])
{
    const i32x16 a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(s.a1s, hs, izero16(), 0, 0, 0);
    const i32x16 a2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(s.a1t, ht, izero16(), 0, 0, 0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 k2 = lds_rd<f32x4>(s.kb + 256 + 16 * g), k3 = lds_rd<f32x4>(s.kb + 384 + 16 * g);
        const f32x4 k4 = lds_rd<f32x4>(s.kb + 512 + 16 * g), k5 = lds_rd<f32x4>(s.kb + 640 + 16 * g);
        const float sc0 = (float)a1[4 * g + 0] * k2[0] + k3[0], sc1 = (float)a1[4 * g + 1] * k2[1] + k3[1],
                    sc2 = (float)a1[4 * g + 2] * k2[2] + k3[2], sc3 = (float)a1[4 * g + 3] * k2[3] + k3[3];
        const float sh0 = (float)a2[4 * g + 0] * k4[0] + k5[0], sh1 = (float)a2[4 * g + 1] * k4[1] + k5[1],
                    sh2 = (float)a2[4 * g + 2] * k4[2] + k5[2], sh3 = (float)a2[4 * g + 3] * k4[3] + k5[3];
        s1p[g] = cvt4(sc0, sc1, sc2, sc3);
        s0p[g] = cvt4(sh0, sh1, sh2, sh3);
    }
}
// y = x * (scale + 1) + shift in packed f16, then the reading conv's quantiser: this lane's 16 codes of its pixel (0 outside the image)
__device__ __forceinline__ i32x4 modulate_quant(const f16x4 (&x)[4], const f16x4 (&s1p)[4], const f16x4 (&s0p)[4], float q_inv, float q_zoff, bool inimg)
{
    i32x4 codes;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const f16x4 y = x[qd] * s1p[qd] + s0p[qd];
        const unsigned w = quant4((float)y[0], (float)y[1], (float)y[2], (float)y[3], q_inv, q_zoff);
        codes[qd] = inimg ? (int)w : 0;
    }
    return codes;
}
// (LAP: bytes from a ring row to its second copy = the ring's logical size)
template <int LAP = YN * Y8_ROWB>
__device__ __forceinline__ void put_codes(unsigned a, int off, bool mirror, const i32x4 &codes)
{
    lds_wr(a + (unsigned)off, codes);
    if (mirror) lds_wr(a + (unsigned)(off + LAP), codes);
}
// conv epilogue of conv32s<.., i8>: o = f16(act(acc * scale[c] + shift[class][c])) on this lane's 16 channels
// (shift_tab: LDS address of class 0's row for this wave's 32 channels; CLSB: bytes from one class's row to the next)
template <int CLSB = 128>
__device__ __forceinline__ void dequant_act(const i32x16 &iacc, const f32x4 (&scq)[4], unsigned shift_tab, int bcls, int lh, float slope, f16x4 (&o)[4])
{
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const f32x4 sh = lds_rd<f32x4>(shift_tab + (unsigned)(bcls * CLSB + 32 * qd + 16 * lh));
        const float v0 = (float)iacc[4 * qd + 0] * scq[qd][0] + sh[0], v1 = (float)iacc[4 * qd + 1] * scq[qd][1] + sh[1],
                    v2 = (float)iacc[4 * qd + 2] * scq[qd][2] + sh[2], v3 = (float)iacc[4 * qd + 3] * scq[qd][3] + sh[3];
        o[qd] = cvt_h4(act_fast(v0, slope), act_fast(v1, slope), act_fast(v2, slope), act_fast(v3, slope));      // packed converts: common.h
    }
}
__device__ __forceinline__ void table_to_lds(char *dst, const RowsConvI8 &c, int tid, int coutp = 32)       // [scale coutp][shift 16 x coutp] floats
{
    for (int e = tid; e < 17 * coutp; e += 512) reinterpret_cast<float *>(dst)[e] = e < coutp ? c.scale[e] : c.shift[e - coutp];
}

// ------------------------------------------------------------------------------------------------------------------------
template <int DPF> struct Rb8Geo {
    static constexpr int LAG = 6;
    static constexpr int XR = 2 * DPF + LAG + 2, CR = 2 * DPF + LAG;
    static constexpr int OFF_X = 0, OFF_C = OFF_X + XR * X_ROWB, OFF_Y1 = OFF_C + CR * C_ROWB, OFF_Y2 = OFF_Y1 + YPH * Y8_ROWB;
    static constexpr int OFF_T = OFF_Y2 + YPH * Y8_ROWB;       // conv1's table (2176 B), then conv2's
    static constexpr int OFF_K = OFF_T + 2 * 2176;             // sft2's constants (768 B), then sft1's
    static constexpr int SMEM = OFF_K + 2 * 768;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(BIG % XR == 0 && BIG % CR == 0 && BIG % YN == 0, "BIG");
};

template <int DPF>
__global__ __launch_bounds__(512) void le_rb_rows_i8_kernel(RowsRbI8Params p)
{
    using G = Rb8Geo<DPF>;
    constexpr int LAG = G::LAG, XR = G::XR, CR = G::CR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned sm = lds_off(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS;
    const int y0 = seg * p.rows_per_seg, yend = min(y0 + p.rows_per_seg, p.H);
    const int ya = y0 - 2;                                             // image row of ring row 0
    const int nsteps = (yend - ya + LAG - 1) / 2 + 1;
    const int H = p.H, W = p.W;
    table_to_lds(smem + G::OFF_T, p.c1, tid);
    table_to_lds(smem + G::OFF_T + 2176, p.c2, tid);
    for (int e = tid; e < 384; e += 512) reinterpret_cast<float *>(smem + G::OFF_K)[e] = e < 192 ? p.s2.konst[e] : p.s1.konst[e - 192];

    const int g = wave & 3, gr = g >> 1, gh = g & 1;                   // this wave's 32-pixel group: row gr of the step's pair, column half gh
    const int cx = 32 * gh + l31;                                      // this lane's pixel slot in its group's ring rows
    unsigned va[3];                                                    // conv fragments: output slot cx reads code slots cx .. cx + 2, half lh
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) va[kx] = sm + L8Std::at(cx + kx, lh);
    const unsigned vw = sm + L8Std::at(cx, lh);                        // code write: this lane's half of slot cx

    if (wave < 4) {
        // ------------------------------------------------------------------ role B: conv1 (int8) -> ReLU -> sft2 -> conv2's codes
        Bank8 w1;
        load_bank8(w1, p.c1.wpk8, l31, lh);
        Sft8 s2;
        load_sft8(s2, p.s2, sm + G::OFF_K, lane, lh);
        f32x4 scq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) scq[qd] = *reinterpret_cast<const f32x4 *>(p.c1.scale + 8 * qd + 4 * lh);
        const unsigned tab = sm + G::OFF_T + 128;                      // conv1's shift classes
        const unsigned vc0 = sm + LCond::at(cx + 1, 0), vc1 = sm + LCond::at(cx + 1, 1);      // Y2 slot cx = image column x0 - 1 + cx = condition slot cx + 1
        const int ox = x0 - 1 + cx;
        const bool col2 = (unsigned)ox < (unsigned)W;
        const int ccls = (ox == 0 ? 1 : 0) | (ox == W - 1 ? 2 : 0);
        const float q_inv = p.c2.q_inv, q_zoff = p.c2.q_zoff, slope = p.slope1;
        // ring rows of step s: conv1 + sft2 on rb = 2 s - 3 + gr
        Cur<G::OFF_C, CR, C_ROWB> cb(gr - 3);
        Cur<G::OFF_Y1, YN, Y8_ROWB> wn(gr - 4);
        Cur<G::OFF_Y2, YN, Y8_ROWB> yb2(gr - 3);
        int rb_img = ya + gr - 3;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nsteps; ++s) {
            const f16x8 c0 = lds_rd<f16x8>(vc0 + cb.o), c1 = lds_rd<f16x8>(vc1 + cb.o);
            const bool in2 = col2 && (unsigned)rb_img < (unsigned)H;   // outside the image: conv2's zero padding = code 0
            const int bcls = (((rb_img == 0 ? 1 : 0) | (rb_img == H - 1 ? 2 : 0)) << 2) | ccls;
            i32x16 hacc;
            i32x4 hs, ht;
            f16x4 s1p[4], s0p[4];
            const i32x16 iacc = conv9<4>(w1, va, wn.o, [&](int t) __attribute__((always_inline)) {
                if (t == 0) hacc = sft8_hidden(s2, c0, c1);
                if (t == 3) sft8_mid(s2, hacc, hs, ht);
                if (t == 6) sft8_heads(s2, hs, ht, s1p, s0p);
            });
            f16x4 o[4];
            dequant_act(iacc, scq, tab, bcls, lh, slope, o);
            put_codes(vw, yb2.o, yb2.mirrored(), modulate_quant(o, s1p, s0p, q_inv, q_zoff, in2));
            cb.step(); wn.step(); yb2.step();
            rb_img += 2;
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // ------------------------------------------------------------------ role C: the DMA, sft1 -> conv1's codes, conv2 (int8) + x, stores
        Bank8 w2;
        load_bank8(w2, p.c2.wpk8, l31, lh);
        Sft8 s1;
        load_sft8(s1, p.s1, sm + G::OFF_K + 768, lane, lh);
        f32x4 scq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) scq[qd] = *reinterpret_cast<const f32x4 *>(p.c2.scale + 8 * qd + 4 * lh);
        const unsigned tab = sm + G::OFF_T + 2176 + 128;               // conv2's shift classes
        unsigned vq[2], vr[2];                                         // x of slot cx (sft1's input) and of slot cx + 2 (the residual): this lane's two chunks
        chunk_addr<LStd>(vq, sm, cx, lh);
        chunk_addr<LStd>(vr, sm, cx + 2, lh);
        const unsigned vc0 = sm + LCond::at(cx, 0), vc1 = sm + LCond::at(cx, 1);             // slot cx = image column x0 - 2 + cx
        const bool col1 = (unsigned)(x0 - 2 + cx) < (unsigned)W;
        const int ox = x0 + cx;
        const int ccls = (ox == 0 ? 1 : 0) | (ox == W - 1 ? 2 : 0);
        const float q_inv = p.c1.q_inv, q_zoff = p.c1.q_zoff;
        const dma_rsrc_t rx = dma_rsrc(p.x), rc = dma_rsrc(p.cond);
        // per step: pieces 2 gh, 2 gh + 1 of x row gr (16 pixels x 64 B each) and piece gh of condition row gr (32 pixels x 32 B)
        const unsigned xl0 = LStd::src_off(2 * gh, lane), xl1 = LStd::src_off(2 * gh + 1, lane), cl = LCond::src_off(gh, lane);
        const bool xok0 = (unsigned)(x0 - 2 + LStd::src_px(2 * gh, lane)) < (unsigned)W, xok1 = (unsigned)(x0 - 2 + LStd::src_px(2 * gh + 1, lane)) < (unsigned)W;
        const bool cok = (unsigned)(x0 - 2 + LCond::src_px(gh, lane)) < (unsigned)W;
        auto issue = [&](int r, int xo_, int co_) __attribute__((always_inline)) {        // image row r into the ring rows at xo_ / co_
            const bool rok = (unsigned)r < (unsigned)H && r <= yend + 1;
            const unsigned pix = (unsigned)(r * W + x0 - 2);
            dma16_at(rx, sm + xo_ + (2 * gh) * 1024, (rok && xok0) ? pix * 64u + xl0 : DMA_OOB);
            dma16_at(rx, sm + xo_ + (2 * gh + 1) * 1024, (rok && xok1) ? pix * 64u + xl1 : DMA_OOB);
            dma16_at(rc, sm + co_ + gh * 1024, (rok && cok) ? pix * 32u + cl : DMA_OOB);
        };
        char *trash = p.trash + tid * 16;
        const bool ocol = cx < WS && x0 + cx < W;
        f16 *const dst0 = p.dst + (size_t)(x0 + cx) * 32 + 8 * lh;
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq)
            issue(ya + 2 * sq + gr, G::OFF_X + ((2 * sq + gr + BIG) % XR) * X_ROWB, G::OFF_C + ((2 * sq + gr + BIG) % CR) * C_ROWB);
        // ring rows of step s: DMA into 2 (s + DPF) + gr; sft1 on ra = 2 s + gr; conv2 + residual on ro = ra - LAG
        Cur<G::OFF_X, XR, X_ROWB> xd(2 * DPF + gr), xa(gr), xres(gr - LAG);
        Cur<G::OFF_C, CR, C_ROWB> cd(2 * DPF + gr), ca(gr);
        Cur<G::OFF_Y1, YN, Y8_ROWB> ya1(gr);
        Cur<G::OFF_Y2, YN, Y8_ROWB> wn(gr - LAG - 1);
        int ro_img = ya + gr - LAG;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nsteps; ++s) {
            issue(ro_img + LAG + 2 * DPF, xd.o, cd.o);
            __builtin_amdgcn_sched_barrier(0);
            const f16x8 c0 = lds_rd<f16x8>(vc0 + ca.o), c1 = lds_rd<f16x8>(vc1 + ca.o);
            f16x4 xq[4];
            get_row(vq, xa.o, xq);
            const f16x8 res0 = lds_rd<f16x8>(vr[0] + (unsigned)xres.o), res1 = lds_rd<f16x8>(vr[1] + (unsigned)xres.o);
            const bool in1 = col1 && (unsigned)(ro_img + LAG) < (unsigned)H;   // outside the image: conv1's zero padding = code 0
            const int bcls = (((ro_img == 0 ? 1 : 0) | (ro_img == H - 1 ? 2 : 0)) << 2) | ccls;
            // conv2 on row ro with row ra's whole SFT pass (independent of it) between its MFMAs
            i32x16 hacc;
            i32x4 hs, ht;
            f16x4 s1p[4], s0p[4];
            const i32x16 iacc = conv9<4>(w2, va, wn.o, [&](int t) __attribute__((always_inline)) {
                if (t == 0) hacc = sft8_hidden(s1, c0, c1);
                if (t == 3) sft8_mid(s1, hacc, hs, ht);
                if (t == 6) sft8_heads(s1, hs, ht, s1p, s0p);
            });
            put_codes(vw, ya1.o, ya1.mirrored(), modulate_quant(xq, s1p, s0p, q_inv, q_zoff, in1));
            {
                f16x4 o[4];
                dequant_act(iacc, scq, tab, bcls, lh, 1.f, o);         // no activation behind conv2 (act_fast(v, 1) = v)
                f16x8 o0, o1;
                quads_to_chunks(o, o0, o1);
                o0 += res0; o1 += res1;                                // x + conv2(..): one f16 rounding per element, as conv32s's epilogue
                f16 *d = (ocol && ro_img >= y0 && ro_img < yend) ? dst0 + (size_t)ro_img * W * 32 : reinterpret_cast<f16 *>(trash);
                *reinterpret_cast<f16x8 *>(d) = o0;
                *reinterpret_cast<f16x8 *>(d == reinterpret_cast<f16 *>(trash) ? d : d + 16) = o1;
            }
            xd.step(); xa.step(); xres.step(); cd.step(); ca.step(); ya1.step(); wn.step();
            ro_img += 2;
            // per step and wave: three DMA pieces, then two stores, all always issued: the pieces that step s + 1 reads were issued
            // at the top of step s + 1 - DPF, in front of 5 (DPF - 1) + 2 younger operations
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(5 * (DPF - 1) + 2, 0));
            __builtin_amdgcn_s_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The full-resolution tail (le_tail_rows) with up_conv3, SFT_layer2, HR_conv2 and conv_last all W8A8.  Per-layer form: conv32p<4,
// plain, i8> (quantise u, up-conv, ReLU, PixelShuffle, + fea0), conv32s<sft-i8, i8> (SFT_layer2, quantise, HR_conv2, ReLU) and
// conv32s<plain, i8, planar> (quantise, conv_last, + the AGCM residual).  Here:
//     step s:  role T1 (waves 0-3; wave b = PixelShuffle phase (b >> 1, b & 1)): LDS-DMA of fea0 / cond rows 2s+6, 2s+7, of the f16 u
//                   row s+5 and of the residual planes; the u row s+2 (landed: issued five steps ago) quantised ONCE into the u code
//                   ring (9 of its 34 pixels per wave); up_conv3 bank b on u code rows s-1 .. s+1 -> ReLU, + fea0, SFT_layer2 (its
//                   int8 MLPs inside the conv's MFMA stream), HR_conv2's quantiser -> Y code rows 2s, 2s+1
//              role T2 (waves 4-7, group (g >> 1, g & 1)): HR_conv2 + ReLU + conv_last's quantiser -> Z code rows 2s-3, 2s-2;
//                   conv_last + residual -> output rows 2s-6, 2s-5 (planar)
constexpr int U8_SLOTS = 48, U16_ROWB = U8_SLOTS * 64, U8_ROWB = U8_SLOTS * 32, UN16 = 6, UN8 = 6, UPH8 = UN8 + 2;
using L8TailY = Lay32<4, 8>;                        // R1 + W2 for 32-byte pixels: written per PixelShuffle phase, read by HR_conv2
template <int DPF> struct Tail8Geo {
    static constexpr int LAG = 6;
    static constexpr int FR = 2 * DPF + 2;                   // fea0 / cond rings: fetched 2 DPF rows ahead of their one use
    static_assert(DPF + 3 <= UN16, "u ring: rows s + 2 (being quantised) .. s + DPF + 2 (DMA target)");
    static constexpr int OFF_U = 0, OFF_U8 = OFF_U + UN16 * U16_ROWB, OFF_F = OFF_U8 + UPH8 * U8_ROWB, OFF_C = OFF_F + FR * X_ROWB;
    static constexpr int OFF_Y = OFF_C + FR * C_ROWB, OFF_Z = OFF_Y + YPH * Y8_ROWB, OFF_TR = OFF_Z + YPH * Y8_ROWB;      // TR: 1 KiB the unused u piece lands in
    static constexpr int R_SLOTB = 256, RN = DPF + 1;        // residual planes: per group RN slots of [3 planes][32 px] f16
    static constexpr int OFF_R = OFF_TR + 1024;
    static constexpr int OFF_TU = OFF_R + 4 * RN * R_SLOTB;  // up_conv3: scale [128], shift [16][128]
    static constexpr int OFF_TH = OFF_TU + 17 * 128 * 4, OFF_TL = OFF_TH + 2176, OFF_K = OFF_TL + 2176;
    static constexpr int SMEM = OFF_K + 768;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(BIG % FR == 0 && BIG % UN16 == 0 && BIG % UN8 == 0, "BIG");
};

template <int DPF>
__global__ __launch_bounds__(512) void le_tail_rows_i8_kernel(RowsTailI8Params p)
{
    using G = Tail8Geo<DPF>;
    constexpr int FR = G::FR, LAG = G::LAG, RN = G::RN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned sm = lds_off(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS, hx0 = x0 >> 1;
    const int y0 = seg * p.rows_per_seg, yend = min(y0 + p.rows_per_seg, p.H);    // rows_per_seg is even
    const int ya = y0 - 2, hya = ya >> 1;                              // image row of ring row 0 (even); its half-resolution row
    const int nsteps = (yend - ya + LAG - 1) / 2 + 1;
    const int H = p.H, W = p.W, H1 = H >> 1, W1 = W >> 1;
    table_to_lds(smem + G::OFF_TU, p.up, tid, 128);
    table_to_lds(smem + G::OFF_TH, p.hr, tid);
    table_to_lds(smem + G::OFF_TL, p.last, tid);
    for (int e = tid; e < 192; e += 512) reinterpret_cast<float *>(smem + G::OFF_K)[e] = p.s.konst[e];
    const int g = wave & 3, gr = g >> 1, gh = g & 1;

    if (wave < 4) {
        // ------------------------------------------------------------------ role T1
        Bank8 wu;
        load_bank8(wu, p.up.wpk8, l31, lh, 128, 32 * g);
        Sft8 s2;
        load_sft8(s2, p.s, sm + G::OFF_K, lane, lh);
        f32x4 scq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) scq[qd] = *reinterpret_cast<const f32x4 *>(p.up.scale + 32 * g + 8 * qd + 4 * lh);
        const unsigned tab = sm + G::OFF_TU + 128 * 4 + 32 * g * 4;    // up_conv3's shift classes, this phase's 32 channels
        unsigned va[3];                                                // half-resolution pixel l31 reads u code slots l31 .. l31 + 2
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) va[kx] = sm + L8Std::at(l31 + kx, lh);
        const int cx = 2 * l31 + gh;                                   // full-resolution slot (image column x0 - 2 + cx) of this lane's pixel
        unsigned vf[2];                                                // fea0 read: this lane's two chunks of slot cx (stride 2 across the lanes)
        chunk_addr<LTailF>(vf, sm, cx, lh);
        const unsigned vw = sm + L8TailY::at(cx, lh);                  // Y code write
        const unsigned vc0 = sm + LCondT::at(cx, 0), vc1 = sm + LCondT::at(cx, 1);
        const bool col = (unsigned)(x0 - 2 + cx) < (unsigned)W;
        const int hcol = hx0 - 1 + l31;                                // this lane's half-resolution output column
        const int ccls = (hcol == 0 ? 1 : 0) | (hcol == W1 - 1 ? 2 : 0);
        const float q_inv = p.hr.q_inv, q_zoff = p.hr.q_zoff, slope = p.slope_relu, uq_inv = p.up.q_inv, uq_zoff = p.up.q_zoff;
        // the DMA: pieces 2 gh, 2 gh + 1 of fea0 row gr, piece gh of condition row gr, (waves 0-2) piece g of the f16 u row, the
        // residual planes of 32 pixels of an output row (lane = plane * 16 + pixel pair) -- always five DMA instructions per step
        const dma_rsrc_t rf = dma_rsrc(p.fea0), rc = dma_rsrc(p.cond), ru = dma_rsrc(p.u), rres = dma_rsrc(p.res_planar);
        const unsigned fl0 = LTailF::src_off(2 * gh, lane), fl1 = LTailF::src_off(2 * gh + 1, lane), cl = LCondT::src_off(gh, lane), ul = LStd::src_off(g, lane);
        const bool fok0 = (unsigned)(x0 - 2 + LTailF::src_px(2 * gh, lane)) < (unsigned)W, fok1 = (unsigned)(x0 - 2 + LTailF::src_px(2 * gh + 1, lane)) < (unsigned)W;
        const bool cok = (unsigned)(x0 - 2 + LCondT::src_px(gh, lane)) < (unsigned)W;
        const bool uok = g < 3 && LStd::src_px(g, lane) < 34 && (unsigned)(hx0 - 2 + LStd::src_px(g, lane)) < (unsigned)W1;
        const size_t plane = (size_t)H * W;
        const unsigned rl = (unsigned)((lane >> 4) * plane * 2 + (32 * gh + 2 * (lane & 15)) * 2);     // plane, pixel pair
        const bool rlok = lane < 48 && x0 + 32 * gh + 2 * (lane & 15) < W;
        const unsigned tr = sm + G::OFF_TR, rbuf = sm + G::OFF_R + g * RN * G::R_SLOTB;
        auto issue_fc = [&](int r, int fo, int co) __attribute__((always_inline)) {
            const bool rok = (unsigned)r < (unsigned)H && r <= yend + 1;
            const unsigned pix = (unsigned)(r * W + x0 - 2);
            dma16_at(rf, sm + fo + (2 * gh) * 1024, (rok && fok0) ? pix * 64u + fl0 : DMA_OOB);
            dma16_at(rf, sm + fo + (2 * gh + 1) * 1024, (rok && fok1) ? pix * 64u + fl1 : DMA_OOB);
            dma16_at(rc, sm + co + gh * 1024, (rok && cok) ? pix * 32u + cl : DMA_OOB);
        };
        auto issue_u = [&](int hr, int uo) __attribute__((always_inline)) {               // half-resolution image row hr into the f16 ring row at uo
            const bool rok = (unsigned)hr < (unsigned)H1 && hr <= ((yend + 1) >> 1) + 1;
            dma16_at(ru, g < 3 ? sm + uo + g * 1024 : tr, (rok && uok) ? (unsigned)((hr * W1 + hx0 - 2) * 64) + ul : DMA_OOB);
        };
        auto issue_r = [&](int r, int slot) __attribute__((always_inline)) {              // residual of output row r (32 pixels of column half gh)
            const bool rok = r >= y0 && r < yend;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rres, (__attribute__((address_space(3))) void *)(uintptr_t)(rbuf + slot * G::R_SLOTB), 4,
                                                     (rok && rlok) ? (unsigned)((r * W + x0) * 2) + rl : DMA_OOB, 0, 0, 0);
        };
        // up_conv3's input quantiser, once per element: f16 u row (ring row at uo) -> u code row (at qo; rows 0 / 1 of a lap also into
        // their second copy).  This wave: pixels 9 g .. 9 g + 8 of the 34, lane = (pixel, half); out-of-image pixels are code 0
        const int qpx = 9 * g + (lane >> 1), qh = lane & 1;
        const bool qact = lane < 18 && qpx < 34;
        const bool qcol = (unsigned)(hx0 - 2 + qpx) < (unsigned)W1;
        unsigned qsrc[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) qsrc[qd] = sm + LStd::at(qpx, qd) + 8 * qh;
        const unsigned qdst = sm + L8Std::at(qpx, qh);
        auto quant_u_row = [&](int uo, int qo, int hr) __attribute__((always_inline)) {   // hr: the row's half-resolution image row
            if (qact) {
                const bool in = qcol && (unsigned)hr < (unsigned)H1;
                i32x4 codes;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f16x4 v = lds_rd<f16x4>(qsrc[qd] + (unsigned)uo);
                    const unsigned w = quant4((float)v[0], (float)v[1], (float)v[2], (float)v[3], uq_inv, uq_zoff);
                    codes[qd] = in ? (int)w : 0;
                }
                put_codes<UN8 * U8_ROWB>(qdst, qo, qo < G::OFF_U8 + 2 * U8_ROWB, codes);
            }
        };
        // u rows -1 .. DPF + 1 (the ring's six rows), fea0 / cond / residual rows of the first DPF steps
#pragma unroll
        for (int ur = -1; ur <= DPF + 1; ++ur) issue_u(hya + ur, G::OFF_U + ((ur + BIG) % UN16) * U16_ROWB);
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq) {
            issue_fc(ya + 2 * sq + gr, G::OFF_F + ((2 * sq + gr + BIG) % FR) * X_ROWB, G::OFF_C + ((2 * sq + gr + BIG) % FR) * C_ROWB);
            issue_r(ya + 2 * sq - LAG + gr, sq % RN);
        }
        // ring rows of step s: this wave produces Y row ra = 2 s + gr (from fea0 / cond row ra, u code rows s - 1 .. s + 1); DMA into
        // fea0 / cond row 2 (s + DPF) + gr, f16 u row s + DPF + 2, residual slot (s + DPF) % RN; quantises u row s + 2
        Cur<G::OFF_F, FR, X_ROWB> fa(gr), fd(2 * DPF + gr);
        Cur<G::OFF_C, FR, C_ROWB> ca(gr), cd(2 * DPF + gr);
        Cur<G::OFF_Y, YN, Y8_ROWB> yw(gr);
        int uw = G::OFF_U8 + ((-1 + BIG) % UN8) * U8_ROWB;             // window start (code ring)
        int uq8 = G::OFF_U8 + ((2 + BIG) % UN8) * U8_ROWB, uq16 = G::OFF_U + ((2 + BIG) % UN16) * U16_ROWB;
        int ud = G::OFF_U + ((DPF + 2 + BIG) % UN16) * U16_ROWB, rs = DPF % RN;
        int ra_img = ya + gr;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();                                  // every wave's prologue pieces have landed
#pragma unroll
        for (int ur = -1; ur <= 1; ++ur)
            quant_u_row(G::OFF_U + ((ur + BIG) % UN16) * U16_ROWB, G::OFF_U8 + ((ur + BIG) % UN8) * U8_ROWB, hya + ur);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nsteps; ++s) {
            issue_fc(ra_img + 2 * DPF, fd.o, cd.o);
            issue_u(hya + s + DPF + 2, ud);
            issue_r(ra_img + 2 * DPF - LAG, rs);
            __builtin_amdgcn_sched_barrier(0);
            quant_u_row(uq16, uq8, hya + s + 2);
            const f16x8 c0 = lds_rd<f16x8>(vc0 + ca.o), c1 = lds_rd<f16x8>(vc1 + ca.o);
            f16x4 sk[4];
            get_row(vf, fa.o, sk);
            const bool in = col && (unsigned)ra_img < (unsigned)H;     // outside the image: HR_conv2's zero padding = code 0
            const int hrow = ra_img >> 1;
            const int bcls = (((hrow == 0 ? 1 : 0) | (hrow == H1 - 1 ? 2 : 0)) << 2) | ccls;
            i32x16 hacc;
            i32x4 hs, ht;
            f16x4 s1p[4], s0p[4];
            const i32x16 iacc = conv9<4, U8_ROWB>(wu, va, uw, [&](int t) __attribute__((always_inline)) {
                if (t == 0) hacc = sft8_hidden(s2, c0, c1);
                if (t == 3) sft8_mid(s2, hacc, hs, ht);
                if (t == 6) sft8_heads(s2, hs, ht, s1p, s0p);
            });
            f16x4 y[4];
            dequant_act<128 * 4>(iacc, scq, tab, bcls, lh, slope, y);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) y[qd] = y[qd] + sk[qd];     // relu(shuffle(up_conv3(u))) + fea0: conv32p's residual add in f16
            put_codes(vw, yw.o, yw.mirrored(), modulate_quant(y, s1p, s0p, q_inv, q_zoff, in));
            fa.step(); fd.step(); ca.step(); cd.step(); yw.step();
            uw += U8_ROWB; if (uw >= G::OFF_U8 + UN8 * U8_ROWB) uw -= UN8 * U8_ROWB;
            uq8 += U8_ROWB; if (uq8 >= G::OFF_U8 + UN8 * U8_ROWB) uq8 -= UN8 * U8_ROWB;
            uq16 += U16_ROWB; if (uq16 >= G::OFF_U + UN16 * U16_ROWB) uq16 -= UN16 * U16_ROWB;
            ud += U16_ROWB; if (ud >= G::OFF_U + UN16 * U16_ROWB) ud -= UN16 * U16_ROWB;
            rs = rs + 1 == RN ? 0 : rs + 1;
            ra_img += 2;
            // per step and wave five DMA instructions and nothing else: those of step s + 1 are older than the 5 (DPF - 1) since
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(5 * (DPF - 1), 0));
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // ------------------------------------------------------------------ role T2
        Bank8 wh, wl;
        load_bank8(wh, p.hr.wpk8, l31, lh);
        load_bank8(wl, p.last.wpk8, l31, lh);
        f32x4 scq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) scq[qd] = *reinterpret_cast<const f32x4 *>(p.hr.scale + 8 * qd + 4 * lh);
        const f32x4 scl = *reinterpret_cast<const f32x4 *>(p.last.scale);
        const unsigned tab_h = sm + G::OFF_TH + 128, tab_l = sm + G::OFF_TL + 128;
        const int cx = 32 * gh + l31;
        unsigned va[3], vy[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) { va[kx] = sm + L8Std::at(cx + kx, lh); vy[kx] = sm + L8TailY::at(cx + kx, lh); }
        const unsigned vw = sm + L8Std::at(cx, lh);                    // Z code write
        const int oxz = x0 - 1 + cx, oxo = x0 + cx;                    // Z slot cx = image column x0 - 1 + cx; the output pixel's column
        const bool colz = (unsigned)oxz < (unsigned)W;
        const int cclz = (oxz == 0 ? 1 : 0) | (oxz == W - 1 ? 2 : 0), cclo = (oxo == 0 ? 1 : 0) | (oxo == W - 1 ? 2 : 0);
        const float q_inv = p.last.q_inv, q_zoff = p.last.q_zoff, slope = p.slope_relu;
        const unsigned rbuf = sm + G::OFF_R + g * RN * G::R_SLOTB + l31 * 2;
        // output: channels 0..2 of pixel l31 sit in accumulator registers 0..2 of the lanes with lh == 0
        char *trash = p.trash + tid * 16;
        const size_t plane = (size_t)H * W;
        const bool cok = lh == 0 && cx < WS && x0 + cx < W;
        // ring rows of step s: HR_conv2 on rb = 2 s - 3 + gr (window rb - 1 ..), conv_last on ro = rb - 3
        Cur<G::OFF_Y, YN, Y8_ROWB> wy(gr - 4);
        Cur<G::OFF_Z, YN, Y8_ROWB> zw(gr - 3), wz(gr - LAG - 1);
        int rb_img = ya + gr - 3, rs = 0;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();                                  // (role T1's quantiser pass over the first u rows)
        for (int s = 0; s < nsteps; ++s) {
            const int r = rb_img - 3;                                  // the output row
            f16 res[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) res[ch] = lds_rd<f16>(rbuf + rs * G::R_SLOTB + ch * 64);
            {   // conv_last + residual -> the three output planes (conv32s<plain, i8, planar>'s epilogue)
                const i32x16 iacc = conv9<6>(wl, va, wz.o, [](int) {});
                const int bcls = (((r == 0 ? 1 : 0) | (r == H - 1 ? 2 : 0)) << 2) | cclo;
                const f32x4 sh = lds_rd<f32x4>(tab_l + (unsigned)(bcls * 128));
                const float o[3] = {act_fast((float)iacc[0] * scl[0] + sh[0], 1.f), act_fast((float)iacc[1] * scl[1] + sh[1], 1.f),
                                    act_fast((float)iacc[2] * scl[2] + sh[2], 1.f)};
                const bool ok = cok && r >= y0 && r < yend;
                f16 *d = p.dst_planar + (size_t)r * W + x0 + cx;
                const f16x4 oh = cvt_h4(o[0], o[1], o[2], 0.f);          // packed converts (common.h): fp32 -> f16 as a step of its own
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    // the conv result is rounded to f16, the residual added in fp32 and the sum rounded again
                    const float v = (float)oh[ch] + (float)res[ch];
                    *(ok ? d + ch * plane : reinterpret_cast<f16 *>(trash)) = (f16)v;
                }
            }
            {   // HR_conv2 + ReLU, conv_last's quantiser -> Z codes
                const i32x16 iacc = conv9<6>(wh, vy, wy.o, [](int) {});
                const bool in = colz && (unsigned)rb_img < (unsigned)H;     // outside the image: conv_last's zero padding = code 0
                const int bcls = (((rb_img == 0 ? 1 : 0) | (rb_img == H - 1 ? 2 : 0)) << 2) | cclz;
                f16x4 z[4];
                dequant_act(iacc, scq, tab_h, bcls, lh, slope, z);
                i32x4 codes;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const unsigned w = quant4((float)z[qd][0], (float)z[qd][1], (float)z[qd][2], (float)z[qd][3], q_inv, q_zoff);
                    codes[qd] = in ? (int)w : 0;
                }
                put_codes(vw, zw.o, zw.mirrored(), codes);
            }
            wy.step(); zw.step(); wz.step();
            rb_img += 2;
            rs = rs + 1 == RN ? 0 : rs + 1;
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));            // stores are never waited for
            __builtin_amdgcn_s_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// The full-resolution head (le_head_rows) with conv_first, SFT_layer1, HR_conv1 and down_conv1 all W8A8.  Per-layer form:
// conv_c3_q8 (image quantised while its patch is staged; two int8 MFMAs per 32 pixels), conv32s<sft-i8, i8> (SFT_layer1,
// quantise, HR_conv1, ReLU -> fea0) and conv_q8<32,3,2> (quantise fea0, down_conv1, ReLU -> fea1).  Here:
//     step s:  role H1 (waves 0-3, group (g >> 1, g & 1)): image rows 2s+3, 2s+4 into registers, staged as 4-byte code pixels
//                   {r, g, b, pad} at the end of the step; conv_first + ReLU + SFT_layer1 + HR_conv1's quantiser -> Y code rows 2s,
//                   2s+1; wave (s & 3): down_conv1 (stride 2) on F code rows 2s-7 .. 2s-5 -> fea1 row s-3
//              role H2 (waves 4-7): LDS-DMA of cond rows 2s+6, 2s+7; HR_conv1 + ReLU -> fea0 (stored from registers) and, through
//                   down_conv1's quantiser, F code rows 2s-3, 2s-2
constexpr int P8_SLOTS = 72, P8_ROWB = P8_SLOTS * 4;         // patch code ring: 68 of 72 pixel slots used (image columns x0 - 3 .. x0 + 64)
using L8HeadF = Lay32<8, 20>;                                // S2 + W1 for 32-byte pixels: written by HR_conv1, read by down_conv1 at stride 2
template <int DPF> struct Head8Geo {
    static constexpr int CR = 2 * DPF + 2;
    static constexpr int OFF_P = 0, OFF_C = OFF_P + YPH * P8_ROWB, OFF_Y = OFF_C + CR * C_ROWB, OFF_F = OFF_Y + YPH * Y8_ROWB;
    static constexpr int OFF_T = OFF_F + YPH * Y8_ROWB;     // tables: conv_first, HR_conv1, down_conv1 (2176 B each)
    static constexpr int OFF_K = OFF_T + 3 * 2176;
    static constexpr int SMEM = OFF_K + 768;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(BIG % CR == 0, "BIG");
};

template <int DPF>
__global__ __launch_bounds__(512) void le_head_rows_i8_kernel(RowsHeadI8Params p)
{
    using G = Head8Geo<DPF>;
    constexpr int CR = G::CR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned sm = lds_off(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS, hx0 = x0 >> 1;
    const int y0 = seg * p.rows_per_seg, yend = min(y0 + p.rows_per_seg, p.H);    // rows_per_seg is even
    const int ya = y0 - 2, hya = ya >> 1;
    const int nsteps = (yend - ya + 1) / 2 + 3;
    const int H = p.H, W = p.W, W1 = (W + 1) >> 1;
    table_to_lds(smem + G::OFF_T, p.cf, tid);
    table_to_lds(smem + G::OFF_T + 2176, p.hr, tid);
    table_to_lds(smem + G::OFF_T + 2 * 2176, p.dn, tid);
    for (int e = tid; e < 192; e += 512) reinterpret_cast<float *>(smem + G::OFF_K)[e] = p.s.konst[e];
    const int g = wave & 3, gr = g >> 1, gh = g & 1;
    const int cx = 32 * gh + l31;
    const float slope = p.slope_relu;

    if (wave < 4) {
        // ------------------------------------------------------------------ role H1
        const i32x4 *wq = reinterpret_cast<const i32x4 *>(p.cf.wpk8);
        const i32x4 cw0 = wq[lane], cw1 = wq[64 + lane];               // conv_first: kernel rows 0 | 1 in the two lane halves, then row 2 | zero
        f32x4 scf[4], scd[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            scf[qd] = *reinterpret_cast<const f32x4 *>(p.cf.scale + 8 * qd + 4 * lh);
            scd[qd] = *reinterpret_cast<const f32x4 *>(p.dn.scale + 8 * qd + 4 * lh);
        }
        const unsigned tab_f = sm + G::OFF_T + 128, tab_d = sm + G::OFF_T + 2 * 2176 + 128;
        Sft8 s1;
        load_sft8(s1, p.s, sm + G::OFF_K, lane, lh);
        Bank8 wd;
        load_bank8(wd, p.dn.wpk8, l31, lh);
        const unsigned vw = sm + L8Std::at(cx, lh);                    // Y code write
        const unsigned vc0 = sm + LCond::at(cx, 0), vc1 = sm + LCond::at(cx, 1);
        const int oxf = x0 - 2 + cx;                                   // Y slot cx = image column x0 - 2 + cx
        const bool col = (unsigned)oxf < (unsigned)W;
        const int cclf = (oxf == 0 ? 1 : 0) | (oxf == W - 1 ? 2 : 0);
        const float q_inv = p.hr.q_inv, q_zoff = p.hr.q_zoff, iq_inv = p.cf.q_inv, iq_zoff = p.cf.q_zoff;
        const unsigned vp = sm + (unsigned)cx * 4;                     // patch pixels cx .. cx + 3 of a row: kernel columns 0 .. 2 (+ the zero-weight slot)
        // patch staging: thread t < 136 owns pixel (t / 68, t % 68) of the two new rows
        const int pr = tid / 68, pc = tid - pr * 68;
        const bool pth = tid < 136, pcol = pth && (unsigned)(x0 - 3 + pc) < (unsigned)W;
        const size_t plane = (size_t)H * W;
        const f16 *pimg = p.img + (x0 - 3 + pc);
        f16 pv[3] = {(f16)0.f, (f16)0.f, (f16)0.f};
        bool pin_img = false;
        auto patch_fetch = [&](int r0) __attribute__((always_inline)) {      // image rows r0, r0 + 1
            const int r = r0 + pr;
            const bool ok = pcol && (unsigned)r < (unsigned)H && r <= yend + 1;
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
                const unsigned a = sm + G::OFF_P + ms * P8_ROWB + pc * 4;
                // conv_c3_q8's staging: the pixel's three codes + the pad slot's (a zero weight meets it); outside the image code 0
                const int v = pin_img ? (int)quant4((float)pv[0], (float)pv[1], (float)pv[2], 0.f, iq_inv, iq_zoff) : 0;
                lds_wr(a, v);
                if (ms < 2) lds_wr(a + YN * P8_ROWB, v);                        // the second copy of a lap's rows 0 and 1
            }
        };
        // down_conv1: half-resolution pixel l31 (column hx0 + l31, 30 used) reads F code slots 2 l31 + kx (slot c = image column x0 - 1 + c)
        unsigned vd[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) vd[kx] = sm + L8HeadF::at(2 * l31 + kx, lh);
        const int hcol = hx0 + l31;
        const bool dcol = l31 < WS / 2 && hcol < W1;
        const int ccld = (2 * hcol - 1 < 0 ? 1 : 0) | (2 * hcol + 1 >= W ? 2 : 0);
        f16 *const d1 = p.fea1 + (size_t)hcol * 32 + 4 * lh;
        for (int e = tid; e < YPH * 4; e += 256)                        // the four pad slots of every patch row hold a defined code
            lds_wr(sm + G::OFF_P + (e >> 2) * P8_ROWB + (68 + (e & 3)) * 4, 0);
        patch_fetch(ya - 1); patch_stage(YN - 1);                      // ring rows -1, 0
        patch_fetch(ya + 1); patch_stage(1);                           // ring rows 1, 2
        // ring rows of step s: Y row ra = 2 s + gr from patch rows ra - 1 .. ra + 1 and cond row ra; staging of patch rows 2 s + 3, + 4;
        // down_conv1 on F rows 2 s - 7 .. 2 s - 5
        Cur<G::OFF_C, CR, C_ROWB> ca(gr);
        Cur<G::OFF_P, YN, P8_ROWB> pw(gr - 1);
        Cur<G::OFF_Y, YN, Y8_ROWB> yw(gr);
        Cur<G::OFF_F, YN, Y8_ROWB> wf(-7);
        int ps = 3;                                                    // staging slot of ring row 2 s + 3 (3, 5, 1, ..)
        int ra_img = ya + gr;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nsteps; ++s) {
            patch_fetch(ya + 2 * s + 3);
            const f16x8 c0 = lds_rd<f16x8>(vc0 + ca.o), c1 = lds_rd<f16x8>(vc1 + ca.o);
            // conv_first (conv_c3_q8): B = four code pixels of patch row ra - 1 + lh, then of row ra + 1
            const unsigned a0 = vp + (unsigned)(pw.o + lh * P8_ROWB), a1 = vp + (unsigned)(pw.o + 2 * P8_ROWB);
            const i32x4 b0 = {lds_rd<int>(a0), lds_rd<int>(a0 + 4), lds_rd<int>(a0 + 8), lds_rd<int>(a0 + 12)};
            const i32x4 b1 = {lds_rd<int>(a1), lds_rd<int>(a1 + 4), lds_rd<int>(a1 + 8), lds_rd<int>(a1 + 12)};
            const bool in = col && (unsigned)ra_img < (unsigned)H;     // outside the image: HR_conv1's zero padding = code 0
            const int bcls = (((ra_img == 0 ? 1 : 0) | (ra_img == H - 1 ? 2 : 0)) << 2) | cclf;
            const i32x16 hacc = sft8_hidden(s1, c0, c1);
            i32x16 iacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(cw0, b0, izero16(), 0, 0, 0);
            iacc = __builtin_amdgcn_mfma_i32_32x32x32_i8(cw1, b1, iacc, 0, 0, 0);
            i32x4 hs, ht;
            sft8_mid(s1, hacc, hs, ht);
            f16x4 s1p[4], s0p[4];
            sft8_heads(s1, hs, ht, s1p, s0p);
            f16x4 f0[4];
            dequant_act(iacc, scf, tab_f, bcls, lh, slope, f0);
            put_codes(vw, yw.o, yw.mirrored(), modulate_quant(f0, s1p, s0p, q_inv, q_zoff, in));
            if (wave == (s & 3)) {
                const int hr = hya + s - 3;                            // down_conv1 on half-resolution row s - 3
                const i32x16 dacc = conv9<6>(wd, vd, wf.o, [](int) {});
                const int bcd = (((2 * hr - 1 < 0 ? 1 : 0) | (2 * hr + 1 >= H ? 2 : 0)) << 2) | ccld;
                f16x4 o[4];
                dequant_act(dacc, scd, tab_d, bcd, lh, slope, o);
                if (dcol && hr >= (y0 >> 1) && hr < ((yend + 1) >> 1)) {
                    f16 *d = d1 + (size_t)hr * W1 * 32;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<f16x4 *>(d + 8 * qd) = o[qd];
                }
            }
            patch_stage(ps);                                           // rows 2 s + 3, 2 s + 4 (fetched at the top of the step)
            ca.step(); pw.step(); yw.step(); wf.step();
            ps = ps + 2 >= YN ? ps + 2 - YN : ps + 2;
            ra_img += 2;
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // ------------------------------------------------------------------ role H2
        Bank8 wh;
        load_bank8(wh, p.hr.wpk8, l31, lh);
        f32x4 scq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) scq[qd] = *reinterpret_cast<const f32x4 *>(p.hr.scale + 8 * qd + 4 * lh);
        const unsigned tab_h = sm + G::OFF_T + 2176 + 128;
        unsigned va[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) va[kx] = sm + L8Std::at(cx + kx, lh);
        const unsigned vw = sm + L8HeadF::at(cx, lh);                  // F code write
        const int oxh = x0 - 1 + cx;                                   // F slot cx = image column x0 - 1 + cx
        const bool colf = (unsigned)oxh < (unsigned)W;
        const int cclh = (oxh == 0 ? 1 : 0) | (oxh == W - 1 ? 2 : 0);
        const float q_inv = p.dn.q_inv, q_zoff = p.dn.q_zoff;
        const dma_rsrc_t rc = dma_rsrc(p.cond);
        const unsigned cl = LCond::src_off(gh, lane);
        const bool cok = (unsigned)(x0 - 2 + LCond::src_px(gh, lane)) < (unsigned)W;
        auto issue_c = [&](int r, int co) __attribute__((always_inline)) {
            const bool rok = (unsigned)r < (unsigned)H && r <= yend;
            dma16_at(rc, sm + co + gh * 1024, (rok && cok) ? (unsigned)((r * W + x0 - 2) * 32) + cl : DMA_OOB);
        };
        char *trash = p.trash + tid * 16;
        const int ox = cx - 1;
        const bool ocol = ox >= 0 && ox < WS && x0 + ox < W;
        f16 *const dst0 = p.fea0 + ((ptrdiff_t)x0 + ox) * 32 + 8 * lh;
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq) issue_c(ya + 2 * sq + gr, G::OFF_C + ((2 * sq + gr + BIG) % CR) * C_ROWB);
        Cur<G::OFF_C, CR, C_ROWB> cd(2 * DPF + gr);
        Cur<G::OFF_Y, YN, Y8_ROWB> wy(gr - 4);
        Cur<G::OFF_F, YN, Y8_ROWB> fw(gr - 3);
        int rb_img = ya + gr - 3;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        for (int s = 0; s < nsteps; ++s) {
            issue_c(rb_img + 3 + 2 * DPF, cd.o);
            __builtin_amdgcn_sched_barrier(0);
            const i32x16 iacc = conv9<6>(wh, va, wy.o, [](int) {});
            const bool in = colf && (unsigned)rb_img < (unsigned)H;        // outside the image: down_conv1's zero padding = code 0
            const int bcls = (((rb_img == 0 ? 1 : 0) | (rb_img == H - 1 ? 2 : 0)) << 2) | cclh;
            f16x4 z[4];
            dequant_act(iacc, scq, tab_h, bcls, lh, slope, z);
            {   // fea0, as the tail's skip reads it: the f16 tensor
                f16x8 z0, z1;
                quads_to_chunks(z, z0, z1);
                f16 *d = (ocol && rb_img >= y0 && rb_img < yend) ? dst0 + (size_t)rb_img * W * 32 : reinterpret_cast<f16 *>(trash);
                *reinterpret_cast<f16x8 *>(d) = z0;
                *reinterpret_cast<f16x8 *>(d == reinterpret_cast<f16 *>(trash) ? d : d + 16) = z1;
            }
            i32x4 codes;                                               // conv_q8 quantises fea0 on load: the same f16 values
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const unsigned w = quant4((float)z[qd][0], (float)z[qd][1], (float)z[qd][2], (float)z[qd][3], q_inv, q_zoff);
                codes[qd] = in ? (int)w : 0;
            }
            put_codes(vw, fw.o, fw.mirrored(), codes);
            cd.step(); wy.step(); fw.step();
            rb_img += 2;
            // per step and wave: one DMA piece, then two stores: the piece step s + 1 reads is older than 3 (DPF - 1) + 2 operations
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(3 * (DPF - 1) + 2, 0));
            __builtin_amdgcn_s_barrier();
        }
    }
}

}  // namespace

hipError_t le_rb_rows_i8_launch(RowsRbI8Params p, int n_cu, hipStream_t s)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash) return hipErrorInvalidValue;
    static DevOnce once;
    int nseg;
    strips(p, n_cu, false, nseg);
    if (hipError_t e = set_lds(le_rb_rows_i8_kernel<3>, Rb8Geo<3>::SMEM, once)) return e;
    hipLaunchKernelGGL((le_rb_rows_i8_kernel<3>), dim3(p.nstrips * nseg), dim3(512), Rb8Geo<3>::SMEM, s, p);
    return hipGetLastError();
}

// H, W even; u is [H/2][W/2][32]
hipError_t le_tail_rows_i8_launch(RowsTailI8Params p, int n_cu, hipStream_t s)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash || (p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;
    static DevOnce once;
    int nseg;
    strips(p, n_cu, true, nseg);
    if (hipError_t e = set_lds(le_tail_rows_i8_kernel<3>, Tail8Geo<3>::SMEM, once)) return e;
    hipLaunchKernelGGL((le_tail_rows_i8_kernel<3>), dim3(p.nstrips * nseg), dim3(512), Tail8Geo<3>::SMEM, s, p);
    return hipGetLastError();
}

// H, W even (strips start on even columns: the half-resolution map is cut at x0 / 2); fea1 is [H/2][W/2][32]
hipError_t le_head_rows_i8_launch(RowsHeadI8Params p, int n_cu, hipStream_t s)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash || (p.W & 1) || (p.H & 1)) return hipErrorInvalidValue;
    static DevOnce once;
    int nseg;
    strips(p, n_cu, true, nseg);
    if (hipError_t e = set_lds(le_head_rows_i8_kernel<3>, Head8Geo<3>::SMEM, once)) return e;
    hipLaunchKernelGGL((le_head_rows_i8_kernel<3>), dim3(p.nstrips * nseg), dim3(512), Head8Geo<3>::SMEM, s, p);
    return hipGetLastError();
}
