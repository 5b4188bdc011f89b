// le_rows.hip -- row-streaming fused kernels of the LE main branch (gfx950): whole layer chains in ONE launch, every
// intermediate tensor in LDS rings of a few pixel rows, nothing but the chain's inputs and outputs in HBM.
//
// Reference: ResBlock_with_SFT.forward (arch_util.py:89-95)  y = x + conv2(sft2(relu(conv1(sft1(x, c))), c)),
// SFTLayer.forward (arch_util.py:66-72); the 16x16-tile kernels they replace are conv32s.hip's (two launches per block,
// the 32-channel intermediate written to and read back from HBM, every input halo fetched 1.27x).
//
// Schedule.  A workgroup owns a STRIP of 60 output columns and a SEGMENT of rows and walks down it two rows per step.
// The stages of the chain run skewed against each other, each on rows the previous stage finished a step earlier
// (DPF = 3, the conv1 -> sft2 hand-over inside a step):
//     step s:   LDS-DMA of rows 2s+6, 2s+7 (x: 64 px x 64 B, cond: 64 px x 32 B; three steps ahead)
//               sft1           -> Y1 rows 2s,   2s+1          (64 columns: the strip + 2 halo columns each side)
//               conv1 + sft2   -> Y2 rows 2s-3, 2s-2          (62 columns)
//               conv2 + x      -> out rows 2s-6, 2s-5         (60 columns; x from the ring the DMA filled)
// so a row is fetched ONCE (plus 4 of 64 columns shared with the neighbour strips and 4 rows per segment), there is no
// vertical recompute, and ONE s_barrier per step orders all rings (every ring slot is written and read in different steps).
// Waves have ROLES, so that a wave's 3x3 filter bank lives in its registers for the whole launch (18 A fragments = 72
// VGPRs; conv32s re-reads it from LDS for every 32 pixels): waves 0-3 (role B) run sft1, conv1 and sft2 -- LDS to LDS, not
// one memory operation -- and waves 4-7 (role C) the LDS-DMA, conv2, the residual and the stores; one 32-pixel group
// (row g >> 1, column half g & 1) per stage, step and wave.  Each SIMD holds one wave of either role; they meet at the barrier.
// The kernel is bound by vector-instruction issue (two waves per SIMD, ~40 MFMAs against a few hundred VALU / LDS
// instructions per step), so the code is written for few instructions: the Y rings keep their first two rows a second
// time behind the last (a 3-row conv window never wraps: one address per fragment column, kernel rows as immediates), an SFT
// pass (a chain of dependent MFMA -> VALU -> MFMA steps) runs between the MFMAs of an independent conv, ReLU / LeakyReLU and the
// residual add work on packed f16.
// Arithmetic, operand order and rounding points are conv32s's (K order (tap, k-step) on v_mfma_f32_32x32x16_f16, SFT in
// packed f16, bias added in fp32 behind the sum, residual add in f16): results are bit-identical to the two-launch form.
#include "launchers.h"

namespace {

// Diagnostic build only (make STAMP=1): per-phase s_memtime sums, written by lane 0 of every wave to
// p.dump[(block * 8 + wave) * 8 + phase] as cycles.  Never compiled into the shipped library.
#ifdef HDRTV_STAMP
#define STAMP_DECL unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(); st_acc[i] += st_t1 - st_t0; st_t0 = st_t1; __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_DUMP(p) do { if ((p).dump && lane == 0) for (int i_ = 0; i_ < 8; ++i_) reinterpret_cast<unsigned long long *>((p).dump)[((size_t)blockIdx.x * 8 + wave) * 8 + i_] = st_acc[i_]; } while (0)
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_DUMP(p)
#endif

// Diagnostic builds only (make EXTRA=-DRB_ABL=n, tools/abl_rows.sh): leave parts of the kernel out to see what its time is
// made of -- 1 no LDS-DMA, 2 no global stores, 4 no conv MFMAs, 8 no SFT passes, 16 no step barrier.  Results are garbage.
#ifndef RB_ABL
#define RB_ABL 0
#endif

constexpr int WS = 60;                  // output columns of a strip
constexpr int WI = 64;                  // input columns: 2 halo columns each side
constexpr int YP = 66;                  // pixel pitch of the Y rings (fragment reads of the two unused lanes run to slot 65)
constexpr int YN = 6, YPH = YN + 2;     // Y ring rows; physical rows: rows 0 and 1 of a lap are kept a second time behind row 5
constexpr int X_ROWB = WI * 64, C_ROWB = WI * 32, Y_ROWB = YP * 64;
constexpr int OUT_ROWB = 64 + 16, STRIP = 32 * OUT_ROWB;
constexpr int BIG = 168;                // multiple of every ring size: keeps (row + BIG) % ring non-negative
constexpr int SFT_TILE_F = 3 * 2 * 16;  // floats of one SFT layer's three bias tiles
// DPF: the LDS-DMA runs DPF steps ahead.  PIPE: sft2 runs one step behind its conv1 (inside the NEXT conv1's MFMA stream).
template <int DPF, bool PIPE> struct RbGeo {
    static constexpr int LAG = PIPE ? 8 : 6;                 // output rows trail the sft1 rows by LAG
    static constexpr int XR = 2 * DPF + LAG + 2;             // x ring: fetched 2 DPF rows ahead, read again LAG rows later (the residual)
    static constexpr int CR = 2 * DPF + LAG;                 // condition ring: last read by sft2, LAG - 3 rows behind
    static constexpr int OFF_X = 0, OFF_C = OFF_X + XR * X_ROWB, OFF_Y1 = OFF_C + CR * C_ROWB, OFF_Y2 = OFF_Y1 + YPH * Y_ROWB;
    static constexpr int OFF_ST = OFF_Y2 + YPH * Y_ROWB;
    static constexpr int OFF_B = OFF_ST + 4 * STRIP;         // conv1 | conv2 bias, then the two SFT layers' bias tiles
    static constexpr int SMEM = OFF_B + 256 + 2 * SFT_TILE_F * 4;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(BIG % XR == 0 && BIG % CR == 0 && BIG % YN == 0, "BIG");
};

__device__ __forceinline__ int swz32(int v) { return (v >> 2) & 3; }

// LDS reads while an LDS-DMA is in flight: hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of every LDS load that
// carries NO alias metadata -- in practice loads of HIP's struct vector types (float4 ...), which are aggregate copies
// without a TBAA tag -- and none in front of loads of clang ext_vector types (f16x8, f32x4: TBAA-tagged; the pass then
// consults its list of DMA stores with alias scopes, which is empty here).  With the DMA running steps ahead a
// vmcnt(0) in the loop drains the whole prefetch queue, so: ext_vector types only for LDS reads inside the step loop
// (tests/test_isa_contracts.py pins the loop's wait set).

// s_waitcnt immediate of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
// four fp32 -> f16 (round to nearest even) as two v_cvt_pk_f16_f32
__device__ __forceinline__ f16x4 cvt4(float a, float b, float c, float d)
{
    const f16x2 lo = __builtin_convertvector(f32x2{a, b}, f16x2), hi = __builtin_convertvector(f32x2{c, d}, f16x2);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
// accumulator quad qd (registers 4 qd .. 4 qd + 3) + bias -> f16
__device__ __forceinline__ f16x4 bias_cvt4(const f32x16 &acc, int qd, const f32x4 &b)
{
    const f32x2 lo = f32x2{acc[4 * qd], acc[4 * qd + 1]} + f32x2{b[0], b[1]}, hi = f32x2{acc[4 * qd + 2], acc[4 * qd + 3]} + f32x2{b[2], b[3]};
    return __builtin_shufflevector(__builtin_convertvector(lo, f16x2), __builtin_convertvector(hi, f16x2), 0, 1, 2, 3);
}
__device__ __forceinline__ f16x4 zero4() { return f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f}; }

__device__ __forceinline__ f16x8 lrelu_pack16(const f32x16 &a, int s)
{
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)a[8 * s + j];
    return __builtin_elementwise_max(o, o * (f16)0.1f);
}

// A 3x3 32 -> 32 filter bank as 18 A fragments of v_mfma_f32_32x32x16_f16 (wpk = [tap][32 out][32 in]; fragment (tap, ks):
// lane = out channel l31, input channels 16 ks + 8 lh ..)
struct Bank { f16x8 f[18]; };
__device__ __forceinline__ void load_bank(Bank &b, const f16 *wpk, int l31, int lh, int coutp = 32, int n0 = 0)
{
#pragma unroll
    for (int st = 0; st < 18; ++st)
        b.f[st] = *reinterpret_cast<const f16x8 *>(wpk + ((st >> 1) * coutp + n0 + l31) * 32 + 16 * (st & 1) + 8 * lh);
}

// The SFT layer's operands (pack_sft, hdrtv_api.hip): three A fragments (hidden stack, scale head, shift head) in registers;
// the three bias tiles (accumulator inits, (scale + 1) folded into the second) in LDS, 2 lane halves x 16 floats each
struct SftW { f16x8 a0, a1s, a1t; };
__device__ __forceinline__ void load_sft(SftW &s, const f16 *wfrag, int lane)
{
    const f16x8 *fr = reinterpret_cast<const f16x8 *>(wfrag);
    s.a0 = fr[lane]; s.a1s = fr[64 + lane]; s.a1t = fr[128 + lane];
}
__device__ __forceinline__ void sft_tiles_to_lds(float *dst, const float *bias, int tid)
{
    if (tid < SFT_TILE_F) {
        const int t = tid >> 5, lh = (tid >> 4) & 1, j = tid & 15;
        dst[tid] = bias[32 * t + 8 * (j >> 2) + 4 * lh + (j & 3)] + (t == 1 ? 1.f : 0.f);
    }
}
__device__ __forceinline__ f32x16 ld_tile(const float *t)
{
    f32x16 a;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(t + 4 * g);
        a[4 * g + 0] = v[0]; a[4 * g + 1] = v[1]; a[4 * g + 2] = v[2]; a[4 * g + 3] = v[3];
    }
    return a;
}
// y = x * (scale + 1) + shift on one pixel's 16 channels of this lane (channel quads qd: channels 8 qd + 4 lh ..), in three
// stages so that a caller can put independent work between the dependent MFMAs (conv32s's arithmetic, bit for bit)
__device__ __forceinline__ f32x16 sft_hidden(const SftW &s, const f16x8 &c0, const float *tiles)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(s.a0, c0, ld_tile(tiles), 0, 0, 0);
}
__device__ __forceinline__ void sft_heads(const SftW &s, const f32x16 &h, const float *tiles, f32x16 &sc, f32x16 &sh)
{
    const f16x8 hs = lrelu_pack16(h, 0), ht = lrelu_pack16(h, 1);
    sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(s.a1s, hs, ld_tile(tiles + 32), 0, 0, 0);
    sh = __builtin_amdgcn_mfma_f32_32x32x16_f16(s.a1t, ht, ld_tile(tiles + 64), 0, 0, 0);
}
__device__ __forceinline__ void sft_modulate(const f32x16 &sc, const f32x16 &sh, f16x4 (&y)[4])
{
#pragma unroll
    for (int qd = 0; qd < 4; ++qd)
        y[qd] = y[qd] * cvt4(sc[4 * qd], sc[4 * qd + 1], sc[4 * qd + 2], sc[4 * qd + 3]) + cvt4(sh[4 * qd], sh[4 * qd + 1], sh[4 * qd + 2], sh[4 * qd + 3]);
}

// 3x3 conv of one 32-pixel group out of a Y ring: a[kx][ks] = this lane's fragment address (kernel column kx, k-step ks) in
// the window's FIRST row; the window's rows are Y_ROWB apart (it never wraps, see YPH).  K order (tap, k-step); reads run
// AHEAD steps in front of the MFMAs; hook(st) runs behind MFMA st.
template <int AHEAD, int ROWB = Y_ROWB, class Hook>
__device__ __forceinline__ f32x16 conv18(const Bank &w, const char *smem, const int (&a)[3][2], Hook hook)
{
    f32x16 acc;
    f16x8 x[18];
    auto ld = [&](int st) __attribute__((always_inline)) {
        const int tap = st >> 1, ks = st & 1, ky = tap / 3, kx = tap % 3;
        x[st] = *reinterpret_cast<const f16x8 *>(smem + a[kx][ks] + ky * ROWB);
    };
#pragma unroll
    for (int st = 0; st < AHEAD; ++st) ld(st);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
        if (st + AHEAD < 18) ld(st + AHEAD);
        if (RB_ABL & 4) {
            if (st == 0) for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            acc[st & 15] += (float)x[st][0] * (float)w.f[st][0];
        } else if (st == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.f[0], x[0], z, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.f[st], x[st], acc, 0, 0, 0);
        }
        hook(st);
    }
    return acc;
}

// One pixel's 16 channels of this lane into ring row m of the Y ring at `ring` (and into its second copy, rows 0 and 1)
__device__ __forceinline__ void put_row(char *smem, int ring, int m, int q0, const f16x4 (&y)[4])
{
    char *d = smem + ring + m * Y_ROWB;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<f16x4 *>(d + (q0 ^ (qd << 4))) = y[qd];
    if (m < 2) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<f16x4 *>(d + YN * Y_ROWB + (q0 ^ (qd << 4))) = y[qd];
    }
}

// Fused ResBlock_with_SFT, rows.  Grid = nstrips x nseg workgroups of 512 threads.
template <int DPF, bool PIPE>
__global__ __launch_bounds__(512) void le_rb_rows_kernel(RowsRbParams p)
{
    using G = RbGeo<DPF, PIPE>;
    constexpr int LAG = G::LAG, XR = G::XR, CR = G::CR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS;
    const int y0 = seg * p.rows_per_seg, y1 = min(y0 + p.rows_per_seg, p.H);
    const int ya = y0 - 2;                                             // image row of ring row 0
    const int nsteps = (y1 - ya + LAG - 1) / 2 + 1;
    const int H = p.H, W = p.W;
    float *sB = reinterpret_cast<float *>(smem + G::OFF_B);
    if (tid < 32) { sB[tid] = p.b1[tid]; sB[32 + tid] = p.b2[tid]; }
    sft_tiles_to_lds(sB + 64, p.sft1_bias, tid);
    sft_tiles_to_lds(sB + 64 + SFT_TILE_F, p.sft2_bias, tid);
    const float *t1 = sB + 64 + 16 * lh, *t2 = t1 + SFT_TILE_F;

    const int g = wave & 3, gr = g >> 1, gh = g & 1;                   // this wave's 32-pixel group: row gr of the step's pair, column half gh
    const int cx = 32 * gh + l31;                                      // this lane's pixel slot in its group's ring rows
    int xo[3][2];                                                      // conv fragment offsets: output slot cx reads input slots cx .. cx + 2
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xo[kx][ks] = (cx + kx) * 64 + ((((ks << 1) | lh) ^ swz32(cx + kx)) << 4);
    const bool colfull = x0 >= 2 && x0 + 62 <= W;                      // no column of the strip's halo lies outside the image

    if (wave < 4) {
        // ------------------------------------------------------------------ role B: sft1, conv1, sft2 -- LDS to LDS, no memory operation
        Bank w1;
        load_bank(w1, p.w1, l31, lh);
        SftW s1, s2;
        load_sft(s1, p.sft1_wfrag, lane);
        load_sft(s2, p.sft2_wfrag, lane);
        const int q0 = cx * 64 + (swz32(cx) << 4) + 8 * lh;            // x read, Y1 / Y2 write: channel quad qd at q0 ^ (qd << 4)
        const int co1 = cx * 32 + ((lh ^ ((cx >> 3) & 1)) << 4);       // sft1: slot cx = image column x0 - 2 + cx
        const int co2 = (cx + 1) * 32 + ((lh ^ (((cx + 1) >> 3) & 1)) << 4);    // sft2: Y2 slot cx = image column x0 - 1 + cx = condition slot cx + 1
        const bool col1 = (unsigned)(x0 - 2 + cx) < (unsigned)W, col2 = (unsigned)(x0 - 1 + cx) < (unsigned)W;
        f16x4 yp[4] = {zero4(), zero4(), zero4(), zero4()};            // PIPE: conv1's result of the previous step
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            const int ra = 2 * s + gr, rb = 2 * s - 3 + gr, r2 = PIPE ? rb - 2 : rb;      // ring rows: sft1 on ra; conv1 on rb; sft2 on r2
            const char *xb = smem + G::OFF_X + ((ra + BIG) % XR) * X_ROWB;
            const f16x8 c1 = *reinterpret_cast<const f16x8 *>(smem + G::OFF_C + ((ra + BIG) % CR) * C_ROWB + co1);
            f16x4 ya4[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) ya4[qd] = *reinterpret_cast<const f16x4 *>(xb + (q0 ^ (qd << 4)));
            const f16x8 c2 = *reinterpret_cast<const f16x8 *>(smem + G::OFF_C + ((r2 + BIG) % CR) * C_ROWB + co2);
            f32x4 bq[4];                                               // conv1's bias: read here, not behind the conv (four dependent LDS round trips there)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) bq[qd] = *reinterpret_cast<const f32x4 *>(sB + 8 * qd + 4 * lh);
            const bool row1 = (unsigned)(ya + ra) < (unsigned)H, row2 = (unsigned)(ya + r2) < (unsigned)H;   // outside the image: zero padding
            const int m1 = (ra + BIG) % YN, m2 = (r2 + BIG) % YN;
            int a[3][2];
            {
                const int wb = G::OFF_Y1 + ((rb - 1 + BIG) % YN) * Y_ROWB;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a[kx][ks] = wb + xo[kx][ks];
            }
            // conv1 on row rb with row ra's SFT pass (and, PIPE, the one on the previous step's conv result) between its MFMAs: a
            // pass is a chain of dependent MFMA -> VALU -> MFMA steps, the conv an independent stream that covers its latencies
            f32x16 h1, sc1, sh1, h2, sc2, sh2;
            auto finish1 = [&]() __attribute__((always_inline)) {
                sft_modulate(sc1, sh1, ya4);
                if (!(colfull && row1)) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) if (!(col1 && row1)) ya4[qd] = zero4();
                }
                put_row(smem, G::OFF_Y1, m1, q0, ya4);
            };
            auto finish2 = [&](f16x4 (&y)[4]) __attribute__((always_inline)) {
                sft_modulate(sc2, sh2, y);
                if (!(colfull && row2)) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) if (!(col2 && row2)) y[qd] = zero4();
                }
                put_row(smem, G::OFF_Y2, m2, q0, y);
            };
            const f32x16 acc = conv18<4, Y_ROWB>(w1, smem, a, [&](int st) __attribute__((always_inline)) {
                if (RB_ABL & 8) { if (st == 12) put_row(smem, G::OFF_Y1, m1, q0, ya4); return; }
                if (st == 1) h1 = sft_hidden(s1, c1, t1);
                if (st == 6) sft_heads(s1, h1, t1, sc1, sh1);
                if (st == 12) finish1();
                if (PIPE) {
                    if (st == 3) h2 = sft_hidden(s2, c2, t2);
                    if (st == 9) sft_heads(s2, h2, t2, sc2, sh2);
                    if (st == 15) finish2(yp);
                }
            });
            STAMP(1);
            f16x4 y[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) y[qd] = __builtin_elementwise_max(bias_cvt4(acc, qd, bq[qd]), zero4());
            if (PIPE) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) yp[qd] = y[qd];
            } else if (RB_ABL & 8) {
                put_row(smem, G::OFF_Y2, m2, q0, y);
            } else {
                h2 = sft_hidden(s2, c2, t2);
                sft_heads(s2, h2, t2, sc2, sh2);
                finish2(y);
            }
            STAMP(2);
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
            STAMP(4);
            if (!(RB_ABL & 16)) __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    } else {
        // ------------------------------------------------------------------ role C: LDS-DMA, conv2 + x, stores
        Bank w2;
        load_bank(w2, p.w2, l31, lh);
        const dma_rsrc_t rx = dma_rsrc(p.x), rc = dma_rsrc(p.cond);
        // per step: pieces 2 gh, 2 gh + 1 of x row gr (16 pixels x 64 B each) and piece gh of condition row gr (32 pixels x 32 B)
        unsigned xl[2], cl;
        bool xok[2], cok;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int px = 16 * (2 * gh + j) + (lane >> 2), slot = lane & 3;
            xl[j] = (unsigned)(px * 64 + ((slot ^ swz32(px)) << 4));
            xok[j] = (unsigned)(x0 - 2 + px) < (unsigned)W;
        }
        {
            const int px = 32 * gh + (lane >> 1);
            cl = (unsigned)(px * 32 + (((lane & 1) ^ ((px >> 3) & 1)) << 4));
            cok = (unsigned)(x0 - 2 + px) < (unsigned)W;
        }
        auto issue = [&](int sq) __attribute__((always_inline)) {
            const int rr = 2 * sq + gr, r = ya + rr;
            const bool rok = (unsigned)r < (unsigned)H && r <= y1 + 1;
            const unsigned pix = (unsigned)(r * W + x0 - 2);
            char *d = smem + G::OFF_X + ((rr + BIG) % XR) * X_ROWB + (2 * gh) * 1024;
            if (RB_ABL & 1) return;
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16(rx, d + j * 1024, (rok && xok[j]) ? pix * 64u + xl[j] : DMA_OOB);
            dma16(rc, smem + G::OFF_C + ((rr + BIG) % CR) * C_ROWB + gh * 1024, (rok && cok) ? pix * 32u + cl : DMA_OOB);
        };
        // epilogue: the result + x goes through a wave-private strip and leaves as 16-byte chunks: this lane stores pixels
        // it * 16 + (lane >> 2) of the group, channel chunk c8 -- one step LATER, in front of the next conv (the strip's
        // write -> read -> store chain then runs under that conv's MFMAs)
        char *strip_b = smem + G::OFF_ST + g * STRIP;
        const int c8 = lane & 3, spx = lane >> 2;
        char *trash = p.trash + tid * 16;
        const int xr0 = (cx + 2) * 64 + (swz32(cx + 2) << 4) + 8 * lh; // x of output column x0 + cx: ring slot cx + 2
        const int sw0 = l31 * OUT_ROWB + 8 * lh;                       // strip write: quad qd at sw0 + 16 qd
        f32x4 bq[4];                                                   // conv2's bias
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) bq[qd] = *reinterpret_cast<const f32x4 *>(p.b2 + 8 * qd + 4 * lh);
        auto store_row = [&](int rr) __attribute__((always_inline)) {   // the strip holds ring row rr (output row ya + rr)
            const int r = ya + rr;
            const bool row_ok = r >= y0 && r < y1;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int ox = 32 * gh + it * 16 + spx;
                const f16x8 v = *reinterpret_cast<const f16x8 *>(strip_b + (it * 16 + spx) * OUT_ROWB + c8 * 16);
                const bool ok = row_ok && ox < WS && x0 + ox < W;
                f16 *d = ok ? p.dst + ((size_t)r * W + x0 + ox) * 32 + c8 * 8 : reinterpret_cast<f16 *>(trash);
                if (RB_ABL & 2) { if (v[0] == (f16)123.25f) *reinterpret_cast<f16x8 *>(trash) = v; continue; }
                *reinterpret_cast<f16x8 *>(d) = v;
            }
        };
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq) issue(sq);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            store_row(2 * (s - 1) - LAG + gr);                         // (step 0: a row above the segment, masked)
            issue(s + DPF);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            const int rr = 2 * s - LAG + gr;
            const char *xb = smem + G::OFF_X + ((rr + BIG) % XR) * X_ROWB;
            f16x4 res[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) res[qd] = *reinterpret_cast<const f16x4 *>(xb + (xr0 ^ (qd << 4)));
            int a[3][2];
            {
                const int wb = G::OFF_Y2 + ((rr - 1 + BIG) % YN) * Y_ROWB;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a[kx][ks] = wb + xo[kx][ks];
            }
            const f32x16 acc = conv18<6, Y_ROWB>(w2, smem, a, [](int) {});
            STAMP(1);
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<f16x4 *>(strip_b + sw0 + 16 * qd) = bias_cvt4(acc, qd, bq[qd]) + res[qd];
            STAMP(3);
            // per step and wave: two stores, then three DMA pieces (all always issued): the pieces of step s + 1 are older
            // than the 5 (DPF - 1) operations of the steps since
            __builtin_amdgcn_s_waitcnt(waitcnt_imm((RB_ABL & 3) ? 0 : 5 * (DPF - 1), 0));
            STAMP(4);
            if (!(RB_ABL & 16)) __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        store_row(2 * (nsteps - 1) - LAG + gr);
        STAMP_DUMP(p);
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The full-resolution tail of the LE net in one launch (HDRUNet3T1_arch.py:196-206):
//     out = agcm + conv_last(relu(HR_conv2(SFT_layer2(relu(shuffle(up_conv3(u))) + fea0, cond1))))
// u is the half-resolution trunk output; the per-layer form is conv32p<4> (up-conv + PixelShuffle + ReLU + skip, 0.53 GB
// written), conv32s<sft> (read back, 0.53 GB written) and conv32s<plain, planar> (read back).  Here the two 32-channel
// full-resolution tensors live in LDS rings (Y: the modulated up-conv output, Z: relu(HR_conv2)); HBM sees u, fea0, cond1, the
// residual planes and the three output planes.  Strip / segment / step structure, rings and roles as the ResBlock kernel above:
//     step s:  LDS-DMA of fea0 / cond rows 2s+6, 2s+7 and of u row s+4 (half resolution: 34 of 48 slots used)
//              role T1 (waves 0-3, wave b = PixelShuffle position (b >> 1, b & 1)): up_conv3 bank b on u rows s-1 .. s+1 -> 32
//                   half-resolution pixels = every other pixel of full-resolution row 2s + (b >> 1); ReLU, + fea0, SFT_layer2
//                   (its two cond MLPs inside the conv's MFMA stream) -> Y rows 2s, 2s+1
//              role T2 (waves 4-7, group (g >> 1, g & 1)): the DMA; HR_conv2 + ReLU -> Z rows 2s-3, 2s-2; conv_last + residual
//                   -> output rows 2s-6, 2s-5 (planar)
// Per element the arithmetic, K order and rounding points are those of the per-layer kernels: results are bit-identical.
constexpr int U_SLOTS = 48, U_ROWB = U_SLOTS * 64, UN = 6, UPH = UN + 2;     // u ring: rows mirrored as the Y rings (a second DMA)
template <int DPF> struct TailGeo {
    static constexpr int FR = 2 * DPF + 2;                   // fea0 / cond rings: fetched 2 DPF rows ahead of their one use
    static_assert(DPF + 3 <= UN, "u ring");
    static constexpr int OFF_U = 0, OFF_F = OFF_U + UPH * U_ROWB, OFF_C = OFF_F + FR * X_ROWB, OFF_Y = OFF_C + FR * C_ROWB;
    static constexpr int OFF_Z = OFF_Y + YPH * Y_ROWB, OFF_TR = OFF_Z + YPH * Y_ROWB;       // TR: 4 x 1 KiB the unused mirror DMAs land in
    static constexpr int OFF_R = OFF_TR + 4096;              // residual planes: per T2 wave (DPF + 1) slots of [3 planes][32 px] f16
    static constexpr int R_SLOTB = 256, R_WAVEB = (DPF + 1) * R_SLOTB;
    static constexpr int OFF_B = OFF_R + 4 * R_WAVEB;        // SFT_layer2's bias tiles
    static constexpr int SMEM = OFF_B + SFT_TILE_F * 4;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(BIG % FR == 0, "BIG");
};

template <int DPF>
__global__ __launch_bounds__(512) void le_tail_rows_kernel(RowsTailParams p)
{
    using G = TailGeo<DPF>;
    constexpr int FR = G::FR, LAG = 6;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS, hx0 = x0 >> 1;
    const int y0 = seg * p.rows_per_seg, y1 = min(y0 + p.rows_per_seg, p.H);      // rows_per_seg is even
    const int ya = y0 - 2, hya = ya >> 1;                              // image row of ring row 0 (even); its half-resolution row
    const int nsteps = (y1 - ya + LAG - 1) / 2 + 1;
    const int H = p.H, W = p.W, H1 = H >> 1, W1 = W >> 1;
    float *sB = reinterpret_cast<float *>(smem + G::OFF_B);
    sft_tiles_to_lds(sB, p.sft_bias, tid);
    const float *t2 = sB + 16 * lh;
    const bool colfull = x0 >= 2 && x0 + 62 <= W;

    if (wave < 4) {
        // ------------------------------------------------------------------ role T1: up_conv3 bank b, + fea0, SFT_layer2 -> Y
        const int b = wave, bi = b >> 1, bj = b & 1;
        Bank wu;
        load_bank(wu, p.w_up, l31, lh, 128, 32 * b);
        SftW s2;
        load_sft(s2, p.sft_wfrag, lane);
        f32x4 bq[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) bq[qd] = *reinterpret_cast<const f32x4 *>(p.b_up + 32 * b + 8 * qd + 4 * lh);
        int xo[3][2];                                                  // half-resolution pixel l31 reads u slots l31 .. l31 + 2
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xo[kx][ks] = (l31 + kx) * 64 + ((((ks << 1) | lh) ^ swz32(l31 + kx)) << 4);
        const int cx = 2 * l31 + bj;                                   // full-resolution slot (image column x0 - 2 + cx) of this lane's pixel
        const int q0 = cx * 64 + (swz32(cx) << 4) + 8 * lh;            // fea0 read, Y write: channel quad qd at q0 ^ (qd << 4)
        const int co = cx * 32 + ((lh ^ ((cx >> 3) & 1)) << 4);
        const bool col = (unsigned)(x0 - 2 + cx) < (unsigned)W;
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            const int ra = 2 * s + bi;                                 // the full-resolution ring row this wave produces
            const char *fb = smem + G::OFF_F + ((ra + BIG) % FR) * X_ROWB;
            const f16x8 c2 = *reinterpret_cast<const f16x8 *>(smem + G::OFF_C + ((ra + BIG) % FR) * C_ROWB + co);
            f16x4 sk[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) sk[qd] = *reinterpret_cast<const f16x4 *>(fb + (q0 ^ (qd << 4)));
            const bool row = (unsigned)(ya + ra) < (unsigned)H;
            const int m = (ra + BIG) % YN;
            int a[3][2];
            {
                const int wb = G::OFF_U + ((s - 1 + BIG) % UN) * U_ROWB;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a[kx][ks] = wb + xo[kx][ks];
            }
            f32x16 h2, sc2, sh2;
            const f32x16 acc = conv18<4, U_ROWB>(wu, smem, a, [&](int st) __attribute__((always_inline)) {
                if (st == 2) h2 = sft_hidden(s2, c2, t2);
                if (st == 9) sft_heads(s2, h2, t2, sc2, sh2);
            });
            STAMP(1);
            f16x4 y[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) y[qd] = __builtin_elementwise_max(bias_cvt4(acc, qd, bq[qd]), zero4()) + sk[qd];
            sft_modulate(sc2, sh2, y);
            if (!(colfull && row)) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) if (!(col && row)) y[qd] = zero4();
            }
            put_row(smem, G::OFF_Y, m, q0, y);
            STAMP(2);
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    } else {
        // ------------------------------------------------------------------ role T2: LDS-DMA, HR_conv2 -> Z, conv_last + residual -> out
        const int g = wave & 3, gr = g >> 1, gh = g & 1;
        Bank wh, wl;
        load_bank(wh, p.w_hr, l31, lh);
        load_bank(wl, p.w_last, l31, lh);
        f32x4 bh[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) bh[qd] = *reinterpret_cast<const f32x4 *>(p.b_hr + 8 * qd + 4 * lh);
        const float bl0 = p.b_last[0], bl1 = p.b_last[1], bl2 = p.b_last[2];
        const int cx = 32 * gh + l31;
        int xo[3][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xo[kx][ks] = (cx + kx) * 64 + ((((ks << 1) | lh) ^ swz32(cx + kx)) << 4);
        const int q0 = cx * 64 + (swz32(cx) << 4) + 8 * lh;            // Z write
        const bool colz = (unsigned)(x0 - 1 + cx) < (unsigned)W;       // Z slot cx = image column x0 - 1 + cx
        const dma_rsrc_t rf = dma_rsrc(p.fea0), rc = dma_rsrc(p.cond), ru = dma_rsrc(p.u), rr_ = dma_rsrc(p.res_planar);
        // per step and wave: pieces 2 gh, 2 gh + 1 of fea0 row gr, piece gh of condition row gr, (waves 0-2) piece g of the u
        // row and of its second copy, and the residual planes of an output row's 32 pixels (4-byte DMA: lane = plane * 16 +
        // pixel pair) -- always six DMA instructions (the unused ones fetch nothing into a trash KiB).  No plain load: its
        // first use would wait (vmcnt retires in order) for every DMA issued before it.
        unsigned fl[2], cl, ul;
        bool fok[2], cok, uok;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int px = 16 * (2 * gh + j) + (lane >> 2), slot = lane & 3;
            fl[j] = (unsigned)(px * 64 + ((slot ^ swz32(px)) << 4));
            fok[j] = (unsigned)(x0 - 2 + px) < (unsigned)W;
        }
        {
            const int px = 32 * gh + (lane >> 1);
            cl = (unsigned)(px * 32 + (((lane & 1) ^ ((px >> 3) & 1)) << 4));
            cok = (unsigned)(x0 - 2 + px) < (unsigned)W;
        }
        {
            const int px = 16 * g + (lane >> 2), slot = lane & 3;      // u slot px = half-resolution column hx0 - 2 + px
            ul = (unsigned)(px * 64 + ((slot ^ swz32(px)) << 4));
            uok = g < 3 && px < 34 && (unsigned)(hx0 - 2 + px) < (unsigned)W1;
        }
        char *tr = smem + G::OFF_TR + g * 1024;
        char *rbuf = smem + G::OFF_R + g * G::R_WAVEB;
        const size_t plane = (size_t)H * W;
        const unsigned rl = (unsigned)((lane >> 4) * plane * 2 + (32 * gh + 2 * (lane & 15)) * 2);     // plane, pixel pair
        const bool rlok = lane < 48 && x0 + 32 * gh + 2 * (lane & 15) < W;
        auto issue_r = [&](int sq) __attribute__((always_inline)) {     // residual of the output row of step sq
            const int r = ya + 2 * sq - LAG + gr;
            const bool rok = r >= y0 && r < y1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rr_, (__attribute__((address_space(3))) void *)(rbuf + ((sq + DPF + 1) % (DPF + 1)) * G::R_SLOTB), 4,
                                                     (rok && rlok) ? (unsigned)((r * W + x0) * 2) + rl : DMA_OOB, 0, 0, 0);
        };
        auto issue_fc = [&](int sq) __attribute__((always_inline)) {
            const int rr = 2 * sq + gr, r = ya + rr;
            const bool rok = (unsigned)r < (unsigned)H && r <= y1 + 1;
            const unsigned pix = (unsigned)(r * W + x0 - 2);
            char *d = smem + G::OFF_F + ((rr + BIG) % FR) * X_ROWB + (2 * gh) * 1024;
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16(rf, d + j * 1024, (rok && fok[j]) ? pix * 64u + fl[j] : DMA_OOB);
            dma16(rc, smem + G::OFF_C + ((rr + BIG) % FR) * C_ROWB + gh * 1024, (rok && cok) ? pix * 32u + cl : DMA_OOB);
        };
        auto issue_u = [&](int ur) __attribute__((always_inline)) {     // u ring row ur = half-resolution image row hya + ur
            const int hr = hya + ur, m = (ur + BIG) % UN;
            const bool rok = (unsigned)hr < (unsigned)H1 && hr <= ((y1 + 1) >> 1) + 1;
            const unsigned off = (rok && uok) ? (unsigned)((hr * W1 + hx0 - 2) * 64) + ul : DMA_OOB;
            dma16(ru, g < 3 ? smem + G::OFF_U + m * U_ROWB + g * 1024 : tr, off);
            dma16(ru, (g < 3 && m < 2) ? smem + G::OFF_U + (m + UN) * U_ROWB + g * 1024 : tr, (m < 2) ? off : DMA_OOB);
        };
        // output: channels 0..2 of pixel l31 sit in accumulator registers 0..2 of the lanes with lh == 0
        char *trash = p.trash + tid * 16;
        issue_u(-1);
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq) { issue_fc(sq); issue_u(sq); issue_r(sq); }
        issue_u(DPF);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            issue_fc(s + DPF);
            issue_u(s + DPF + 1);
            issue_r(s + DPF);
            __builtin_amdgcn_sched_barrier(0);
            // residual planes of this step's output row (fetched DPF steps ago)
            const int ro = 2 * s - LAG + gr, r = ya + ro;
            const bool ok = lh == 0 && r >= y0 && r < y1 && cx < WS && x0 + cx < W;
            const size_t e = ok ? (size_t)r * W + x0 + cx : 0;
            f16 res[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) res[ch] = *reinterpret_cast<const f16 *>(rbuf + (s % (DPF + 1)) * G::R_SLOTB + ch * 64 + l31 * 2);
            STAMP(0);
            {   // HR_conv2 + ReLU on ring row 2 s - 3 + gr -> Z
                const int rb = 2 * s - 3 + gr;
                int a[3][2];
                const int wb = G::OFF_Y + ((rb - 1 + BIG) % YN) * Y_ROWB;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a[kx][ks] = wb + xo[kx][ks];
                const f32x16 acc = conv18<6, Y_ROWB>(wh, smem, a, [](int) {});
                f16x4 z[4];
                const bool in = colz && (unsigned)(ya + rb) < (unsigned)H;     // outside the image: conv_last's zero padding
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    z[qd] = __builtin_elementwise_max(bias_cvt4(acc, qd, bh[qd]), zero4());
                    if (!in) z[qd] = zero4();
                }
                put_row(smem, G::OFF_Z, (rb + BIG) % YN, q0, z);
            }
            STAMP(1);
            {   // conv_last on ring row 2 s - 6 + gr, + residual -> the three output planes
                int a[3][2];
                const int wb = G::OFF_Z + ((ro - 1 + BIG) % YN) * Y_ROWB;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a[kx][ks] = wb + xo[kx][ks];
                const f32x16 acc = conv18<6, Y_ROWB>(wl, smem, a, [](int) {});
                const float o[3] = {acc[0] + bl0, acc[1] + bl1, acc[2] + bl2};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    // the conv result is rounded to f16, the residual added in fp32 and the sum rounded again (conv32s PLANAR)
                    const float v = (float)(f16)o[ch] + (float)res[ch];
                    f16 *d = ok ? p.dst_planar + ch * plane + e : reinterpret_cast<f16 *>(trash);
                    *d = (f16)v;
                }
            }
            STAMP(3);
            // per step and wave: 6 DMA instructions, then 3 stores (all always issued): the pieces of step s + 1 are older than
            // the 9 (DPF - 1) operations of the steps since and the 3 stores of their own step
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(3 + 9 * (DPF - 1), 0));
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    }
}


// ------------------------------------------------------------------------------------------------------------------------
// The full-resolution head of the LE net in one launch (HDRUNet3T1_arch.py:168-172):
//     fea0 = relu(HR_conv1(SFT_layer1(relu(conv_first(img)), cond1)));   fea1a = relu(down_conv1(fea0))
// The per-layer form is conv32s<c3+sft> (conv_first recomputed per 16x16 tile on the prep waves: 2.4 TB/s) and conv_t16 (fea0
// read back).  Here HBM sees the three image planes, cond1, and the two outputs.  Structure as above:
//     step s:  role H1 (waves 0-3, group (g >> 1, g & 1)): global loads of image rows 2s+3, 2s+4 into registers, staged as
//                   {R, G, B, 1} pixels into the patch ring at the end of the step; conv_first + ReLU + SFT_layer1 -> Y rows 2s, 2s+1;
//                   wave (s & 3): down_conv1 on half-resolution row s-3 (fea0 rows 2s-7 .. 2s-5 of the F ring) -> fea1a
//              role H2 (waves 4-7): LDS-DMA of cond rows 2s+6, 2s+7; HR_conv1 + ReLU -> F rows 2s-3, 2s-2 and (one step
//                   later, through a strip) fea0
// conv_first is conv32s's C3 form (K = (kernel column | pad, channel | bias slot) per kernel row, three MFMAs).
constexpr int P_SLOTS = 72, P_ROWB = P_SLOTS * 8;            // patch ring: 68 of 72 pixel slots used (image columns x0 - 3 .. x0 + 64)
template <int DPF> struct HeadGeo {
    static constexpr int CR = 2 * DPF + 2;
    static constexpr int OFF_P = 0, OFF_C = OFF_P + YPH * P_ROWB, OFF_Y = OFF_C + CR * C_ROWB, OFF_F = OFF_Y + YPH * Y_ROWB;
    static constexpr int OFF_ST = OFF_F + YPH * Y_ROWB;      // H2's four output strips
    static constexpr int OFF_B = OFF_ST + 4 * STRIP;         // SFT_layer1's bias tiles
    static constexpr int SMEM = OFF_B + SFT_TILE_F * 4;
    static_assert(SMEM <= 160 * 1024, "LDS budget");
    static_assert(BIG % CR == 0, "BIG");
};

template <int DPF>
__global__ __launch_bounds__(512) void le_head_rows_kernel(RowsHeadParams p)
{
    using G = HeadGeo<DPF>;
    constexpr int CR = G::CR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS, hx0 = x0 >> 1;
    const int y0 = seg * p.rows_per_seg, y1 = min(y0 + p.rows_per_seg, p.H);      // rows_per_seg is even
    const int ya = y0 - 2, hya = ya >> 1;
    const int nsteps = (y1 - ya + 1) / 2 + 3;
    const int H = p.H, W = p.W, W1 = (W + 1) >> 1;
    float *sB = reinterpret_cast<float *>(smem + G::OFF_B);
    sft_tiles_to_lds(sB, p.sft_bias, tid);
    const float *t1 = sB + 16 * lh;
    const bool colfull = x0 >= 2 && x0 + 62 <= W;
    const int g = wave & 3, gr = g >> 1, gh = g & 1;
    const int cx = 32 * gh + l31;
    int xo[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xo[kx][ks] = (cx + kx) * 64 + ((((ks << 1) | lh) ^ swz32(cx + kx)) << 4);
    const int q0 = cx * 64 + (swz32(cx) << 4) + 8 * lh;                // Y / F write: channel quad qd at q0 ^ (qd << 4)

    if (wave < 4) {
        // ------------------------------------------------------------------ role H1: image patch, conv_first, SFT_layer1 -> Y; down_conv1
        f16x8 c3w[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) c3w[ky] = reinterpret_cast<const f16x8 *>(p.c3_wfrag)[ky * 64 + lane];
        SftW s1;
        load_sft(s1, p.sft_wfrag, lane);
        Bank wd;
        load_bank(wd, p.w_down, l31, lh);
        f32x4 bd[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) bd[qd] = *reinterpret_cast<const f32x4 *>(p.b_down + 8 * qd + 4 * lh);
        const int co = cx * 32 + ((lh ^ ((cx >> 3) & 1)) << 4);
        const bool col = (unsigned)(x0 - 2 + cx) < (unsigned)W;
        const int po = (cx + 2 * lh) * 8;                              // patch pixels cx + 2 lh, + 1 of a kernel row = K slots 8 lh .. 8 lh + 7
        // patch staging: thread t < 136 owns pixel (t / 68, t % 68) of the two new rows
        const int pr = tid / 68, pc = tid - pr * 68;
        const bool pth = tid < 136, pcol = pth && (unsigned)(x0 - 3 + pc) < (unsigned)W;
        const size_t plane = (size_t)H * W;
        f16 pv[3] = {(f16)0.f, (f16)0.f, (f16)0.f};
        auto patch_fetch = [&](int rr0) __attribute__((always_inline)) {     // ring rows rr0, rr0 + 1
            const int r = ya + rr0 + pr;
            const bool ok = pcol && (unsigned)r < (unsigned)H && r <= y1 + 1;
            const size_t o = ok ? (size_t)r * W + (x0 - 3 + pc) : 0;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                pv[ch] = p.img[ch * plane + o];
                if (!ok) pv[ch] = (f16)0.f;
            }
        };
        auto patch_stage = [&](int rr0) __attribute__((always_inline)) {
            if (pth) {
                const int m = (rr0 + pr + BIG) % YN;
                char *d = smem + G::OFF_P + m * P_ROWB + pc * 8;
                const f16x4 v = f16x4{pv[0], pv[1], pv[2], (f16)1.f};          // 1: the bias slot
                *reinterpret_cast<f16x4 *>(d) = v;
                if (m < 2) *reinterpret_cast<f16x4 *>(d + YN * P_ROWB) = v;
            }
        };
        // down_conv1: half-resolution pixel l31 (column hx0 + l31, 30 used) reads F slots 2 l31 + kx (slot c = image column x0 - 1 + c)
        int xd[3][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xd[kx][ks] = (2 * l31 + kx) * 64 + ((((ks << 1) | lh) ^ swz32(2 * l31 + kx)) << 4);
        const bool dcol = l31 < WS / 2 && hx0 + l31 < W1;
        for (int e = tid; e < YPH * 4; e += 256)                        // the four pad slots of every patch row stay finite (zero weights read them)
            *reinterpret_cast<f16x4 *>(smem + G::OFF_P + (e >> 2) * P_ROWB + (68 + (e & 3)) * 8) = zero4();
        patch_fetch(-1); patch_stage(-1);
        patch_fetch(1); patch_stage(1);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            patch_fetch(2 * s + 3);
            const int ra = 2 * s + gr;
            const f16x8 c1 = *reinterpret_cast<const f16x8 *>(smem + G::OFF_C + ((ra + BIG) % CR) * C_ROWB + co);
            const char *pb = smem + G::OFF_P + ((ra - 1 + BIG) % YN) * P_ROWB + po;
            f32x16 h;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const f16x4 u = *reinterpret_cast<const f16x4 *>(pb + ky * P_ROWB), v = *reinterpret_cast<const f16x4 *>(pb + ky * P_ROWB + 8);
                const f16x8 xf = __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7);
                if (ky == 0) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    h = __builtin_amdgcn_mfma_f32_32x32x16_f16(c3w[0], xf, z, 0, 0, 0);
                } else {
                    h = __builtin_amdgcn_mfma_f32_32x32x16_f16(c3w[ky], xf, h, 0, 0, 0);
                }
            }
            f32x16 sc1, sh1;
            sft_heads(s1, sft_hidden(s1, c1, t1), t1, sc1, sh1);
            f16x4 y[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) y[qd] = __builtin_elementwise_max(cvt4(h[4 * qd], h[4 * qd + 1], h[4 * qd + 2], h[4 * qd + 3]), zero4());
            sft_modulate(sc1, sh1, y);
            const bool row = (unsigned)(ya + ra) < (unsigned)H;
            if (!(colfull && row)) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) if (!(col && row)) y[qd] = zero4();
            }
            put_row(smem, G::OFF_Y, (ra + BIG) % YN, q0, y);
            STAMP(1);
            if (wave == (s & 3)) {
                // down_conv1 on ring half row s - 3: F rows 2 s - 7 .. 2 s - 5
                const int j = s - 3, hr = hya + j;
                int a[3][2];
                const int wb = G::OFF_F + ((2 * j - 1 + BIG) % YN) * Y_ROWB;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a[kx][ks] = wb + xd[kx][ks];
                const f32x16 acc = conv18<4, Y_ROWB>(wd, smem, a, [](int) {});
                if (dcol && hr >= (y0 >> 1) && hr < ((y1 + 1) >> 1)) {
                    f16 *d = p.fea1 + ((size_t)hr * W1 + hx0 + l31) * 32 + 4 * lh;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) *reinterpret_cast<f16x4 *>(d + 8 * qd) = __builtin_elementwise_max(bias_cvt4(acc, qd, bd[qd]), zero4());
                }
            }
            STAMP(2);
            patch_stage(2 * s + 3);
            STAMP(3);
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    } else {
        // ------------------------------------------------------------------ role H2: LDS-DMA of cond, HR_conv1 -> F ring and fea0
        Bank wh;
        load_bank(wh, p.w_hr, l31, lh);
        f32x4 bh[4];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) bh[qd] = *reinterpret_cast<const f32x4 *>(p.b_hr + 8 * qd + 4 * lh);
        const bool colf = (unsigned)(x0 - 1 + cx) < (unsigned)W;       // F slot cx = image column x0 - 1 + cx
        const dma_rsrc_t rc = dma_rsrc(p.cond);
        const int cpx = 32 * gh + (lane >> 1);
        const unsigned cl = (unsigned)(cpx * 32 + (((lane & 1) ^ ((cpx >> 3) & 1)) << 4));
        const bool cok = (unsigned)(x0 - 2 + cpx) < (unsigned)W;
        auto issue_c = [&](int sq) __attribute__((always_inline)) {
            const int rr = 2 * sq + gr, r = ya + rr;
            const bool rok = (unsigned)r < (unsigned)H && r <= y1;
            dma16(rc, smem + G::OFF_C + ((rr + BIG) % CR) * C_ROWB + gh * 1024, (rok && cok) ? (unsigned)((r * W + x0 - 2) * 32) + cl : DMA_OOB);
        };
        char *strip_b = smem + G::OFF_ST + g * STRIP;
        const int c8 = lane & 3, spx = lane >> 2;
        char *trash = p.trash + tid * 16;
        const int sw0 = l31 * OUT_ROWB + 8 * lh;
        auto store_row = [&](int rr) __attribute__((always_inline)) {   // the strip holds F ring row rr: slots 32 gh .. (slot c = column x0 - 1 + c)
            const int r = ya + rr;
            const bool row_ok = r >= y0 && r < y1;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int c = 32 * gh + it * 16 + spx, ox = c - 1;
                const f16x8 v = *reinterpret_cast<const f16x8 *>(strip_b + (it * 16 + spx) * OUT_ROWB + c8 * 16);
                const bool ok = row_ok && ox >= 0 && ox < WS && x0 + ox < W;
                f16 *d = ok ? p.fea0 + ((size_t)r * W + x0 + ox) * 32 + c8 * 8 : reinterpret_cast<f16 *>(trash);
                *reinterpret_cast<f16x8 *>(d) = v;
            }
        };
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq) issue_c(sq);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            store_row(2 * (s - 1) - 3 + gr);
            issue_c(s + DPF);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            const int rb = 2 * s - 3 + gr;
            int a[3][2];
            const int wb = G::OFF_Y + ((rb - 1 + BIG) % YN) * Y_ROWB;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[kx][ks] = wb + xo[kx][ks];
            const f32x16 acc = conv18<6, Y_ROWB>(wh, smem, a, [](int) {});
            STAMP(1);
            f16x4 z[4];
            const bool in = colf && (unsigned)(ya + rb) < (unsigned)H;     // outside the image: down_conv1's zero padding
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                z[qd] = __builtin_elementwise_max(bias_cvt4(acc, qd, bh[qd]), zero4());
                *reinterpret_cast<f16x4 *>(strip_b + sw0 + 16 * qd) = z[qd];
                if (!in) z[qd] = zero4();
            }
            put_row(smem, G::OFF_F, (rb + BIG) % YN, q0, z);
            STAMP(3);
            // per step and wave: two stores, then one DMA piece: the piece of step s + 1 is older than 3 (DPF - 1) operations
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(3 * (DPF - 1), 0));
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        store_row(2 * (nsteps - 1) - 3 + gr);
        STAMP_DUMP(p);
    }
}

template <int DPF, bool PIPE>
hipError_t launch_rb(RowsRbParams p, int nseg, hipStream_t s)
{
    using G = RbGeo<DPF, PIPE>;
    static DevOnce attr_once;
    auto kern = le_rb_rows_kernel<DPF, PIPE>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(p.nstrips * nseg), dim3(512), G::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

// H, W even; u is [H/2][W/2][32]
hipError_t le_tail_rows_launch(RowsTailParams p, int n_cu, hipStream_t s)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash || (p.H & 1) || (p.W & 1)) return hipErrorInvalidValue;
    using G = TailGeo<3>;
    static DevOnce attr_once;
    auto kern = le_tail_rows_kernel<3>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    p.nstrips = (p.W + WS - 1) / WS;
    int nseg = n_cu / p.nstrips;
    if (nseg < 1) nseg = 1;
    p.rows_per_seg = ((p.H + nseg - 1) / nseg + 1) & ~1;
    nseg = (p.H + p.rows_per_seg - 1) / p.rows_per_seg;
    hipLaunchKernelGGL(kern, dim3(p.nstrips * nseg), dim3(512), G::SMEM, s, p);
    return hipGetLastError();
}

// W even (strips start on even columns: the half-resolution map is cut at x0 / 2); fea1 is [(H + 1) / 2][(W + 1) / 2][32]
hipError_t le_head_rows_launch(RowsHeadParams p, int n_cu, hipStream_t s)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash || (p.W & 1) || (p.H & 1)) return hipErrorInvalidValue;
    using G = HeadGeo<3>;
    static DevOnce attr_once;
    auto kern = le_head_rows_kernel<3>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    p.nstrips = (p.W + WS - 1) / WS;
    int nseg = n_cu / p.nstrips;
    if (nseg < 1) nseg = 1;
    p.rows_per_seg = ((p.H + nseg - 1) / nseg + 1) & ~1;
    nseg = (p.H + p.rows_per_seg - 1) / p.rows_per_seg;
    hipLaunchKernelGGL(kern, dim3(p.nstrips * nseg), dim3(512), G::SMEM, s, p);
    return hipGetLastError();
}

// variant: 0 = sft2 inside its own step (DMA three steps ahead), 1 = sft2 one step behind, inside the next conv1 (DMA two steps ahead)
hipError_t le_rb_rows_launch(RowsRbParams p, int n_cu, hipStream_t s, int variant)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash) return hipErrorInvalidValue;
    p.nstrips = (p.W + WS - 1) / WS;
    int nseg = n_cu / p.nstrips;
    if (nseg < 1) nseg = 1;
    if (nseg > p.H) nseg = p.H;
    p.rows_per_seg = (p.H + nseg - 1) / nseg;
    nseg = (p.H + p.rows_per_seg - 1) / p.rows_per_seg;
    return variant == 1 ? launch_rb<2, true>(p, nseg, s) : launch_rb<3, false>(p, nseg, s);
}
