// le_rows.hip -- row-streaming fused kernels of the LE main branch (gfx950): whole layer chains in ONE launch, every
// intermediate tensor in LDS rings of a few pixel rows, nothing but the chain's inputs and outputs in HBM.
//
// Reference: ResBlock_with_SFT.forward (arch_util.py:89-95)  y = x + conv2(sft2(relu(conv1(sft1(x, c))), c)),
// SFTLayer.forward (arch_util.py:66-72); the 16x16-tile kernels they replace are conv32s.hip's (two launches per block,
// the 32-channel intermediate written to and read back from HBM, every input halo fetched 1.27x).
//
// Schedule.  A workgroup owns a STRIP of 60 output columns and a SEGMENT of rows and walks down it two rows per step.
// The stages of the chain run skewed against each other, each on rows the previous stage finished a step earlier:
//     step s:   LDS-DMA of rows 2s+8, 2s+9 (x: 64 px x 64 B, cond: 64 px x 32 B; four steps ahead)
//               sft1           -> Y1 rows 2s,   2s+1          (64 columns: the strip + 2 halo columns each side)
//               conv1 + sft2   -> Y2 rows 2s-3, 2s-2          (62 columns)
//               conv2 + x      -> out rows 2s-6, 2s-5         (60 columns; x from the ring the DMA filled)
// so a row is fetched ONCE (plus 4 of 64 columns shared with the neighbour strips and 4 rows per segment), there is no
// vertical recompute, and ONE s_barrier per step orders all rings (every ring slot is written and read in different steps).
// Waves have ROLES, so that a wave's 3x3 filter bank lives in its registers for the whole launch (18 A fragments = 72
// VGPRs; conv32s re-reads it from LDS for every 32 pixels): waves 0-3 run conv1 + sft2, waves 4-7 sft1 + conv2 + the
// stores, one 32-pixel group (row g >> 1, column half g & 1) per stage, step and wave.  Each SIMD holds one wave of
// either role; they meet only at the barrier.
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

constexpr int WS = 60;                  // output columns of a strip
constexpr int WI = 64;                  // input columns: 2 halo columns each side
constexpr int YP = 68;                  // pixel pitch of the Y rings (fragment reads of the two unused lanes run to slot 65)
constexpr int DPF = 4;                  // the LDS-DMA runs DPF steps ahead
constexpr int XR = 16, CR = 14, YR = 6; // ring rows (see header: live ranges 2 DPF + 8, 2 DPF + 5 (even), 6)
constexpr int X_ROWB = WI * 64, C_ROWB = WI * 32, Y_ROWB = YP * 64;
constexpr int OFF_X = 0, OFF_C = OFF_X + XR * X_ROWB, OFF_Y1 = OFF_C + CR * C_ROWB, OFF_Y2 = OFF_Y1 + YR * Y_ROWB;
constexpr int OUT_ROWB = 64 + 16, STRIP = 32 * OUT_ROWB;
constexpr int OFF_ST = OFF_Y2 + YR * Y_ROWB;
constexpr int OFF_B = OFF_ST + 4 * STRIP;            // conv1 / conv2 bias (in the dynamic buffer: hipcc guards every read of a
                                                     // NAMED LDS array with vmcnt(0) while an LDS-DMA is in flight)
constexpr int SMEM_RB = OFF_B + 256;
static_assert(SMEM_RB <= 160 * 1024, "LDS budget");
constexpr int BIG = 336;                // multiple of every ring size: keeps (row + BIG) % ring non-negative

__device__ __forceinline__ int swz32(int v) { return (v >> 2) & 3; }

// LDS reads while an LDS-DMA is in flight: hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of every LDS load that
// carries NO alias metadata -- in practice loads of HIP's struct vector types (float4 ...), which are aggregate copies
// without a TBAA tag -- and none in front of loads of clang ext_vector types (f16x8, f32x4: TBAA-tagged; the pass then
// consults its list of DMA stores with alias scopes, which is empty here).  With the DMA running four steps ahead a
// vmcnt(0) in the loop drains the whole prefetch queue, so: ext_vector types only for LDS reads inside the step loop
// (tests/test_isa_contracts.py pins the loop's wait set).

// s_waitcnt immediate of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

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

// A 3x3 32 -> 32 filter bank as 18 A fragments of v_mfma_f32_32x32x16_f16 (wpk = [tap][32 out][32 in]; fragment (tap, ks):
// lane = out channel l31, input channels 16 ks + 8 lh ..)
struct Bank { f16x8 f[18]; };
__device__ __forceinline__ void load_bank(Bank &b, const f16 *wpk, int l31, int lh)
{
#pragma unroll
    for (int st = 0; st < 18; ++st)
        b.f[st] = *reinterpret_cast<const f16x8 *>(wpk + ((st >> 1) * 32 + l31) * 32 + 16 * (st & 1) + 8 * lh);
}

// The SFT layer's operands (pack_sft, hdrtv_api.hip): hidden stack, scale head, shift head; biases as accumulator tiles
struct Sft { f16x8 a0, a1s, a1t; f32x16 bh, bs, bt; };
__device__ __forceinline__ void load_sft(Sft &s, const f16 *wfrag, const float *bias, int lane, int lh)
{
    const f16x8 *fr = reinterpret_cast<const f16x8 *>(wfrag);
    s.a0 = fr[lane]; s.a1s = fr[64 + lane]; s.a1t = fr[128 + lane];
    s.bh = tile16(bias, lh); s.bs = tile16(bias + 32, lh); s.bt = tile16(bias + 64, lh);
#pragma unroll
    for (int k = 0; k < 16; ++k) s.bs[k] += 1.f;           // (scale + 1) enters through the accumulator init
}
// y = x * (scale + 1) + shift on one pixel's 16 channels of this lane (channel quads qd: channels 8 qd + 4 lh ..), conv32s's form
__device__ __forceinline__ void sft_apply(const Sft &s, const f16x8 &c0, f16x4 (&y)[4])
{
    const f32x16 h = __builtin_amdgcn_mfma_f32_32x32x16_f16(s.a0, c0, s.bh, 0, 0, 0);
    const f16x8 hs = lrelu_pack16(h, 0), ht = lrelu_pack16(h, 1);
    const f32x16 sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(s.a1s, hs, s.bs, 0, 0, 0);
    const f32x16 sh = __builtin_amdgcn_mfma_f32_32x32x16_f16(s.a1t, ht, s.bt, 0, 0, 0);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        f16x4 s1, s0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s1[k] = (f16)sc[4 * qd + k]; s0[k] = (f16)sh[4 * qd + k]; }
        y[qd] = y[qd] * s1 + s0;
    }
}

// 3x3 conv of one 32-pixel group: rows r0..r2 are the LDS bases of the three input rows, xo[kx][ks] this lane's fragment
// offsets inside a row.  K order (tap, k-step); reads run three steps ahead of the MFMAs.
__device__ __forceinline__ f32x16 conv18(const Bank &w, const char *r0, const char *r1, const char *r2, const int (&xo)[3][2])
{
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    f16x8 x[18];
    auto ld = [&](int st) __attribute__((always_inline)) {
        const int tap = st >> 1, ks = st & 1, ky = tap / 3, kx = tap % 3;
        x[st] = *reinterpret_cast<const f16x8 *>((ky == 0 ? r0 : (ky == 1 ? r1 : r2)) + xo[kx][ks]);
    };
    ld(0); ld(1); ld(2);
#pragma unroll
    for (int st = 0; st < 18; ++st) {
        if (st + 3 < 18) ld(st + 3);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.f[st], x[st], acc, 0, 0, 0);
    }
    return acc;
}

}  // namespace

// Fused ResBlock_with_SFT, rows.  Grid = nstrips x nseg workgroups of 512 threads.
__global__ __launch_bounds__(512) void le_rb_rows_kernel(RowsRbParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int strip = blockIdx.x % p.nstrips, seg = blockIdx.x / p.nstrips;
    const int x0 = strip * WS;
    const int y0 = seg * p.rows_per_seg, y1 = min(y0 + p.rows_per_seg, p.H);
    const int ya = y0 - 2;                                             // image row of ring row 0
    const int nsteps = (y1 - ya + 5) / 2 + 1;
    const int H = p.H, W = p.W;
    float *sB = reinterpret_cast<float *>(smem + OFF_B);
    if (tid < 32) { sB[tid] = p.b1[tid]; sB[32 + tid] = p.b2[tid]; }

    const int g = wave & 3, gr = g >> 1, gh = g & 1;                   // this wave's 32-pixel group: row gr of the step's pair, column half gh
    const dma_rsrc_t rx = dma_rsrc(p.x), rc = dma_rsrc(p.cond);

    if (wave < 4) {
        // ------------------------------------------------------------------ role B: conv1 + sft2, issues the x rows
        Bank w1;
        load_bank(w1, p.w1, l31, lh);
        Sft s2;
        load_sft(s2, p.sft2_wfrag, p.sft2_bias, lane, lh);
        const int cx = 32 * gh + l31;                                  // Y2 slot = conv1 output column x0 - 1 + cx
        int xo[3][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xo[kx][ks] = (cx + kx) * 64 + ((((ks << 1) | lh) ^ swz32(cx + kx)) << 4);
        const int yw = cx * 64 + (swz32(cx) << 4) + 8 * lh;            // Y2 write: channel quad qd at yw ^ (qd << 4)
        const int co = (cx + 1) * 32 + ((lh ^ (((cx + 1) >> 3) & 1)) << 4);     // condition pixel of the same column (ring slot cx + 1)
        const bool col_in = (unsigned)(x0 - 1 + cx) < (unsigned)W;
        // LDS-DMA of the x rows: pieces 2 gh, 2 gh + 1 of row gr (16 pixels x 64 B each)
        unsigned xl[2];
        bool xok[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int px = 16 * (2 * gh + j) + (lane >> 2), slot = lane & 3;
            xl[j] = (unsigned)(px * 64 + ((slot ^ swz32(px)) << 4));
            xok[j] = (unsigned)(x0 - 2 + px) < (unsigned)W;
        }
        auto issue_x = [&](int sq) __attribute__((always_inline)) {
            const int rr = 2 * sq + gr, r = ya + rr;
            const bool rok = (unsigned)r < (unsigned)H && r <= y1 + 1;
            const unsigned base = (unsigned)((r * W + x0 - 2) * 64);
            char *d = smem + OFF_X + ((rr + BIG) % XR) * X_ROWB + (2 * gh) * 1024;
#pragma unroll
            for (int j = 0; j < 2; ++j) dma16(rx, d + j * 1024, (rok && xok[j]) ? base + xl[j] : DMA_OOB);
        };
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq) issue_x(sq);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            issue_x(s + DPF);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            const int rr = 2 * s - 3 + gr, r = ya + rr;                // the row this wave convolves
            const f16x8 c0 = *reinterpret_cast<const f16x8 *>(smem + OFF_C + ((rr + BIG) % CR) * C_ROWB + co);
            const char *y1b = smem + OFF_Y1;
            const f32x16 acc = conv18(w1, y1b + ((rr - 1 + BIG) % YR) * Y_ROWB, y1b + ((rr + BIG) % YR) * Y_ROWB,
                                      y1b + ((rr + 1 + BIG) % YR) * Y_ROWB, xo);
            STAMP(1);
            f16x4 y[4];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 b4 = *reinterpret_cast<const f32x4 *>(sB + 8 * qd + 4 * lh);
                y[qd][0] = (f16)act_fast(acc[4 * qd + 0] + b4[0], 0.f); y[qd][1] = (f16)act_fast(acc[4 * qd + 1] + b4[1], 0.f);
                y[qd][2] = (f16)act_fast(acc[4 * qd + 2] + b4[2], 0.f); y[qd][3] = (f16)act_fast(acc[4 * qd + 3] + b4[3], 0.f);
            }
            sft_apply(s2, c0, y);
            const bool in = col_in && (unsigned)r < (unsigned)H;       // outside the image: conv2's zero padding
            char *yd = smem + OFF_Y2 + ((rr + BIG) % YR) * Y_ROWB;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                if (!in) y[qd] = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                *reinterpret_cast<f16x4 *>(yd + (yw ^ (qd << 4))) = y[qd];
            }
            STAMP(2);
            // the rows of step s + 1 were issued DPF - 1 steps ago: all but the 2 (DPF - 1) youngest pieces have landed
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 * (DPF - 1), 0));
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    } else {
        // ------------------------------------------------------------------ role C: sft1, conv2 + x, stores; issues the cond rows
        Bank w2;
        load_bank(w2, p.w2, l31, lh);
        Sft s1;
        load_sft(s1, p.sft1_wfrag, p.sft1_bias, lane, lh);
        const int cx = 32 * gh + l31;                                  // sft1: Y1 / x / cond ring slot = image column x0 - 2 + cx
        const int xr0 = cx * 64 + (swz32(cx) << 4) + 8 * lh;           // x read and Y1 write: quad qd at xr0 ^ (qd << 4)
        const int co = cx * 32 + ((lh ^ ((cx >> 3) & 1)) << 4);
        const bool col_in = (unsigned)(x0 - 2 + cx) < (unsigned)W;
        int xo[3][2];                                                  // conv2: output column x0 + cx reads Y2 slots cx .. cx + 2
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xo[kx][ks] = (cx + kx) * 64 + ((((ks << 1) | lh) ^ swz32(cx + kx)) << 4);
        // LDS-DMA of the condition rows: piece gh of row gr (32 pixels x 32 B)
        const int cpx = 32 * gh + (lane >> 1);
        const unsigned cl = (unsigned)(cpx * 32 + (((lane & 1) ^ ((cpx >> 3) & 1)) << 4));
        const bool cok = (unsigned)(x0 - 2 + cpx) < (unsigned)W;
        auto issue_c = [&](int sq) __attribute__((always_inline)) {
            const int rr = 2 * sq + gr, r = ya + rr;
            const bool rok = (unsigned)r < (unsigned)H && r <= y1 + 1;
            dma16(rc, smem + OFF_C + ((rr + BIG) % CR) * C_ROWB + gh * 1024, (rok && cok) ? (unsigned)((r * W + x0 - 2) * 32) + cl : DMA_OOB);
        };
        // epilogue: this lane's two 16-byte chunks = pixels (it * 16 + (lane >> 2)) of the group, channel chunk c8
        char *strip_b = smem + OFF_ST + g * STRIP;
        const int c8 = lane & 3, spx = lane >> 2;
        char *trash = p.trash + tid * 16;
#pragma unroll
        for (int sq = 0; sq < DPF; ++sq) issue_c(sq);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
        __builtin_amdgcn_s_barrier();
        STAMP_DECL;
        for (int s = 0; s < nsteps; ++s) {
            STAMP(7);
            issue_c(s + DPF);
            __builtin_amdgcn_sched_barrier(0);
            STAMP(0);
            {   // sft1 on row 2 s + gr
                const int rr = 2 * s + gr, r = ya + rr;
                const char *xb = smem + OFF_X + ((rr + BIG) % XR) * X_ROWB;
                const f16x8 c0 = *reinterpret_cast<const f16x8 *>(smem + OFF_C + ((rr + BIG) % CR) * C_ROWB + co);
                f16x4 y[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) y[qd] = *reinterpret_cast<const f16x4 *>(xb + (xr0 ^ (qd << 4)));
                sft_apply(s1, c0, y);
                const bool in = col_in && (unsigned)r < (unsigned)H;   // outside the image: conv1's zero padding
                char *yd = smem + OFF_Y1 + ((rr + BIG) % YR) * Y_ROWB;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    if (!in) y[qd] = f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
                    *reinterpret_cast<f16x4 *>(yd + (xr0 ^ (qd << 4))) = y[qd];
                }
            }
            STAMP(2);
            {   // conv2 on row 2 s - 6 + gr, + x, store
                const int rr = 2 * s - 6 + gr, r = ya + rr;
                const char *y2b = smem + OFF_Y2;
                const f32x16 acc = conv18(w2, y2b + ((rr - 1 + BIG) % YR) * Y_ROWB, y2b + ((rr + BIG) % YR) * Y_ROWB,
                                          y2b + ((rr + 1 + BIG) % YR) * Y_ROWB, xo);
                STAMP(1);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4 *>(sB + 32 + 8 * qd + 4 * lh);
                    f16x4 o;
                    o[0] = (f16)(acc[4 * qd + 0] + b4[0]); o[1] = (f16)(acc[4 * qd + 1] + b4[1]);
                    o[2] = (f16)(acc[4 * qd + 2] + b4[2]); o[3] = (f16)(acc[4 * qd + 3] + b4[3]);
                    *reinterpret_cast<f16x4 *>(strip_b + l31 * OUT_ROWB + (8 * qd + 4 * lh) * 2) = o;
                }
                const char *xb = smem + OFF_X + ((rr + BIG) % XR) * X_ROWB;
                const bool row_ok = r >= y0 && r < y1;
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int ox = 32 * gh + it * 16 + spx, xs = ox + 2;          // output column x0 + ox; its x lives in ring slot ox + 2
                    f16x8 v = *reinterpret_cast<const f16x8 *>(strip_b + (it * 16 + spx) * OUT_ROWB + c8 * 16);
                    const f16x8 res = *reinterpret_cast<const f16x8 *>(xb + xs * 64 + ((c8 ^ swz32(xs)) << 4));
                    v = v + res;
                    const bool ok = row_ok && ox < WS && x0 + ox < W;
                    f16 *d = ok ? p.dst + ((size_t)r * W + x0 + ox) * 32 + c8 * 8 : reinterpret_cast<f16 *>(trash);
                    *reinterpret_cast<f16x8 *>(d) = v;
                }
            }
            STAMP(3);
            // per step and wave: one DMA piece, then two stores (always issued): the piece of step s + 1 is older than
            // 3 (DPF - 1) + 2 operations
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(3 * (DPF - 1) + 2, 0));
            STAMP(4);
            __builtin_amdgcn_s_barrier();
            STAMP(5);
        }
        STAMP_DUMP(p);
    }
}

hipError_t le_rb_rows_launch(RowsRbParams p, int n_cu, hipStream_t s)
{
    if ((size_t)p.H * p.W * 64 >= 0x7f000000ull || !p.trash) return hipErrorInvalidValue;
    static DevOnce attr_once;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(le_rb_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_RB);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    p.nstrips = (p.W + WS - 1) / WS;
    int nseg = n_cu / p.nstrips;
    if (nseg < 1) nseg = 1;
    if (nseg > p.H) nseg = p.H;
    p.rows_per_seg = (p.H + nseg - 1) / nseg;
    nseg = (p.H + p.rows_per_seg - 1) / p.rows_per_seg;
    hipLaunchKernelGGL(le_rb_rows_kernel, dim3(p.nstrips * nseg), dim3(512), SMEM_RB, s, p);
    return hipGetLastError();
}
