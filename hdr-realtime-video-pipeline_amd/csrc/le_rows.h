// le_rows.h -- what the row-streaming fused LE kernels share (le_rows.hip: fp16 and fake-quant forms; le_rows_i8.hip: W8A8 layers on
// int8 MFMA): strip geometry, LDS ring layouts and cursors, LDS / LDS-DMA access helpers, the accumulator <-> chunk exchange.
#pragma once
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

// Diagnostic builds only (make EXTRA=-DRB_ABL=n, tools/abl_rows.sh): leave parts of a kernel out to see what its time is
// made of -- 1 no LDS-DMA, 2 no global stores, 4 no conv MFMAs, 8 no SFT passes, 16 no step barrier, 32 the first role idle,
// 64 the second role without its convs.  Results are garbage.
#ifndef RB_ABL
#define RB_ABL 0
#endif
// Schedule knobs (tools/ab_rows.sh builds variants with EXTRA="-DROWS_...=n" on one box): fragment-read lead and PIN mode
// (see conv18) of the convs that carry an SFT pass in their MFMA stream (H) and of the plain ones (F)
#ifndef ROWS_PIN_H
#define ROWS_PIN_H 0
#endif
#ifndef ROWS_PIN_F
#define ROWS_PIN_F 1
#endif
#ifndef ROWS_AHEAD_H
#define ROWS_AHEAD_H 4
#endif
#ifndef ROWS_AHEAD_F
#define ROWS_AHEAD_F 10
#endif

constexpr int WS = 60;                  // output columns of a strip
constexpr int WI = 64;                  // input columns: 2 halo columns each side
constexpr int YP = 66;                  // pixel pitch of the Y rings (fragment reads of the two unused lanes run to slot 65)
constexpr int YN = 6, YPH = YN + 2;     // Y ring rows; physical rows: rows 0 and 1 of a lap are kept a second time behind row 5
constexpr int X_ROWB = WI * 64, C_ROWB = WI * 32, Y_ROWB = YP * 64;
constexpr int BIG = 168;                // multiple of every ring size: keeps (row + BIG) % ring non-negative
constexpr int SFT_TILE_B = 2 * 2 * 16 * 4;   // bytes of one SFT layer's two head bias tiles (2 lane halves x 16 floats each)

// ---- ring layouts.  A ring row holds one pixel per SLOT (64 or 32 bytes) in 16-byte CHUNKS (8 channels); where a pixel's chunk k
// lies is a per-ring choice -- slot(c) = c with bit 0 flipped by a parity of higher bits of c, chunk position = k ^ s(c), s two
// parities of bits of c -- made so that EVERY LDS instruction of the ring's readers and writers is free of bank conflicts
// (tools/lds_bank_model.py restates MI355X_MICROARCH.md's lane groups and bank functions; tools/lds_ring_layouts.py searches the
// layouts and prints the conflict cycles of every access below: 0).  The accesses (16 bytes per lane throughout):
//   R1  MFMA fragment reads, lane (l31, lh) reads chunk 2 ks + lh of pixel c0 + l31 + kx (c0 = 0 / 32, kx = 0..2)  [ds_read_b128: 4 x 16 lanes]
//   S2  the same at stride 2, pixel 2 l31 + kx (the head's down_conv1)
//   W1  chunk writes / reads, lane (l31, lh) chunk q + ... of pixel c0 + l31 (+ 0..2)                              [ds_write_b128: 8 x 8 lanes]
//   W2  chunk writes / reads at stride 2, pixel 2 l31 + gh (the tail's PixelShuffle phases)
// Round 4's layout (s = bits 2..3 of c for every ring) served R1 only: accumulator-layout 8-byte writes and reads were 2-way
// conflicts by construction (the 16 lanes of a ds_write_b64 group share lh, i.e. use half of the 8-byte slots), the stride-2
// accesses 2-way, the output strips 2- and 3-way: SQ_LDS_BANK_CONFLICT 23 - 30 % of SQ_LDS_IDX_ACTIVE (profiles/r04_sq_lds_breakdown.txt).
__device__ __forceinline__ int par(int v) { return __builtin_popcount((unsigned)v) & 1; }
template <int A0, int S0, int S1> struct Lay64 {            // 64-byte pixels (32 channels f16)
    static_assert((A0 & 3) == 0, "slot() must be an involution");
    static __device__ __forceinline__ int slot(int c) { return c ^ par(c & A0); }
    static __device__ __forceinline__ int sw(int c) { return par(c & S0) | (par(c & S1) << 1); }
    static __device__ __forceinline__ unsigned at(int c, int k) { return (unsigned)((slot(c) << 6) | ((k ^ sw(c)) << 4)); }
    // LDS-DMA piece (16 slots; lane i lands at byte 16 i of the piece): the lane's SOURCE pixel and its byte offset in the source row
    static __device__ __forceinline__ int src_px(int piece, int lane) { return slot(16 * piece + (lane >> 2)); }
    static __device__ __forceinline__ unsigned src_off(int piece, int lane) { const int c = src_px(piece, lane); return (unsigned)(c * 64 + (((lane & 3) ^ sw(c)) << 4)); }
};
template <int A0, int S0> struct Lay32 {                    // 32-byte pixels (the 16-channel condition maps)
    static_assert((A0 & 1) == 0, "slot() must be an involution");
    static __device__ __forceinline__ int slot(int c) { return c ^ par(c & A0); }
    static __device__ __forceinline__ unsigned at(int c, int h) { return (unsigned)((slot(c) << 5) | ((h ^ par(c & S0)) << 4)); }
    static __device__ __forceinline__ int src_px(int piece, int lane) { return slot(32 * piece + (lane >> 1)); }
    static __device__ __forceinline__ unsigned src_off(int piece, int lane) { const int c = src_px(piece, lane); return (unsigned)(c * 32 + (((lane & 1) ^ par(c & S0)) << 4)); }
};
using LStd = Lay64<0, 4, 10>;        // R1 + W1: every stride-1 ring (x, Y1, Y2, u, Z, the head's Y)
using LTailY = Lay64<4, 3, 8>;       // R1 + W2: the tail's Y ring (written per PixelShuffle phase, read by HR_conv2)
using LTailF = Lay64<4, 8, 16>;      // W2: the tail's fea0 ring (DMA in, read per PixelShuffle phase)
using LHeadF = Lay64<4, 9, 18>;      // W1 + S2: the head's fea0 ring (written by HR_conv1, read by down_conv1 at stride 2)
using LCond = Lay32<0, 8>;           // condition ring read at stride 1
using LCondT = Lay32<8, 16>;         // ... at stride 2 (the tail)

// LDS reads while an LDS-DMA is in flight: hipcc's waitcnt pass puts s_waitcnt vmcnt(0) in front of every LDS load that
// carries NO alias metadata -- in practice loads of HIP's struct vector types (float4 ...), which are aggregate copies
// without a TBAA tag -- and none in front of loads of clang ext_vector types (f16x8, f32x4: TBAA-tagged; the pass then
// consults its list of DMA stores with alias scopes, which is empty here).  With the DMA running steps ahead a
// vmcnt(0) in the loop drains the whole prefetch queue, so: ext_vector types only for LDS reads inside the step loops
// (tests/test_isa_contracts.py pins the loops' wait sets).
// LDS addresses are plain integers (the dynamic buffer's base folded into lane constants, ring positions in scalar cursors):
typedef __attribute__((address_space(3))) char lds_c;
__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(uintptr_t)(const lds_c *)p; }
template <class T> __device__ __forceinline__ T lds_rd(unsigned a) { return *(const __attribute__((address_space(3))) T *)(uintptr_t)a; }
template <class T> __device__ __forceinline__ void lds_wr(unsigned a, const T &v) { *(__attribute__((address_space(3))) T *)(uintptr_t)a = v; }
__device__ __forceinline__ void dma16_at(dma_rsrc_t r, unsigned lds, unsigned voff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(uintptr_t)lds, 16, voff, 0, 0, 0);
}
// A ring position as a byte offset from the buffer start (ring base included) that advances two rows per step
template <int BASE, int N, int ROWB> struct Cur {
    int o;
    __device__ __forceinline__ explicit Cur(int row) : o(BASE + ((row + BIG) % N) * ROWB) {}
    __device__ __forceinline__ void step() { o += 2 * ROWB; if (o >= BASE + N * ROWB) o -= N * ROWB; }
    __device__ __forceinline__ bool mirrored() const { return o < BASE + 2 * ROWB; }     // rows 0, 1 of a lap: the Y rings' second copy
};

// s_waitcnt immediate of gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | (7 << 4) | ((lgkm & 15) << 8) | ((vm >> 4) << 14); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
// four fp32 -> f16 (round to nearest even) as two v_cvt_pk_f16_f32
__device__ __forceinline__ f16x4 cvt4(float a, float b, float c, float d) { return cvt_h4(a, b, c, d); }      // common.h
// accumulator quad qd (registers 4 qd .. 4 qd + 3) + bias -> f16
__device__ __forceinline__ f16x4 bias_cvt4(const f32x16 &acc, int qd, const f32x4 &b)
{
    // scalar adds on purpose (no f32x2 arithmetic: it becomes v_pk_add_f32, see the Makefile's note on packed f32 beside MFMAs)
    return cvt4(acc[4 * qd] + b[0], acc[4 * qd + 1] + b[1], acc[4 * qd + 2] + b[2], acc[4 * qd + 3] + b[3]);
}
__device__ __forceinline__ f16x4 zero4() { return f16x4{(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f}; }
__device__ __forceinline__ f32x16 zero16() { return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }

// ---- accumulator layout <-> chunk layout.  A 32x32 MFMA leaves lane (l31, lh) with channels 8 qd + 4 lh .. + 3 of pixel l31
// (QUADS, 8 bytes each); memory and the MFMA B operand want 16-byte CHUNKS (8 consecutive channels).  The two lanes of a pixel
// trade halves with v_permlane32_swap (lanes 32-63 of the first register <-> lanes 0-31 of the second; tools/permlane_probe.hip):
// four swaps turn the four quads into chunks lh and 2 + lh -- exactly the fragment of k-step 0 / 1 -- and the same four turn them
// back.  No LDS round trip, no 8-byte LDS access.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void xhalf(unsigned &a, unsigned &b)
{
    const u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
}
__device__ __forceinline__ void quads_to_chunks(const f16x4 (&y)[4], f16x8 &c0, f16x8 &c1)      // c0 = chunk lh, c1 = chunk 2 + lh
{
    u32x2 q0 = __builtin_bit_cast(u32x2, y[0]), q1 = __builtin_bit_cast(u32x2, y[1]), q2 = __builtin_bit_cast(u32x2, y[2]), q3 = __builtin_bit_cast(u32x2, y[3]);
    unsigned a0 = q0[0], a1 = q0[1], b0 = q1[0], b1 = q1[1], d0 = q2[0], d1 = q2[1], e0 = q3[0], e1 = q3[1];
    xhalf(a0, b0); xhalf(a1, b1); xhalf(d0, e0); xhalf(d1, e1);
    c0 = __builtin_bit_cast(f16x8, u32x4{a0, a1, b0, b1});
    c1 = __builtin_bit_cast(f16x8, u32x4{d0, d1, e0, e1});
}
__device__ __forceinline__ void chunks_to_quads(const f16x8 &c0, const f16x8 &c1, f16x4 (&y)[4])
{
    const u32x4 u = __builtin_bit_cast(u32x4, c0), v = __builtin_bit_cast(u32x4, c1);
    unsigned a0 = u[0], a1 = u[1], b0 = u[2], b1 = u[3], d0 = v[0], d1 = v[1], e0 = v[2], e1 = v[3];
    xhalf(a0, b0); xhalf(a1, b1); xhalf(d0, e0); xhalf(d1, e1);
    y[0] = __builtin_bit_cast(f16x4, u32x2{a0, a1}); y[1] = __builtin_bit_cast(f16x4, u32x2{b0, b1});
    y[2] = __builtin_bit_cast(f16x4, u32x2{d0, d1}); y[3] = __builtin_bit_cast(f16x4, u32x2{e0, e1});
}

__device__ __forceinline__ f16x8 lrelu_pack16(const f32x16 &a, int s)
{
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)a[8 * s + j];
    return __builtin_elementwise_max(o, o * (f16)0.1f);
}

// this lane's fragment addresses for output slot c of a ring row in layout L: input slots c .. c + 2 (row 0, base `b`)
template <class L> __device__ __forceinline__ void frag_addr(unsigned (&va)[3][2], unsigned b, int c, int lh, int stride = 1)
{
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) va[kx][ks] = b + L::at(stride * c + kx, (ks << 1) | lh);
}
// this lane's two chunks (lh and 2 + lh, the pair quads_to_chunks produces) of slot c
template <class L> __device__ __forceinline__ void chunk_addr(unsigned (&vc)[2], unsigned b, int c, int lh)
{
    vc[0] = b + L::at(c, lh); vc[1] = b + L::at(c, 2 + lh);
}
// One pixel's 16 channels of this lane (accumulator layout) into a Y ring row at byte offset `off` (and into the row's second copy)
__device__ __forceinline__ void put_row(const unsigned (&vc)[2], int off, bool mirror, const f16x4 (&y)[4])
{
    f16x8 c0, c1;
    quads_to_chunks(y, c0, c1);
    lds_wr(vc[0] + (unsigned)off, c0); lds_wr(vc[1] + (unsigned)off, c1);
    if (mirror) { lds_wr(vc[0] + (unsigned)(off + YN * Y_ROWB), c0); lds_wr(vc[1] + (unsigned)(off + YN * Y_ROWB), c1); }
}
// ... and back: slot c of a ring row at byte offset `off` as the lane's four quads
__device__ __forceinline__ void get_row(const unsigned (&vc)[2], int off, f16x4 (&y)[4])
{
    chunks_to_quads(lds_rd<f16x8>(vc[0] + (unsigned)off), lds_rd<f16x8>(vc[1] + (unsigned)off), y);
}

template <class P> void strips(P &p, int n_cu, bool even_rows, int &nseg)
{
    p.nstrips = (p.W + WS - 1) / WS;
    nseg = n_cu / p.nstrips;
    if (nseg < 1) nseg = 1;
    if (nseg > p.H) nseg = p.H;
    p.rows_per_seg = (p.H + nseg - 1) / nseg;
    if (even_rows) p.rows_per_seg = (p.rows_per_seg + 1) & ~1;
    nseg = (p.H + p.rows_per_seg - 1) / p.rows_per_seg;
}
template <class K> hipError_t set_lds(K kern, int bytes, DevOnce &once)
{
    if (once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        once.done();
    }
    return hipSuccess;
}


}  // namespace
