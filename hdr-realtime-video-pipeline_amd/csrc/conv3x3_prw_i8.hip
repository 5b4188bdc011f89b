// conv3x3_prw_i8.hip -- int8 twin of conv3x3_prw.hip: the W8A8 HG 3x3 convolutions (Cin, Cout multiples of 128 / 256) on
// v_mfma_i32_16x16x64_i8 with the "private weights" schedule.  A 128-channel int8 chunk is the same 128 bytes per pixel as a
// 64-channel f16 chunk, so the halo image, the wave-private weight ring (32 rows x 128 B per tap), the DMA pieces, the
// software pipeline across taps / chunks / tiles and every wait carry over from conv3x3_prw.hip unchanged; one
// v_mfma_i32_16x16x64_i8 consumes the 64 bytes of K that one f16 MFMA did.  The epilogue is conv3x3_pglds_i8.hip's: codes of
// the reading layer's quantiser, clamp(rint(acc * scale + shift), lo, 127), out-of-image halo pixels = code 0 (what an
// LDS-DMA lane outside its buffer writes) with the per-border-class constants `delta` added for pixels on the image border,
// 2x2 max-pool on the (monotone) codes, PixelShuffle addressing.  Exact integer arithmetic: bit-identical to conv_pglds_i8.
#include "launchers.h"

namespace {

constexpr int TW = 16, HW = 18;
constexpr int CT = 128, PIXB = CT;                       // 128-channel int8 chunk = 128 B per pixel
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
__global__ __launch_bounds__(512) void conv_prw_i8_kernel(ConvI8Params p)
{
    constexpr int NPIX = Geo<TH>::NPIX, A_PIECES = Geo<TH>::A_PIECES, A_PIECES_PER_WAVE = Geo<TH>::A_PIECES_PER_WAVE;
    constexpr int NG = TH / 2;                               // groups of two pixel rows per tap
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char *sW = smem + W_OFF + wave * (2 * W_SLOT);

    // ---- this block's run of tiles: XCD x owns a contiguous range, its blocks interleave in it (as conv_pglds) --
    const int ntn = p.Cout / BN;
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
        constexpr bool nt_slow = MODE == ST_PS;           // the Up convs walk Cout-tile slowest (an XCD shares one weight slab)
        const int nt_i = nt_slow ? t / nsp : t % ntn, sp = nt_slow ? t - nt_i * nsp : t / ntn;
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
        const int8_t *src;
        int cs, coff;
        if (cc < nchunk0) { src = p.src0; cs = p.c0; coff = cc * CT; }
        else { src = p.src1; cs = p.c1; coff = (cc - nchunk0) * CT; }
        const rsrc_t rs = make_rsrc(src, (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)cs);
#pragma unroll
        for (int it = 0; it < A_PIECES_PER_WAVE; ++it) {
            const int piece = wave + it * 8;
            const int hp = piece * 8 + l_row;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = T.oy0 - 1 + hy, ix = T.ox0 - 1 + hx;
            const bool ok = (hp < NPIX) & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
            const unsigned off = (unsigned)(iy * p.Wi + ix) * (unsigned)cs + (unsigned)(coff + ((l_slot ^ (hx & 7)) << 4));
            bdma16<0>(rs, piece < A_PIECES ? sA + buf * A_BYTES + piece * 1024 : smem + TRASH_OFF, ok ? off : OOB, 0);
        }
    };
    // this wave's 32 weight rows of (chunk cc, tap): four 1-KiB pieces = one scalar offset + four immediates
    const unsigned w_lane = (unsigned)((lane >> 3) * CT + (((lane & 7) ^ (lane >> 3)) << 4));
    auto issue_W = [&](int cc, int tap, int n0, int slot) {
        const rsrc_t rs = make_rsrc(p.wpk, 9u * (unsigned)nchunk * (unsigned)p.Cout * (unsigned)PIXB);
        const unsigned so = (unsigned)((tap * nchunk + cc) * p.Cout + n0 + wave * WCH) * (unsigned)PIXB;
        char *d = sW + slot * W_SLOT;
        bdma16<0>(rs, d, w_lane, so); bdma16<1024>(rs, d, w_lane, so); bdma16<2048>(rs, d, w_lane, so); bdma16<3072>(rs, d, w_lane, so);
    };
    auto issue_SS = [&](int n0, int slot) {      // wave 0: scale[256], wave 1: shift[256]; the others keep the piece count equal
        const rsrc_t rs = make_rsrc(wave == 1 ? p.shift : p.scale, (unsigned)p.Cout * 4u);
        bdma16<0>(rs, wave < 2 ? smem + SS_OFF + slot * 2048 + wave * 1024 : smem + TRASH_OFF, (unsigned)opaque_lane() * 16u, (unsigned)n0 * 4u);
    };

    i32x4 acc[2][TH];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TH; ++j) acc[i][j] = i32x4{0, 0, 0, 0};

    // ---- the MFMA stream -------------------------------------------------------------------------
    // A tap is TH/2 groups of two pixel rows: group g's four fragment reads are issued in front of group g-1's eight MFMAs
    // (the compiler's counted lgkmcnt waits leave them in flight).  The pipeline runs across taps, chunks and tiles: group 7 of a tap reads the next
    // tap's weight fragments and its first group, behind the vmcnt wait for those weights (issued at the top of this tap)
    // and, in front of a new chunk, the barrier that says every wave's halo pieces have landed.
    i32x4 wf[2][2], xa[2][2], xb[2][2];
    // (fragment addresses are rebuilt from an opaque lane id at every tap: kept loop-invariant they cost a dozen VGPRs)
    auto rd_w = [&](i32x4 (&w)[2][2], int slot, int ln) {
        const int q15 = ln & 15, qg = ln >> 4;
        const char *bw = sW + slot * W_SLOT + q15 * PIXB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                w[ks][i] = *reinterpret_cast<const i32x4 *>(bw + i * 16 * PIXB + (((ks * 4 + qg) ^ (q15 & 7)) << 4));
    };
    // ax: the tap's halo origin (buffer + tap offset, wave-uniform); dx = tap % 3
    auto rd_x = [&](i32x4 (&x)[2][2], const char *ax, int dx, int row0, int ln) {
        const int q15 = ln & 15, qg = ln >> 4, kx = (q15 + dx) & 7;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                x[r][ks] = *reinterpret_cast<const i32x4 *>(ax + q15 * PIXB + (row0 + r) * HW * PIXB + (((ks * 4 + qg) ^ kx) << 4));
    };
    auto mm = [&](const i32x4 (&x)[2][2], int row0) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = row0 + r;
            acc[0][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[0][0], x[r][0], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[0][1], x[r][0], acc[1][j], 0, 0, 0);
            acc[0][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[1][0], x[r][1], acc[0][j], 0, 0, 0);
            acc[1][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[1][1], x[r][1], acc[1][j], 0, 0, 0);
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
                    i32x4 wn[2][2];
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
        // the strip: the weight slot whose fragments (the next tile's first tap) are already in registers (conv3x3_prw.hip)
        char *stg = sW + (ws & 1) * W_SLOT;
        // pixels whose 3x3 window leaves the image get their border class's constant added to the shift; only tiles on the
        // image border take the branch (wave-uniform test first)
        const bool edge_tile = p.delta && (cur.oy0 == 0 || cur.ox0 == 0 || cur.oy0 + TH >= p.Hi || cur.ox0 + TW >= p.Wi);
        float4 sc[2], sh[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sc[i] = *reinterpret_cast<const float4 *>(ss + cw + i * 16);
            sh[i] = *reinterpret_cast<const float4 *>(ss + BN + cw + i * 16);
            sh[i].x += 128.f; sh[i].y += 128.f; sh[i].z += 128.f; sh[i].w += 128.f;
        }
        const float lo128 = p.lo_clamp + 128.f;
        // q = max(acc * scale + shift + 128, lo + 128), still unrounded (the 2x2 max commutes with the monotone quantiser);
        // v_cvt_pk_u8_f32 rounds to nearest even, saturates to [0, 255] and packs; xor 0x80 per byte -> int8 code
        auto qv = [&](int i, int j, float (&q)[4]) {
            float4 s4 = sh[i];
            if (edge_tile) {
                const int oy_ = cur.oy0 + j, ox_ = cur.ox0 + l15;
                const int cls = ((((oy_ == 0) | ((oy_ == p.Hi - 1) << 1)) << 2) | ((ox_ == 0) | ((ox_ == p.Wi - 1) << 1))) & 15;
                if (cls) {
                    const float4 d = *reinterpret_cast<const float4 *>(p.delta + (size_t)cls * p.Cout + cur.n0 + cw + i * 16);
                    s4.x += d.x; s4.y += d.y; s4.z += d.z; s4.w += d.w;
                }
            }
            q[0] = fmaxf((float)acc[i][j][0] * sc[i].x + s4.x, lo128);
            q[1] = fmaxf((float)acc[i][j][1] * sc[i].y + s4.y, lo128);
            q[2] = fmaxf((float)acc[i][j][2] * sc[i].z + s4.z, lo128);
            q[3] = fmaxf((float)acc[i][j][3] * sc[i].w + s4.w, lo128);
            acc[i][j] = i32x4{0, 0, 0, 0};
        };
        auto pack4 = [](const float (&v)[4]) -> unsigned {
            unsigned w = 0;
            w = __builtin_amdgcn_cvt_pk_u8_f32(v[0], 0, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(v[1], 1, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(v[2], 2, w);
            w = __builtin_amdgcn_cvt_pk_u8_f32(v[3], 3, w);
            return w ^ 0x80808080u;
        };
        int8_t *dst = reinterpret_cast<int8_t *>(p.dst);
        const int s_px = eln >> 1, s_half = eln & 1;            // strip read: 32 pixels x two 16-byte halves of their 32 channels
        if constexpr (MODE == ST_NHWC || MODE == ST_PS) {
            const int cps = p.dstC;
            const int chw = cur.n0 + wave * WCH;
            const int sub = MODE == ST_PS ? chw / cps : 0;
            const int cbase = (MODE == ST_PS ? chw - sub * cps : chw) + s_half * 16;
#pragma unroll
            for (int pass = 0; pass < TH / 8; ++pass) {         // eight pixel rows per pass = the 4 KiB strip (32 B per pixel)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        float q[4];
                        qv(i, pass * 8 + jj, q);
                        *reinterpret_cast<unsigned *>(stg + (jj * 16 + l15) * 32 + i * 16 + 4 * kg) = pack4(q);
                    }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int px = rr * 32 + s_px;
                    const i32x4 v = *reinterpret_cast<const i32x4 *>(stg + px * 32 + s_half * 16);
                    const int oy = cur.oy0 + pass * 8 + (px >> 4);
                    const int oxx = cur.ox0 + (px & 15);
                    int8_t *d;
                    if constexpr (MODE == ST_NHWC) {
                        const bool ok = oy < p.Ho && oxx < p.Wo;
                        d = ok ? dst + ((size_t)oy * p.Wo + oxx) * p.dstC + cbase : reinterpret_cast<int8_t *>(trash);
                    } else {
                        const int Y = 2 * oy + (sub >> 1), X = 2 * oxx + (sub & 1);
                        const bool ok = oy < p.Ho && oxx < p.Wo && Y < p.Hd && X < p.Wd;
                        d = ok ? dst + ((size_t)Y * p.Wd + X) * cps + cbase : reinterpret_cast<int8_t *>(trash);
                    }
                    *reinterpret_cast<i32x4 *>(d) = v;
                }
            }
        } else {   // ST_POOL: 2x2 max of the unrounded codes: rows in-lane, columns by a DPP quad swap (no LDS)
#pragma unroll
            for (int jj = 0; jj < TH / 2; ++jj)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float qa[4], qb[4], m[4];
                    qv(i, 2 * jj, qa);
                    qv(i, 2 * jj + 1, qb);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float a = fmaxf(qa[r], qb[r]);
                        const float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, false));
                        m[r] = fmaxf(a, b);
                    }
                    if ((l15 & 1) == 0) *reinterpret_cast<unsigned *>(stg + (jj * 8 + (l15 >> 1)) * 32 + i * 16 + 4 * kg) = pack4(m);
                }
#pragma unroll
            for (int rr = 0; rr < TH / 8; ++rr) {               // 32 pooled pixels (4 rows x 8) per store
                const int px = rr * 32 + s_px;
                const i32x4 v = *reinterpret_cast<const i32x4 *>(stg + px * 32 + s_half * 16);
                const int py = (cur.oy0 >> 1) + (px >> 3), pxx = (cur.ox0 >> 1) + (px & 7);
                const bool ok = py < p.Hd && pxx < p.Wd;
                int8_t *d = ok ? dst + ((size_t)py * p.Wd + pxx) * p.dstC + cur.n0 + wave * WCH + s_half * 16 : reinterpret_cast<int8_t *>(trash);
                *reinterpret_cast<i32x4 *>(d) = v;
            }
        }
        cur = nxt;
    }
}

template <int MODE, int TH>
hipError_t launch_mode(const ConvI8Params &p, int grid, hipStream_t stream)
{
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv_prw_i8_kernel<MODE, TH>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SMEM, stream, p);
    return hipGetLastError();
}

}  // namespace

// 3x3, stride 1, pad 1 on int8 codes: Cin (src0 [+ src1 concat]) multiple of 128, Cout multiple of 256, int8 output codes;
// store modes NHWC / PS / POOL; th = pixel rows per tile (16 or 8).  One block per CU, each walking tiles.
hipError_t conv_prw_i8_launch(ConvI8Params p, int th, int n_cu, hipStream_t stream)
{
    if ((p.c0 % CT) || (p.c1 % CT) || p.c0 + p.c1 < CT || (p.Cout % BN) || !p.trash || n_cu < 8 || p.out_f16 || (th != 8 && th != 16) ||
        (p.mode != ST_NHWC && p.mode != ST_PS && p.mode != ST_POOL) || (p.mode == ST_PS && (p.dstC % 64)))
        return hipErrorInvalidValue;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + th - 1) / th;
    const int total = p.tiles_x * p.tiles_y * (p.Cout / BN);
    const int grid = total < n_cu ? total : n_cu;
    if (th == 16) {
        switch (p.mode) {
        case ST_NHWC: return launch_mode<ST_NHWC, 16>(p, grid, stream);
        case ST_PS: return launch_mode<ST_PS, 16>(p, grid, stream);
        default: return launch_mode<ST_POOL, 16>(p, grid, stream);
        }
    }
    switch (p.mode) {
    case ST_NHWC: return launch_mode<ST_NHWC, 8>(p, grid, stream);
    case ST_PS: return launch_mode<ST_PS, 8>(p, grid, stream);
    default: return launch_mode<ST_POOL, 8>(p, grid, stream);
    }
}
