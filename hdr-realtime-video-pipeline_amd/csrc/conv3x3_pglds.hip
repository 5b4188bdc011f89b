// conv3x3_pglds.hip -- persistent version of the HG head's 3x3 convolutions (Cin multiple of 64,
// Cout multiple of 128, stride 1; Hallucination_arch.py:97-137 conv2..conv_code2, Up_conv1..5).
//
// Same GEMM view, LDS images and LDS-DMA pipeline as conv3x3_glds.hip (128 output channels x
// 16x16-pixel tile, K = 64-channel chunk x tap, halo tile staged once per chunk, weights in a 3-slot
// ring two taps ahead, one raw s_barrier per tap behind a counted s_waitcnt vmcnt), but one block
// per CU walks a run of tiles and the (tile, chunk, tap) iterations form ONE stream:
//   * the next tile's first halo chunk, its first weight taps and its 128 scale/shift pairs are
//     DMA'd during the current tile's last taps, so a tile costs no prologue latency and no block
//     launch (the K = 9 layers conv2 / Up_conv5 spent ~40 % of their time there);
//   * the epilogue runs out of the accumulator registers: scale/shift from LDS, pooling / pixel
//     shuffle / the fused 64->3 dot products in registers, then a transposition through a
//     wave-private strip of the halo buffer the tile has just released (LDS operations of one wave
//     execute in order: no barrier) so that global stores are 16 bytes per lane and 8 lanes cover a
//     pixel's 128-byte channel run; the other tile buffers stay free for the DMAs already in
//     flight.  Every wave issues the SAME number of stores per tile
//     (out-of-image lanes write a trash line instead of being masked): vmcnt counts loads, stores
//     and LDS-DMA together in issue order, so the counted wait after a tile boundary depends on it.
// MFMA shape 16x16x32 (lane = row & 15, k-group = lane >> 4): under load the chip holds a higher
// clock on it than on 32x32x16 (MI355X_MICROARCH.md, DVFS give-back item 7; measured here +4 %).
#include "launchers.h"

namespace {

constexpr int TH = 16, TW = 16, HW = 18, NPIX = HW * HW;
constexpr int CT = 64, PIXB = CT * 2;                    // 64-channel chunk = 128 B per pixel
constexpr int BN = 128;
constexpr int A_PIECES_PER_WAVE = 6, A_BYTES = 8 * A_PIECES_PER_WAVE * 1024;   // 324 halo px -> 48 KiB
constexpr int B_BYTES = BN * PIXB, B_PIECES_PER_WAVE = 2;                       // 16 KiB
constexpr int SS_OFF = 2 * A_BYTES + 3 * B_BYTES;        // two 1-KiB {scale[128], shift[128]} slots
constexpr int DOTW_OFF = SS_OFF + 2048;                  // ST_PS_DOT3: 3 x 64 floats
constexpr int SMEM = DOTW_OFF + 1024;                    // 147 KiB

// stores per wave and tile, by store mode (see the epilogues)
template <int MODE> struct NStores { static constexpr int N = MODE == ST_POOL ? 2 : (MODE == ST_PS_DOT3 ? 1 : 8); };

// LDS-DMA as a buffer load (see conv3x3_prw.hip): counted waits from hipcc, zeros for out-of-range lanes
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void bdma16(rsrc_t r, void *lds, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, 0, 0);
}
constexpr unsigned OOB = 0x80000000u;

template <int N> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Tile { int n0, oy0, ox0; };

template <int MODE>
__global__ __launch_bounds__(512) void conv_pglds_kernel(ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;
    char *sB = smem + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;

    // ---- this block's run of tiles: XCD x owns a contiguous range, its blocks interleave in it --
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
    const int nit = nchunk * 9;

    // ---- LDS-DMA issue helpers (wave-uniform LDS base, per-lane swizzled source) ------------
    const int l_row = lane >> 3, l_slot = lane & 7;
    auto issue_A = [&](int cc, int buf, const Tile &T) {
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
            bdma16(rs, sA + buf * A_BYTES + piece * 1024, ok ? off : OOB, 0);
        }
    };
    auto issue_B = [&](int it_i, int n0, int slot) {
        const int cc = it_i / 9, tap = it_i - cc * 9;
        const rsrc_t rs = make_rsrc(p.wpk, 9u * (unsigned)nchunk * (unsigned)p.CoutPad * (unsigned)PIXB);
        const unsigned so = (unsigned)((tap * nchunk + cc) * p.CoutPad + n0) * (unsigned)PIXB;
#pragma unroll
        for (int k = 0; k < B_PIECES_PER_WAVE; ++k) {
            const int piece = wave * B_PIECES_PER_WAVE + k;
            const int n = piece * 8 + l_row;
            bdma16(rs, sB + slot * B_BYTES + piece * 1024, (unsigned)(n * CT + ((l_slot ^ (n & 7)) << 3)) * 2u, so);
        }
    };
    auto issue_SS = [&](int n0, int slot) {      // every wave writes the same 1 KiB: {scale[128], shift[128]}
        const char *sc = reinterpret_cast<const char *>(p.scale), *sh = reinterpret_cast<const char *>(p.shift);
        const char *lo = sc < sh ? sc : sh;                      // one (wave-uniform) resource over both arrays
        const rsrc_t rs = make_rsrc(lo, 0xffffffffu);
        bdma16(rs, smem + SS_OFF + slot * 1024, (unsigned)((lane < 32 ? sc : sh) - lo) + (unsigned)(lane & 31) * 16u, (unsigned)n0 * 4u);
    };

    if constexpr (MODE == ST_PS_DOT3) {          // before any DMA is in flight (ordinary loads drain the queue)
        float *s_w = reinterpret_cast<float *>(smem + DOTW_OFF);
        for (int e = tid; e < 3 * 64; e += 512) s_w[e] = p.dotw[e];
        __syncthreads();
    }

    // ---- wave tiling: 2 (channels) x 4 (pixel rows) waves, each 64 ch x 64 px = 4x4 tiles of 16x16
    const int wc = wave & 1, wp = wave >> 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ReLU or none (the launcher refuses anything else): one v_max against 0 or -inf instead of max(v, slope * v)
    // (A/B on one box, four builds: 6.89 -> 6.82 ms over the 14 launches)
    const float act_lb = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    const int kw = l15 & 7;
    const int b_lane = (wc * 64 + l15) * PIXB;
    const int a_lane = (wp * 4 * HW + l15) * PIXB;

    // ---- prologue: first tile's halo, scale/shift and weights of taps 0 and 1 -----------------
    Tile cur = decode(t_first), nxt = cur;
    issue_A(0, 0, cur);
    issue_SS(cur.n0, 0);
    issue_B(0, cur.n0, 0);
    issue_B(1, cur.n0, 1);
    issue_B(2, cur.n0, 2);
    wait_vm<4>();
    __builtin_amdgcn_s_barrier();

    int gch = 0;                                  // chunks done so far: halo buffer parity
    for (int k = 0; k < ntile; ++k) {
        const bool has_next = k + 1 < ntile;
        if (has_next) nxt = decode(t_first + (k + 1) * t_step);
        for (int cc = 0; cc < nchunk; ++cc, ++gch) {
            const char *a = sA + (gch & 1) * A_BYTES;
            const bool last_chunk = cc + 1 == nchunk;
            const bool pfA = !last_chunk || has_next;    // a halo tile is staged during this chunk's tap 6
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int it_i = cc * 9 + tap;
                // stream index = 9 * gch + tap, so the ring slot of this iteration is tap % 3
                // (a tile's weights(2) are issued before the tile starts: prologue / end of the previous tile)
                bool pfB = true;
                if (tap == 0 && cc == 0) {}
                else if (it_i + 2 < nit) issue_B(it_i + 2, cur.n0, (tap + 2) % 3);
                else if (has_next) issue_B(it_i + 2 - nit, nxt.n0, (tap + 2) % 3);
                else pfB = false;
                if (tap == 6 && pfA) {
                    if (!last_chunk) issue_A(cc + 1, (gch + 1) & 1, cur);
                    else { issue_A(0, (gch + 1) & 1, nxt); issue_SS(nxt.n0, (k + 1) & 1); }
                }

                const char *bw = sB + (tap % 3) * B_BYTES + b_lane;
                const char *ax = a + a_lane + ((tap / 3) * HW + tap % 3) * PIXB;
                const int kx = (l15 + tap % 3) & 7;
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
                    acc[g >> 2][g & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][g >> 2], xf[0][g & 3], acc[g >> 2][g & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // k-step 1 fragments in the order its MFMAs want them: w0 x0 x1 x2 x3 w1 w2 w3
                    if (g == 0) ldw(1, 0); else if (g < 5) ldx(1, g - 1); else ldw(1, g - 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 8; m < 16; ++m)
                    acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0][m >> 2], xf[0][m & 3], acc[m >> 2][m & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1][m >> 2], xf[1][m & 3], acc[m >> 2][m & 3], 0, 0, 0);

                // The next iteration reads weights(s+1), issued one iteration ago, and at a chunk boundary
                // the halo staged at tap 6.  Allow exactly the DMAs (and, right after a tile boundary, the
                // previous tile's stores) that are younger than those.
                if (!pfB) {
                    wait_vm<0>();
                } else if ((tap == 6 || tap == 7) && pfA) {
                    if (last_chunk) wait_vm<9>();        // halo (6) + scale/shift (1) + weights(s+2) (2)
                    else wait_vm<8>();
                } else if (tap <= 1 && cc == 0 && k > 0) {
                    wait_vm<NStores<MODE>::N + 2>();     // weights(s+1) are older than the last tile's stores
                } else {
                    wait_vm<2>();
                }
                __builtin_amdgcn_s_barrier();
            }
        }

        // weights(2) of the next tile go out BEFORE this tile's stores: vmcnt retires in issue order, so the
        // first DMA wait that has to see the stores complete is then three taps away instead of one
        if (has_next) issue_B(2, nxt.n0, 2);
        // ------------------------------------------------------------ epilogue, from registers
        // lane: pixel (row wp*4 + j, column l15), channels wc*64 + i*16 + 4*kg + {0..3}
        const float *ss = reinterpret_cast<const float *>(smem + SS_OFF + (k & 1) * 1024);
        char *trash = reinterpret_cast<char *>(p.trash) + lane * 16;
        const int cw = wc * 64 + 4 * kg;
        f16x4 o[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 sc = *reinterpret_cast<const float4 *>(ss + cw + i * 16);
            const float4 sh = *reinterpret_cast<const float4 *>(ss + 128 + cw + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[i][j][0] = (f16)fmaxf(acc[i][j][0] * sc.x + sh.x, act_lb);
                o[i][j][1] = (f16)fmaxf(acc[i][j][1] * sc.y + sh.y, act_lb);
                o[i][j][2] = (f16)fmaxf(acc[i][j][2] * sc.z + sh.z, act_lb);
                o[i][j][3] = (f16)fmaxf(acc[i][j][3] * sc.w + sh.w, act_lb);
                acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const int ox = cur.ox0 + l15;   // (ST_PS_DOT3)
        // NHWC / PS / POOL: transpose through a wave-private strip of the halo buffer this tile just finished
        // with (free until the next tile's tap 6), so that global stores are 16 bytes per lane and 8 lanes
        // cover a pixel's 128-byte channel run (8-byte quads straight from the accumulator layout cost ~25 %
        // of a K = 9 layer).  LDS operations of one wave execute in order: no barrier.
        constexpr int SP = 144;                                  // strip row pitch: 64 ch x 2 B + 16
        char *stg = sA + ((gch - 1) & 1) * A_BYTES + wave * (32 * SP);
        const int s_row = lane >> 3, s_chunk = lane & 7;
        if constexpr (MODE == ST_NHWC || MODE == ST_PS) {
            const int cps = p.dstC;
            const int chw = cur.n0 + wc * 64;
            const int sub = MODE == ST_PS ? chw / cps : 0;
            const int cbase = (MODE == ST_PS ? chw - sub * cps : chw) + s_chunk * 8;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<f16x4 *>(stg + (jj * 16 + l15) * SP + (i * 16 + 4 * kg) * 2) = o[i][2 * pass + jj];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const f16x8 v = *reinterpret_cast<const f16x8 *>(stg + (rr * 8 + s_row) * SP + s_chunk * 16);
                    const int oy = cur.oy0 + wp * 4 + 2 * pass + (rr >> 1);
                    const int oxx = cur.ox0 + (rr & 1) * 8 + s_row;
                    f16 *d;
                    if constexpr (MODE == ST_NHWC) {
                        const bool ok = oy < p.Ho && oxx < p.Wo;
                        d = ok ? p.dst + ((size_t)oy * p.Wo + oxx) * p.dstC + cbase : reinterpret_cast<f16 *>(trash);
                    } else {
                        // channels were permuted at pack time: ch = sub * dstC + c; the wave's 64 channels share one sub
                        const int Y = 2 * oy + (sub >> 1), X = 2 * oxx + (sub & 1);
                        const bool ok = oy < p.Ho && oxx < p.Wo && Y < p.Hd && X < p.Wd;
                        d = ok ? p.dst + ((size_t)Y * p.Wd + X) * cps + cbase : reinterpret_cast<f16 *>(trash);
                    }
                    *reinterpret_cast<f16x8 *>(d) = v;
                }
            }
        } else if constexpr (MODE == ST_POOL) {
            // 2x2 max: rows j, j+1 are in this lane; columns 2c, 2c+1 meet in the strip
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f16x4 m;
#pragma unroll
                    for (int r = 0; r < 4; ++r) m[r] = o[i][2 * jj][r] > o[i][2 * jj + 1][r] ? o[i][2 * jj][r] : o[i][2 * jj + 1][r];
                    *reinterpret_cast<f16x4 *>(stg + (jj * 16 + l15) * SP + (i * 16 + 4 * kg) * 2) = m;
                }
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                f16x8 v = *reinterpret_cast<const f16x8 *>(stg + (jj * 16 + 2 * s_row) * SP + s_chunk * 16);
                const f16x8 v1 = *reinterpret_cast<const f16x8 *>(stg + (jj * 16 + 2 * s_row + 1) * SP + s_chunk * 16);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = v[r] > v1[r] ? v[r] : v1[r];
                const int py = (cur.oy0 >> 1) + wp * 2 + jj, px = (cur.ox0 >> 1) + s_row;
                const bool ok = py < p.Hd && px < p.Wd;
                f16 *d = ok ? p.dst + ((size_t)py * p.Wd + px) * p.dstC + cur.n0 + wc * 64 + s_chunk * 8 : reinterpret_cast<f16 *>(trash);
                *reinterpret_cast<f16x8 *>(d) = v;
            }
        } else {   // ST_PS_DOT3: pixel shuffle, then 64 -> 3 dot products; only 3 partial sums per pixel leave the CU
            const float *s_w = reinterpret_cast<const float *>(smem + DOTW_OFF);
            float a0[4], a1[4], a2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a0[j] = a1[j] = a2[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 w0 = *reinterpret_cast<const float4 *>(s_w + i * 16 + 4 * kg);
                const float4 w1 = *reinterpret_cast<const float4 *>(s_w + 64 + i * 16 + 4 * kg);
                const float4 w2 = *reinterpret_cast<const float4 *>(s_w + 128 + i * 16 + 4 * kg);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x0 = (float)o[i][j][0], x1 = (float)o[i][j][1], x2 = (float)o[i][j][2], x3 = (float)o[i][j][3];
                    a0[j] += w0.x * x0 + w0.y * x1 + w0.z * x2 + w0.w * x3;
                    a1[j] += w1.x * x0 + w1.y * x1 + w1.z * x2 + w1.w * x3;
                    a2[j] += w2.x * x0 + w2.y * x1 + w2.z * x2 + w2.w * x3;
                }
            }
            // Sum the four k-groups of a pixel through the wave-private strip (ordinary LDS writes and reads), NOT with
            // ds_bpermute: with LDS-DMA of the workgroup in flight -- the next tile's weights are -- bpermute
            // sporadically returned wrong lanes' data on gfx950 (tests/test_gpu_parity.py:
            // test_persistent_schedules_do_not_change_results caught ~5000 of 8.3 M pixels per frame).  Lane
            // (kg, l15) then finishes row j = kg of the wave's four rows: one float4 store per lane, none wasted.
            const int sub = cur.n0 / 64 + wc;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4 *>(stg + j * 1024 + kg * 256 + l15 * 16) = make_float4(a0[j], a1[j], a2[j], 0.f);
            float4 r = *reinterpret_cast<const float4 *>(stg + kg * 1024 + l15 * 16);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float4 v = *reinterpret_cast<const float4 *>(stg + kg * 1024 + k * 256 + l15 * 16);
                r.x += v.x; r.y += v.y; r.z += v.z;
            }
            const int oy = cur.oy0 + wp * 4 + kg;
            const int Y = 2 * oy + (sub >> 1), X = 2 * ox + (sub & 1);
            const bool ok = oy < p.Ho && ox < p.Wo && Y < p.Hd && X < p.Wd;
            float4 *d = ok ? reinterpret_cast<float4 *>(p.dst_dot + ((size_t)Y * p.Wd + X) * 4) : reinterpret_cast<float4 *>(trash);
            *d = make_float4(r.x, r.y, r.z, 0.f);
        }
        cur = nxt;
    }
}

template <int MODE>
hipError_t launch_mode(const ConvParams &p, int grid, hipStream_t stream)
{
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv_pglds_kernel<MODE>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SMEM, stream, p);
    return hipGetLastError();
}

}  // namespace

// 3x3, stride 1, pad 1, Cin (src0 [+ src1 concat]) multiple of 64, Cout == CoutPad multiple of 128, no residuals;
// store modes NHWC / PS / POOL / PS_DOT3.  One block per CU (n_cu), each walking tiles.  hipErrorInvalidValue otherwise.
hipError_t conv_pglds_launch(ConvParams p, int n_cu, hipStream_t stream)
{
    if ((p.c0 % CT) || (p.c1 % CT) || p.c0 + p.c1 < CT || (p.CoutPad % BN) || p.Cout != p.CoutPad || p.res1 || p.res2 ||
        p.dst_full || !p.zeros || !p.trash || n_cu < 8 || (p.act != ACT_RELU && p.act != ACT_NONE) ||
        (p.mode != ST_NHWC && p.mode != ST_PS && p.mode != ST_POOL && p.mode != ST_PS_DOT3) ||
        (p.mode == ST_PS && (p.dstC % 64)) || (p.mode == ST_PS_DOT3 && (p.dstC != 64 || !p.dotw || !p.dst_dot)))
        return hipErrorInvalidValue;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const int total = p.tiles_x * p.tiles_y * (p.CoutPad / BN);
    const int grid = total < n_cu ? total : n_cu;
    switch (p.mode) {
    case ST_NHWC: return launch_mode<ST_NHWC>(p, grid, stream);
    case ST_PS: return launch_mode<ST_PS>(p, grid, stream);
    case ST_POOL: return launch_mode<ST_POOL>(p, grid, stream);
    default: return launch_mode<ST_PS_DOT3>(p, grid, stream);
    }
}
