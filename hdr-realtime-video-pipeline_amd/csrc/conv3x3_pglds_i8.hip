// conv3x3_pglds_i8.hip -- the persistent HG 3x3 convolution of conv3x3_pglds.hip on int8 MFMA (W8A8 layers of the HG head,
// BASELINE.json configs[4]; semantics: W8A8Conv2d.forward, hdrtvnet_torch.py:351-364, asymmetric u8 activations).
//
// Activations live in HBM as int8 codes c = q - 128 (q = the reference's u8 code), NHWC; weights as the checkpoint's
// int8.  With an integer zero point k (x_zero = -k * x_scale) the reference's zero padding after dequantisation is the
// code k, so out-of-image halo pixels are filled from a constant line and
//     conv = x_scale * w_scale[co] * (sum c * w + (128 - k) * sum w)
// exactly, in integers; the second term, the bias, BatchNorm and the OUTPUT tensor's quantiser are folded on the host
// into one per-channel {scale, shift}, so the epilogue is  code = clamp(rint(acc * scale + shift), -128, 127)
// (ReLU is the lower clamp: post-ReLU tensors have k = 0).  The last quantised layer, Up_conv5, has an fp16 reader: its
// {scale, shift} give real values and its epilogue is the f16 kernel's ST_PS_DOT3 (pixel shuffle, ReLU, 64 -> 3 dot products).
//
// Byte geometry is the f16 kernel's: a 128-channel int8 chunk is the same 128 B per pixel as its 64-channel f16 chunk,
// so tile sizes, LDS images, swizzles, LDS-DMA pieces, the weight ring and every counted wait are unchanged; one
// v_mfma_i32_16x16x64_i8 consumes the 64 bytes of K that two f16 MFMAs did, at the same cycles.
#include "launchers.h"

namespace {

constexpr int TH = 16, TW = 16, HW = 18, NPIX = HW * HW;
constexpr int CT = 128, PIXB = CT;                       // 128-channel int8 chunk = 128 B per pixel
constexpr int BN = 128;
constexpr int A_PIECES_PER_WAVE = 6, A_BYTES = 8 * A_PIECES_PER_WAVE * 1024;   // 324 halo px -> 48 KiB
constexpr int B_BYTES = BN * PIXB, B_PIECES_PER_WAVE = 2;                       // 16 KiB
constexpr int SS_OFF = 2 * A_BYTES + 3 * B_BYTES;        // two 1-KiB {scale[128], shift[128]} slots
constexpr int DOTW_OFF = SS_OFF + 2048;                  // ST_PS_DOT3: 3 x 64 floats
constexpr int SMEM = DOTW_OFF + 1024;                    // 147 KiB

// stores per wave and tile, by store mode (see the epilogues)
template <int MODE> struct NStores { static constexpr int N = (MODE == ST_POOL || MODE == ST_PS_DOT3) ? 1 : 4; };


template <int N> __device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

struct Tile { int n0, oy0, ox0; };


// C64: a 64-channel input (conv2, Up_conv5).  One LDS row is then a PAIR of horizontally adjacent pixels (pixel hx in
// bytes 0..63, pixel hx+1 in bytes 64..127; each halo pixel is DMA'd twice), and the 3x3 filter becomes 3 rows x 2
// row-taps at columns {0, 2}: tap (ky, 0) multiplies [w(ky,0) | w(ky,1)], tap (ky, 2) multiplies [w(ky,2) | 0].  Everything
// else -- tile, ring, waits, epilogues -- is the 128-channel kernel with 6 taps and one chunk; a quarter of the MFMA
// work multiplies zeros, which is still 1.5x the fp16 kernel's rate on these layers.
template <int MODE, bool C64>
__global__ __launch_bounds__(512) void conv_pglds_i8_kernel(ConvI8Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;
    char *sB = smem + 2 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;

    // ---- this block's run of tiles: XCD x owns a contiguous range, its blocks interleave in it --
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
        const int nt_i = t % ntn, sp = t / ntn;
        const int ty = sp / p.tiles_x, tx = sp - ty * p.tiles_x;
        o.n0 = nt_i * BN; o.oy0 = ty * TH; o.ox0 = tx * TW;
        return o;
    };

    constexpr int NT = C64 ? 6 : 9;                    // taps per chunk
    const int nchunk = C64 ? 1 : (p.c0 + p.c1) / CT, nchunk0 = C64 ? 1 : p.c0 / CT;
    const int nit = nchunk * NT;

    // ---- LDS-DMA issue helpers (wave-uniform LDS base, per-lane swizzled source) ------------
    const int l_row = lane >> 3, l_slot = lane & 7;
    auto issue_A = [&](int cc, int buf, const Tile &T) {
        const int8_t *src;
        int cs, coff;
        if (cc < nchunk0) { src = p.src0; cs = p.c0; coff = cc * CT; }
        else { src = p.src1; cs = p.c1; coff = (cc - nchunk0) * CT; }
        // LDS-DMA as a buffer load (common.h): lanes outside the image write zeros = code 0, whose share of the sum the
        // epilogue's border-class constants take back out
        const dma_rsrc_t ra = dma_rsrc(src, (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)(C64 ? 64 : cs));
#pragma unroll
        for (int it = 0; it < A_PIECES_PER_WAVE; ++it) {
            const int piece = wave + it * 8;
            const int hp = piece * 8 + l_row;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int iy = T.oy0 - 1 + hy, ix = T.ox0 - 1 + hx;
            unsigned off;
            bool ok;
            if constexpr (C64) {
                const int sc = l_slot ^ (hx & 7);            // source chunk 0..3: this pixel, 4..7: its right neighbour
                const int ixx = ix + (sc >> 2);
                ok = (hp < NPIX) & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ixx < (unsigned)p.Wi);
                off = (unsigned)(iy * p.Wi + ixx) * 64u + (unsigned)((sc & 3) << 4);
            } else {
                ok = (hp < NPIX) & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
                off = (unsigned)(iy * p.Wi + ix) * (unsigned)cs + (unsigned)(coff + ((l_slot ^ (hx & 7)) << 4));
            }
            dma16(ra, sA + buf * A_BYTES + piece * 1024, ok ? off : DMA_OOB);
        }
    };
    auto issue_B = [&](int it_i, int n0, int slot) {
        const int cc = it_i / NT, tap = it_i - cc * NT;
        const dma_rsrc_t rb = dma_rsrc(p.wpk, (unsigned)NT * (unsigned)nchunk * (unsigned)p.Cout * (unsigned)CT);
        const unsigned so = (unsigned)((tap * nchunk + cc) * p.Cout + n0) * (unsigned)CT;
#pragma unroll
        for (int k = 0; k < B_PIECES_PER_WAVE; ++k) {
            const int piece = wave * B_PIECES_PER_WAVE + k;
            const int n = piece * 8 + l_row;
            dma16(rb, sB + slot * B_BYTES + piece * 1024, (unsigned)(n * CT + ((l_slot ^ (n & 7)) << 4)), so);
        }
    };
    auto issue_SS = [&](int n0, int slot) {      // every wave writes the same 1 KiB: {scale[128], shift[128]}
        const char *sc = reinterpret_cast<const char *>(p.scale), *sh = reinterpret_cast<const char *>(p.shift);
        const char *lo = sc < sh ? sc : sh;                      // one (wave-uniform) resource over both arrays
        dma16(dma_rsrc(lo, 0xffffffffu), smem + SS_OFF + slot * 1024, (unsigned)((lane < 32 ? sc : sh) - lo) + (unsigned)(lane & 31) * 16u,
              (unsigned)n0 * 4u);
    };

    if constexpr (MODE == ST_PS_DOT3) {          // before any DMA is in flight (ordinary loads drain the queue)
        // The 64 -> 3 dot products run on the matrix pipe (epilogue; scheme in conv3x3_prw.hip).  Table entry
        // ((khalf * 3 + o) * 4 + kg): the A fragment of output o for k-group kg over channels 32 khalf .. 32 khalf + 31 -- K slot e of
        // the group is channel 32 khalf + 4 kg + e (e < 4) or 32 khalf + 16 + 4 kg + e - 4, the order in which a lane's f16x4 results
        // of two accumulator blocks form a B fragment as they lie.  An fp32 weight rides as hi + lo * 2^-10 (two f16 operands).
        if (tid < 24) {
            const int kgq = tid & 3, o = (tid >> 2) % 3, khalf = tid / 12;
            f16x8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float w = p.dotw[o * 64 + khalf * 32 + (e < 4 ? 4 * kgq + e : 16 + 4 * kgq + e - 4)];
                hi[e] = (f16)w;
                lo[e] = (f16)((w - (float)hi[e]) * 1024.f);
            }
            *reinterpret_cast<f16x8 *>(smem + DOTW_OFF + tid * 32) = hi;
            *reinterpret_cast<f16x8 *>(smem + DOTW_OFF + tid * 32 + 16) = lo;
        }
        __syncthreads();
    }

    // ---- wave tiling: 2 (channels) x 4 (pixel rows) waves, each 64 ch x 64 px = 4x4 tiles of 16x16
    const int wc = wave & 1, wp = wave >> 1;
    i32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
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
            for (int tap = 0; tap < NT; ++tap) {
                const int it_i = cc * NT + tap;
                // stream index = NT * gch + tap (NT = 9 or 6), so the ring slot of this iteration is tap % 3
                // (a tile's weights(2) are issued before the tile starts: prologue / end of the previous tile)
                bool pfB = true;
                if (tap == 0 && cc == 0) {}
                else if (it_i + 2 < nit) issue_B(it_i + 2, cur.n0, (tap + 2) % 3);
                else if (has_next) issue_B(it_i + 2 - nit, nxt.n0, (tap + 2) % 3);
                else pfB = false;
                if (tap == NT - 3 && pfA) {
                    if (!last_chunk) issue_A(cc + 1, (gch + 1) & 1, cur);
                    else { issue_A(0, (gch + 1) & 1, nxt); issue_SS(nxt.n0, (k + 1) & 1); }
                }

                const char *bw = sB + (tap % 3) * B_BYTES + b_lane;
                const int t_ky = C64 ? tap / 2 : tap / 3, t_kx = C64 ? 2 * (tap % 2) : tap % 3;
                const char *ax = a + a_lane + (t_ky * HW + t_kx) * PIXB;
                const int kx = (l15 + t_kx) & 7;
                i32x4 wf[2][4], xf[2][4];
                auto ldw = [&](int ks, int i) {
                    wf[ks][i] = *reinterpret_cast<const i32x4 *>(bw + i * 16 * PIXB + (((ks * 4 + kg) ^ kw) << 4));
                };
                auto ldx = [&](int ks, int j) {
                    xf[ks][j] = *reinterpret_cast<const i32x4 *>(ax + j * HW * PIXB + (((ks * 4 + kg) ^ kx) << 4));
                };
                // program order IS the schedule: sched_barrier(0) lets nothing cross
#pragma unroll
                for (int i = 0; i < 4; ++i) { ldw(0, i); ldx(0, i); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    acc[g >> 2][g & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[0][g >> 2], xf[0][g & 3], acc[g >> 2][g & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // k-step 1 fragments in the order its MFMAs want them: w0 x0 x1 x2 x3 w1 w2 w3
                    if (g == 0) ldw(1, 0); else if (g < 5) ldx(1, g - 1); else ldw(1, g - 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 8; m < 16; ++m)
                    acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[0][m >> 2], xf[0][m & 3], acc[m >> 2][m & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 16; ++m)
                    acc[m >> 2][m & 3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[1][m >> 2], xf[1][m & 3], acc[m >> 2][m & 3], 0, 0, 0);

                // The next iteration reads weights(s+1), issued one iteration ago, and at a chunk boundary
                // the halo staged at tap 6.  Allow exactly the DMAs (and, right after a tile boundary, the
                // previous tile's stores) that are younger than those.
                if (!pfB) {
                    wait_vm<0>();
                } else if ((tap == NT - 3 || tap == NT - 2) && pfA) {
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
        // wave-private strip of the halo buffer this tile just finished with (free until the next tile's tap 6); LDS
        // operations of one wave execute in order: no barrier
        char *stg = sA + ((gch - 1) & 1) * A_BYTES + wave * 5120;
        // float zero point (p.delta): pixels whose 3x3 window leaves the image get their border class's constant added to the
        // shift.  Only tiles on the image border take the branch (wave-uniform test first); the class is recomputed per use
        // rather than kept, this kernel has no registers to spare
        const bool edge_tile = p.delta && (cur.oy0 == 0 || cur.ox0 == 0 || cur.oy0 + TH >= p.Hi || cur.ox0 + TW >= p.Wi);
        auto shift_of = [&](const float4 &sh, int i, int j) -> float4 {
            if (!edge_tile) return sh;
            const int oy_ = cur.oy0 + wp * 4 + j, ox_ = cur.ox0 + l15;
            const int cls = ((((oy_ == 0) | ((oy_ == p.Hi - 1) << 1)) << 2) | ((ox_ == 0) | ((ox_ == p.Wi - 1) << 1))) & 15;
            if (!cls) return sh;
            const float4 d = *reinterpret_cast<const float4 *>(p.delta + (size_t)cls * p.Cout + cur.n0 + cw + i * 16);
            return make_float4(sh.x + d.x, sh.y + d.y, sh.z + d.z, sh.w + d.w);
        };
        if constexpr (MODE == ST_PS_DOT3) {
            // float zero point: this epilogue has no register to spare for per-use shift corrections, and its output is rounded to
            // f16 anyway, so the border constant is applied to the integer sums, rounded to accumulator units (error <= half a
            // unit = x_scale * w_scale / 2, three orders of magnitude below the f16 rounding that follows)
            if (edge_tile) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int oy_ = cur.oy0 + wp * 4 + j, ox_ = cur.ox0 + l15;
                    const int cls = ((((oy_ == 0) | ((oy_ == p.Hi - 1) << 1)) << 2) | ((ox_ == 0) | ((ox_ == p.Wi - 1) << 1))) & 15;
                    if (cls) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[i][j] += *reinterpret_cast<const i32x4 *>(p.delta_acc + (size_t)cls * p.Cout + cur.n0 + cw + i * 16);
                    }
                }
            }
            // Up_conv5: {scale, shift} give real values; ReLU, f16 rounding (the tensor the reference's fp16 conv10 reads),
            // pixel shuffle, then the first half of conv10 as 3 dot products per pixel (conv3x3_pglds.hip, same epilogue)
            // f16 results of pixel row j as B fragments (blocks 0|1 and 2|3), A = the three weight rows placed in rows 4 j + o: the
            // four rows accumulate into one f32x4 whose lane (kg, l15) is pixel (wp * 4 + kg, l15)
            const int orow = l15 & 3, qrow = l15 >> 2;
            const f16x8 zero8 = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            f16x8 fh[2], fl[2];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const char *tb = smem + DOTW_OFF + (((kh * 3 + (orow < 3 ? orow : 0)) * 4 + kg) << 5);
                fh[kh] = *reinterpret_cast<const f16x8 *>(tb);
                fl[kh] = *reinterpret_cast<const f16x8 *>(tb + 16);
                if (orow == 3) { fh[kh] = zero8; fl[kh] = zero8; }
            }
            float4 scv[4], shv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                scv[i] = *reinterpret_cast<const float4 *>(ss + cw + i * 16);
                shv[i] = *reinterpret_cast<const float4 *>(ss + 128 + cw + i * 16);
            }
            f32x4 ah = {0.f, 0.f, 0.f, 0.f}, al = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f16x4 x[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    x[i][0] = (f16)fmaxf((float)acc[i][j][0] * scv[i].x + shv[i].x, 0.f);
                    x[i][1] = (f16)fmaxf((float)acc[i][j][1] * scv[i].y + shv[i].y, 0.f);
                    x[i][2] = (f16)fmaxf((float)acc[i][j][2] * scv[i].z + shv[i].z, 0.f);
                    x[i][3] = (f16)fmaxf((float)acc[i][j][3] * scv[i].w + shv[i].w, 0.f);
                    acc[i][j] = i32x4{0, 0, 0, 0};
                }
                const bool mine = qrow == j;
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const f16x8 b = {x[2 * kh][0], x[2 * kh][1], x[2 * kh][2], x[2 * kh][3], x[2 * kh + 1][0], x[2 * kh + 1][1], x[2 * kh + 1][2], x[2 * kh + 1][3]};
                    ah = __builtin_amdgcn_mfma_f32_16x16x32_f16(mine ? fh[kh] : zero8, b, ah, 0, 0, 0);
                    al = __builtin_amdgcn_mfma_f32_16x16x32_f16(mine ? fl[kh] : zero8, b, al, 0, 0, 0);
                }
            }
            const int sub = cur.n0 / 64 + wc;
            constexpr float kscale = 1.f / 1024.f;
            const float4 r = make_float4(ah[0] + al[0] * kscale, ah[1] + al[1] * kscale, ah[2] + al[2] * kscale, 0.f);
            const int oy = cur.oy0 + wp * 4 + kg, ox = cur.ox0 + l15;
            const int Y = 2 * oy + (sub >> 1), X = 2 * ox + (sub & 1);
            const bool ok = oy < p.Ho && ox < p.Wo && Y < p.Hd && X < p.Wd;
            float4 *d = ok ? reinterpret_cast<float4 *>(p.dst_dot + ((size_t)Y * p.Wd + X) * 4) : reinterpret_cast<float4 *>(trash);
            *d = make_float4(r.x, r.y, r.z, 0.f);
        } else {
            // int8 codes of the output tensor's quantiser: clamp(rint(acc * scale + shift), lo, 127), computed as the u8 code
            // 128 higher: q holds max(acc * scale + shift + 128, lo + 128), still unrounded (the 2x2 max of ST_POOL commutes with
            // the monotone quantiser), and v_cvt_pk_u8_f32 rounds to nearest even, saturates to [0, 255] and packs in one
            // instruction (common.h quant4; tools/cvt_pk_u8_probe.hip); xor 0x80 per byte brings the code back to int8.
            // 4 VALU operations per value where rint / min / max / cvt / mask / shift / or took 8.5.
            float q[4][4][4];
            const float lo128 = p.lo_clamp + 128.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 sc = *reinterpret_cast<const float4 *>(ss + cw + i * 16);
                float4 sh = *reinterpret_cast<const float4 *>(ss + 128 + cw + i * 16);
                sh.x += 128.f; sh.y += 128.f; sh.z += 128.f; sh.w += 128.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 sj = shift_of(sh, i, j);
                    q[i][j][0] = fmaxf((float)acc[i][j][0] * sc.x + sj.x, lo128);
                    q[i][j][1] = fmaxf((float)acc[i][j][1] * sc.y + sj.y, lo128);
                    q[i][j][2] = fmaxf((float)acc[i][j][2] * sc.z + sj.z, lo128);
                    q[i][j][3] = fmaxf((float)acc[i][j][3] * sc.w + sj.w, lo128);
                    acc[i][j] = i32x4{0, 0, 0, 0};
                }
            }
            auto pack4 = [](const float *v) -> unsigned {
                unsigned w = 0;
                w = __builtin_amdgcn_cvt_pk_u8_f32(v[0], 0, w);
                w = __builtin_amdgcn_cvt_pk_u8_f32(v[1], 1, w);
                w = __builtin_amdgcn_cvt_pk_u8_f32(v[2], 2, w);
                w = __builtin_amdgcn_cvt_pk_u8_f32(v[3], 3, w);
                return w ^ 0x80808080u;
            };
            constexpr int SP = 80;                                   // strip row pitch: 64 ch x 1 B + 16
            const int s_px = lane >> 2, s_chunk = lane & 3;
            int8_t *dst = reinterpret_cast<int8_t *>(p.dst);
            if constexpr (MODE == ST_NHWC || MODE == ST_PS) {
                const int cps = p.dstC;
                const int chw = cur.n0 + wc * 64;
                const int sub = MODE == ST_PS ? chw / cps : 0;
                const int cbase = (MODE == ST_PS ? chw - sub * cps : chw) + s_chunk * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<unsigned *>(stg + (j * 16 + l15) * SP + i * 16 + 4 * kg) = pack4(q[i][j]);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {                     // strip row rr*16 + s_px = wave pixel row rr, column s_px
                    const i32x4 v = *reinterpret_cast<const i32x4 *>(stg + (rr * 16 + s_px) * SP + s_chunk * 16);
                    const int oy = cur.oy0 + wp * 4 + rr;
                    const int oxx = cur.ox0 + s_px;
                    int8_t *d;
                    if constexpr (MODE == ST_NHWC) {
                        const bool ok = oy < p.Ho && oxx < p.Wo;
                        d = ok ? dst + ((size_t)oy * p.Wo + oxx) * p.dstC + cbase : reinterpret_cast<int8_t *>(trash);
                    } else {
                        // channels were permuted at pack time: ch = sub * dstC + c; the wave's 64 channels share one sub
                        const int Y = 2 * oy + (sub >> 1), X = 2 * oxx + (sub & 1);
                        const bool ok = oy < p.Ho && oxx < p.Wo && Y < p.Hd && X < p.Wd;
                        d = ok ? dst + ((size_t)Y * p.Wd + X) * cps + cbase : reinterpret_cast<int8_t *>(trash);
                    }
                    *reinterpret_cast<i32x4 *>(d) = v;
                }
            } else {   // ST_POOL: 2x2 max of the codes (the quantiser is monotone): rows in-lane, columns by DPP (no LDS)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float m[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float a = fmaxf(q[i][2 * jj][r], q[i][2 * jj + 1][r]);
                            const float b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, false));
                            m[r] = fmaxf(a, b);
                        }
                        if ((l15 & 1) == 0)
                            *reinterpret_cast<unsigned *>(stg + (jj * 8 + (l15 >> 1)) * SP + i * 16 + 4 * kg) = pack4(m);
                    }
                const i32x4 v = *reinterpret_cast<const i32x4 *>(stg + s_px * SP + s_chunk * 16);
                const int py = (cur.oy0 >> 1) + wp * 2 + (s_px >> 3), px = (cur.ox0 >> 1) + (s_px & 7);
                const bool ok = py < p.Hd && px < p.Wd;
                int8_t *d = ok ? dst + ((size_t)py * p.Wd + px) * p.dstC + cur.n0 + wc * 64 + s_chunk * 16 : reinterpret_cast<int8_t *>(trash);
                *reinterpret_cast<i32x4 *>(d) = v;
            }
        }
        cur = nxt;
    }
}

template <int MODE, bool C64>
hipError_t launch_mode(const ConvI8Params &p, int grid, hipStream_t stream)
{
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv_pglds_i8_kernel<MODE, C64>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SMEM, stream, p);
    return hipGetLastError();
}

}  // namespace

// 3x3, stride 1, pad 1 on int8 codes: Cin (src0 [+ src1 concat]) multiple of 128, or exactly 64 (pixel-pair rows; NHWC
// store, or PS_DOT3 = pixel shuffle + ReLU + fused 64->3 dot products; weights packed as 6 row-taps); Cout multiple of 128; store modes
// NHWC / PS / POOL to int8 codes.  One block per CU, each walking tiles.
hipError_t conv_pglds_i8_launch(ConvI8Params p, int n_cu, hipStream_t stream)
{
    const bool c64 = p.c0 == 64 && p.c1 == 0;
    if ((!c64 && ((p.c0 % CT) || (p.c1 % CT) || p.c0 + p.c1 < CT)) || (p.Cout % BN) || !p.padline || !p.trash || n_cu < 8 ||
        (p.mode != ST_NHWC && p.mode != ST_PS && p.mode != ST_POOL && p.mode != ST_PS_DOT3) ||
        (p.out_f16 != (p.mode == ST_PS_DOT3)) || (p.mode == ST_PS && (p.dstC % 64)) ||
        (p.mode == ST_PS_DOT3 && (!c64 || p.dstC != 64 || !p.dotw || !p.dst_dot)))
        return hipErrorInvalidValue;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const int total = p.tiles_x * p.tiles_y * (p.Cout / BN);
    const int grid = total < n_cu ? total : n_cu;
    if (c64)
        return p.mode == ST_NHWC ? launch_mode<ST_NHWC, true>(p, grid, stream)
                                 : (p.mode == ST_PS_DOT3 ? launch_mode<ST_PS_DOT3, true>(p, grid, stream) : hipErrorInvalidValue);
    switch (p.mode) {
    case ST_NHWC: return launch_mode<ST_NHWC, false>(p, grid, stream);
    case ST_PS: return launch_mode<ST_PS, false>(p, grid, stream);
    default: return launch_mode<ST_POOL, false>(p, grid, stream);
    }
}
