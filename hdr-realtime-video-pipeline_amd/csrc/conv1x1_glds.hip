// conv1x1_glds.hip -- the HG head's 1x1 fuse convolutions conv6..conv9 over a channel concat
// (Hallucination_arch.py:116-134: cat(Up_convN, skip) -> 1x1): implicit GEMM with LDS-DMA staging.
//
// GEMM view: M = 128 output channels (A operand = packed weights), N = a 16x16-pixel tile, K = the
// 64-channel chunks of src0 then of src1 (the concat is never materialised).  These layers are
// HBM-bound (2 x C in, C/2 out per pixel), so the design goal is simply to keep enough bytes in
// flight: activations and weights of a chunk ride 3-slot LDS rings filled by global_load_lds two
// chunks ahead, one raw s_barrier per chunk behind a counted s_waitcnt vmcnt (never
// __syncthreads(): its fence would drain the DMA queue), XOR-swizzled 16-byte chunks on the SOURCE
// side (the LDS image of a DMA is lane-linear), epilogue through LDS with 16-byte NHWC stores.
#include <cstdlib>

#include "launchers.h"

namespace {

constexpr int TH = 16, TW = 16, NPIX = TH * TW;
constexpr int CT = 64, PIXB = CT * 2;                    // 64-channel chunk = 128 B per pixel
constexpr int A_PIECES_PER_WAVE = 4, A_BYTES = 8 * A_PIECES_PER_WAVE * 1024;   // 256 px -> 32 KiB per slot
constexpr int BN = 128;
constexpr int B_BYTES = BN * PIXB;                       // 16 KiB = 16 pieces = 2 per wave
constexpr int B_PIECES_PER_WAVE = 2;
constexpr int SMEM = 3 * A_BYTES + 3 * B_BYTES;          // 144 KiB
constexpr int OUT_ROWB = BN * 2 + 16;
static_assert(NPIX * OUT_ROWB <= SMEM, "epilogue tile must fit");

// 16-byte chunk swizzle: weight rows by row index, pixels by their column in the tile (with the column as key
// every ds_read_b128 lane group of a 32x32x16 fragment hits 16 distinct 16-byte slots of the 256-byte bank row)
__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }


#ifdef HDRTV_AB      // the one-tile-per-workgroup form: superseded by conv_glds1p below, kept in the A/B library as its yardstick
__global__ __launch_bounds__(512) void conv_glds1_kernel(ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;
    char *sB = smem + 3 * A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // bijective XCD-aware remap: each XCD (own L2) gets a contiguous run of tiles
    const int nwg = gridDim.x;
    int t;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int ntn = p.CoutPad / BN;
    const int nt_i = t % ntn, sp = t / ntn;
    const int ty = sp / p.tiles_x, tx = sp % p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = nt_i * BN;

    const int nchunk = (p.c0 + p.c1) / CT, nchunk0 = p.c0 / CT;

    // ---- LDS-DMA issue helpers (wave-uniform LDS base, per-lane swizzled source) ------------
    const int l_row = lane >> 3, l_slot = lane & 7;
    auto issue_A = [&](int cc, int buf) {
        const f16 *src;
        int cs, coff;
        if (cc < nchunk0) { src = p.src0; cs = p.s0_stride; coff = cc * CT; }
        else { src = p.src1; cs = p.s1_stride; coff = (cc - nchunk0) * CT; }
        const dma_rsrc_t ra = dma_rsrc(src, (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)cs * 2u);
#pragma unroll
        for (int it = 0; it < A_PIECES_PER_WAVE; ++it) {
            const int piece = wave + it * 8;
            const int hp = piece * 8 + l_row;
            const int hy = hp / TW, hx = hp - hy * TW;
            const int iy = oy0 + hy, ix = ox0 + hx;
            const bool ok = (iy < p.Hi) & (ix < p.Wi);
            const unsigned off = ((unsigned)(iy * p.Wi + ix) * (unsigned)cs + (unsigned)(coff + ((l_slot ^ swz64(hx)) << 3))) * 2u;
            dma16(ra, sA + buf * A_BYTES + piece * 1024, ok ? off : DMA_OOB);         // out of the image: zeros
        }
    };
    auto issue_B = [&](int cc, int slot) {
        const dma_rsrc_t rb = dma_rsrc(p.wpk, (unsigned)nchunk * (unsigned)p.CoutPad * (unsigned)(CT * 2));
        const unsigned so = (unsigned)(cc * p.CoutPad + n0) * (unsigned)(CT * 2);
#pragma unroll
        for (int k = 0; k < B_PIECES_PER_WAVE; ++k) {
            const int piece = wave * B_PIECES_PER_WAVE + k;
            const int n = piece * 8 + l_row;
            dma16(rb, sB + slot * B_BYTES + piece * 1024, (unsigned)(n * CT + ((l_slot ^ swz64(n)) << 3)) * 2u, so);
        }
    };

    // ---- wave tiling: 2 (channels) x 4 (pixels) waves, each 64 ch x 64 px = 2x2 MFMA tiles ---
    const int wc = wave & 1, wp = wave >> 1;
    int hp_base[2], hx_base[2], wrow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (wp * 2 + j) * 32 + l31;
        hp_base[j] = q;
        hx_base[j] = q % TW;
        wrow[j] = (wc * 2 + j) * 32 + l31;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    issue_A(0, 0);
    issue_B(0, 0);
    if (nchunk > 1) {
        issue_A(1, 1);
        issue_B(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    int slot = 0;
#pragma unroll 1
    for (int cc = 0; cc < nchunk; ++cc) {
        const bool pf = cc + 2 < nchunk;
        const int s2 = slot == 0 ? 2 : slot - 1;          // (slot + 2) % 3
        if (pf) { issue_A(cc + 2, s2); issue_B(cc + 2, s2); }
        const char *a = sA + slot * A_BYTES;
        const char *b = sB + slot * B_BYTES;
#pragma unroll
        for (int ks = 0; ks < CT / 16; ++ks) {
            const int chunk = ks * 2 + lh;
            f16x8 wf[2], xf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
                wf[i] = *reinterpret_cast<const f16x8 *>(b + wrow[i] * PIXB + ((chunk ^ swz64(wrow[i])) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j)
                xf[j] = *reinterpret_cast<const f16x8 *>(a + hp_base[j] * PIXB + ((chunk ^ swz64(hx_base[j])) << 4));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        if (pf) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // A(cc+2): 4 pieces, B(cc+2): 2 pieces
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot == 2 ? 0 : slot + 1;
    }

    // ---------------------------------------------------------------- epilogue (LDS staged)
    const float aslope = act_slope(p.act);
    char *so = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int cl = (wc * 2 + i) * 32 + 8 * qd + 4 * lh;
            const float4 sc = *reinterpret_cast<const float4 *>(p.scale + n0 + cl);
            const float4 sh = *reinterpret_cast<const float4 *>(p.shift + n0 + cl);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = (wp * 2 + j) * 32 + l31;
                f16x4 o;
                o[0] = (f16)act_fast(acc[i][j][4 * qd + 0] * sc.x + sh.x, aslope);
                o[1] = (f16)act_fast(acc[i][j][4 * qd + 1] * sc.y + sh.y, aslope);
                o[2] = (f16)act_fast(acc[i][j][4 * qd + 2] * sc.z + sh.z, aslope);
                o[3] = (f16)act_fast(acc[i][j][4 * qd + 3] * sc.w + sh.w, aslope);
                *reinterpret_cast<f16x4 *>(so + q * OUT_ROWB + cl * 2) = o;
            }
        }
    }
    __syncthreads();
    constexpr int CPP = BN / 8;
    for (int e = tid; e < NPIX * CPP; e += 512) {
        const int q = e / CPP, c8 = e % CPP;
        const int oy = oy0 + q / TW, ox = ox0 + q % TW;
        const int ch = n0 + c8 * 8;
        if (oy < p.Ho && ox < p.Wo && ch < p.Cout)
            *reinterpret_cast<f16x8 *>(p.dst + ((size_t)oy * p.Wo + ox) * p.dstC + ch) =
                *reinterpret_cast<const f16x8 *>(so + q * OUT_ROWB + c8 * 16);
    }
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Persistent form (round 2): one workgroup per CU walks a run of tiles and the (tile, chunk) iterations form ONE stream
// through the 3-slot rings, so that the next tile's first two chunks are in flight while the current tile finishes and a
// tile pays neither a launch nor a prologue (these layers are bandwidth-bound with K = 4..16 chunks: the prologue was a
// quarter of a conv9 tile).  Differences to the kernel above:
//   * the epilogue cannot take all of LDS (two slots hold the next tile's chunks): every wave transposes its 64 channels
//     x 64 pixels through a wave-private strip in the ring slot the tile's last chunk has just released, in two passes of
//     32 pixels (waves 0-5 in the A slot, 6-7 in the B slot); one extra barrier per tile lets the slot be refilled;
//   * vmcnt counts DMA pieces and stores together in issue order: every wave issues exactly NST stores per tile (masked
//     lanes store to the trash line), so the first wait of a tile is vmcnt(6 + NST) and every other one vmcnt(6).
constexpr int P_NST = 8;          // stores per wave and tile: 2 passes x 4
constexpr int P_SP = 144;         // strip row pitch: 64 ch x 2 B + 16
constexpr int P_MAXC = 512;       // most output channels the scale / shift table holds

__global__ __launch_bounds__(512) void conv_glds1p_kernel(ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;
    char *sB = smem + 3 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // ---- this block's run of tiles: XCD x owns a contiguous range, its blocks interleave in it (as conv_pglds)
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
    struct Tile { int n0, oy0, ox0; };
    auto decode = [&](int t) {
        Tile o;
        const int nt_i = t % ntn, sp = t / ntn;
        const int ty = sp / p.tiles_x, tx = sp - ty * p.tiles_x;
        o.n0 = nt_i * BN; o.oy0 = ty * TH; o.ox0 = tx * TW;
        return o;
    };
    const int nchunk = (p.c0 + p.c1) / CT, nchunk0 = p.c0 / CT;

    const int l_row = lane >> 3, l_slot = lane & 7;
    auto issue = [&](const Tile &T, int cc, int slot) {           // 4 A pieces + 2 B pieces per wave
        const f16 *src;
        int cs, coff;
        if (cc < nchunk0) { src = p.src0; cs = p.s0_stride; coff = cc * CT; }
        else { src = p.src1; cs = p.s1_stride; coff = (cc - nchunk0) * CT; }
        const dma_rsrc_t ra = dma_rsrc(src, (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)cs * 2u);
#pragma unroll
        for (int it = 0; it < A_PIECES_PER_WAVE; ++it) {
            const int piece = wave + it * 8;
            const int hp = piece * 8 + l_row;
            const int hy = hp / TW, hx = hp - hy * TW;
            const int iy = T.oy0 + hy, ix = T.ox0 + hx;
            const bool ok = (iy < p.Hi) & (ix < p.Wi);
            const unsigned off = ((unsigned)(iy * p.Wi + ix) * (unsigned)cs + (unsigned)(coff + ((l_slot ^ swz64(hx)) << 3))) * 2u;
            dma16(ra, sA + slot * A_BYTES + piece * 1024, ok ? off : DMA_OOB);      // out of the image: zeros
        }
        const dma_rsrc_t rb = dma_rsrc(p.wpk, (unsigned)nchunk * (unsigned)p.CoutPad * (unsigned)(CT * 2));
        const unsigned so = (unsigned)(cc * p.CoutPad + T.n0) * (unsigned)(CT * 2);
#pragma unroll
        for (int k = 0; k < B_PIECES_PER_WAVE; ++k) {
            const int piece = wave * B_PIECES_PER_WAVE + k;
            const int n = piece * 8 + l_row;
            dma16(rb, sB + slot * B_BYTES + piece * 1024, (unsigned)(n * CT + ((l_slot ^ swz64(n)) << 3)) * 2u, so);
        }
    };

    // ---- wave tiling: 2 (channels) x 4 (pixels) waves, each 64 ch x 64 px = 2x2 MFMA tiles
    const int wc = wave & 1, wp = wave >> 1;
    int hp_base[2], hx_base[2], wrow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (wp * 2 + j) * 32 + l31;
        hp_base[j] = q;
        hx_base[j] = q % TW;
        wrow[j] = (wc * 2 + j) * 32 + l31;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    const float act_lb = p.act == ACT_RELU ? 0.f : -__builtin_inff();       // ReLU or none (launcher)
    char *trash = reinterpret_cast<char *>(p.trash) + lane * 16;
    // scale / shift of every output channel in LDS, once per workgroup and before any DMA is in flight (a global load in
    // the epilogue is waited for with vmcnt(0) by hipcc, one round trip per use)
    __shared__ __attribute__((aligned(16))) float s_ss[2 * P_MAXC];
    for (int e = tid; e < p.CoutPad; e += 512) { s_ss[e] = p.scale[e]; s_ss[P_MAXC + e] = p.shift[e]; }
    __syncthreads();

    Tile cur = decode(t_first), nxt = cur;
    issue(cur, 0, 0);
    issue(cur, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int slot = 0;
    for (int k = 0; k < ntile; ++k) {
        const bool has_next = k + 1 < ntile;
        if (has_next) nxt = decode(t_first + (k + 1) * t_step);
#pragma unroll 1
        for (int cc = 0; cc < nchunk; ++cc) {
            const int s2 = slot == 0 ? 2 : slot - 1;          // (slot + 2) % 3
            bool pf = true;
            if (cc + 2 < nchunk) issue(cur, cc + 2, s2);
            else if (has_next) issue(nxt, cc + 2 - nchunk, s2);
            else pf = false;
            const char *a = sA + slot * A_BYTES;
            const char *b = sB + slot * B_BYTES;
#pragma unroll
            for (int ks = 0; ks < CT / 16; ++ks) {
                const int chunk = ks * 2 + lh;
                f16x8 wf[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    wf[i] = *reinterpret_cast<const f16x8 *>(b + wrow[i] * PIXB + ((chunk ^ swz64(wrow[i])) << 4));
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    xf[j] = *reinterpret_cast<const f16x8 *>(a + hp_base[j] * PIXB + ((chunk ^ swz64(hx_base[j])) << 4));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
            // the next iteration reads the chunk issued one iteration ago: allow exactly what is younger than it
            if (!pf) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (cc == 0 && k > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 + P_NST) : "memory");   // + the last tile's stores
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (cc + 1 < nchunk) slot = slot == 2 ? 0 : slot + 1;
        }
        // ---- epilogue from the accumulators through a wave-private strip in the slot the last chunk has released
        char *stg = wave < 6 ? sA + slot * A_BYTES + wave * (32 * P_SP) : sB + slot * B_BYTES + (wave - 6) * (32 * P_SP);
        const int s_row = lane >> 3, s_chunk = lane & 7;
        const int chb = cur.n0 + wc * 64 + s_chunk * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int cl = (wc * 2 + i) * 32 + 8 * qd + 4 * lh;
                    const float4 sc = *reinterpret_cast<const float4 *>(s_ss + cur.n0 + cl);
                    const float4 sh = *reinterpret_cast<const float4 *>(s_ss + P_MAXC + cur.n0 + cl);
                    f16x4 o;
                    o[0] = (f16)fmaxf(acc[i][j][4 * qd + 0] * sc.x + sh.x, act_lb);
                    o[1] = (f16)fmaxf(acc[i][j][4 * qd + 1] * sc.y + sh.y, act_lb);
                    o[2] = (f16)fmaxf(acc[i][j][4 * qd + 2] * sc.z + sh.z, act_lb);
                    o[3] = (f16)fmaxf(acc[i][j][4 * qd + 3] * sc.w + sh.w, act_lb);
                    *reinterpret_cast<f16x4 *>(stg + l31 * P_SP + (i * 32 + 8 * qd + 4 * lh) * 2) = o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][4 * qd + r] = 0.f;
                }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int px = rr * 8 + s_row;                                   // pixel of this pass: row px >> 4 of two
                const f16x8 v = *reinterpret_cast<const f16x8 *>(stg + px * P_SP + s_chunk * 16);
                const int oy = cur.oy0 + (wp * 2 + j) * 2 + (px >> 4), ox = cur.ox0 + (px & 15);
                const bool ok = oy < p.Ho && ox < p.Wo && chb < p.Cout;
                f16 *d = ok ? p.dst + ((size_t)oy * p.Wo + ox) * p.dstC + chb : reinterpret_cast<f16 *>(trash);
                *reinterpret_cast<f16x8 *>(d) = v;
            }
        }
        // every wave is done with its strip before the next iteration's DMA refills the slot
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        slot = slot == 2 ? 0 : slot + 1;
        cur = nxt;
    }
}

}  // namespace

// 1x1, stride 1, Cin (src0 [+ src1 concat]) multiple of 64, CoutPad multiple of 128, NHWC store, no residuals.
hipError_t conv_glds1_launch(ConvParams p, hipStream_t stream, int n_cu, bool old_form)
{
    if ((p.c0 % CT) || (p.c1 % CT) || p.c0 + p.c1 < CT || (p.CoutPad % BN) || p.res1 || p.res2 || p.dst_full ||
        p.mode != ST_NHWC || !p.zeros)
        return hipErrorInvalidValue;
#ifdef HDRTV_AB
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_glds1_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
#endif
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const int grid = p.tiles_x * p.tiles_y * (p.CoutPad / BN);
    // persistent form: needs two chunks for its pipeline, the trash line, ReLU or no activation (old_form: the caller's A/B switch)
    const int nchunk = (p.c0 + p.c1) / CT;
    if (!old_form && n_cu >= 8 && nchunk >= 2 && p.trash && p.CoutPad <= P_MAXC && (p.act == ACT_RELU || p.act == ACT_NONE)) {
        static DevOnce attr_p;
        if (attr_p.need()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_glds1p_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
            if (e != hipSuccess) return e;
            attr_p.done();
        }
        hipLaunchKernelGGL(conv_glds1p_kernel, dim3(grid < n_cu ? grid : n_cu), dim3(512), SMEM, stream, p);
        return hipGetLastError();
    }
#ifdef HDRTV_AB
    hipLaunchKernelGGL(conv_glds1_kernel, dim3(grid), dim3(512), SMEM, stream, p);
    return hipGetLastError();
#else
    return hipErrorNotSupported;       // the one-tile-per-workgroup kernel exists in the A/B library only (make AB=1)
#endif
}
