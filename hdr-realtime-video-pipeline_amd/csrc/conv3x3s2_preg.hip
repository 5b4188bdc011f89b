// conv3x3s2_preg.hip -- 3x3 / stride-2 convolution from 64 channels (gfx950): the first layers of
// LE's CondNet2/3/4 (HDRUNet3T1_arch.py:47-55), merged into ONE launch with 192 output channels
// that reads the 64-channel full-resolution condition map once, and the 64->64 second layers.
//
// Persistent, weights in registers.  Cin = 64 makes the whole filter bank small (9 x 64 x Cout f16:
// 216 KiB at Cout = 192), so instead of streaming it through LDS for every tile, wave w keeps the
// complete K = 576 filter rows of output channels 16w..16w+15 in 72 VGPRs for the life of the
// block (12 waves at Cout = 192, 4 at Cout = 64) and the block walks output tiles of 8 x 16 pixels:
//   * the (17 x 33 pixel) x 64-channel input halo of a tile is staged by LDS-DMA into one of two
//     72-KiB buffers, two tiles ahead of its use; there is no per-tap weight traffic and no per-tap
//     barrier, a tile is 144 MFMAs (16x16x32) per wave against 144 ds_read_b128;
//   * stride-2 taps read every second halo column.  At 128 bytes per pixel that keeps all 16 lanes
//     of a ds_read_b128 group in one half of the 256-byte bank row (2-way conflict whatever the
//     swizzle), so the halo is staged column-de-interleaved -- even columns first, then the odd
//     ones; the per-lane DMA source address makes that free -- and a tap's 16 pixels are again 16
//     consecutive LDS rows, conflict-free under the (row & 7) chunk swizzle;
//   * epilogue through the halo buffer the tile just released: bias + LeakyReLU, padded pixel
//     rows, 16-byte stores of whole NHWC pixels.
#include "launchers.h"

namespace {

constexpr int TH = 8, TW = 16;
constexpr int HH = 2 * TH + 1, HWD = 2 * TW + 1, NPIX = HH * HWD;   // 17 x 33 = 561 halo pixels
constexpr int NEVEN = TW + 1;                                       // even halo columns 0,2,..,32 come first
constexpr int PIXB = 128;
constexpr int A_PIECES = (NPIX + 7) / 8;                            // 71 one-KiB pieces (8 pixels each)
constexpr int A_BYTES = 72 * 1024;
constexpr int SMEM = 2 * A_BYTES;
constexpr int TAIL_OFF = 2 * A_BYTES, TAIL_WB = 12 * 1024, TAIL_SMEM = TAIL_WB + 96 * 4;   // the fused CondNet2 tail's fragments + bias

// (cond_tail_kernel's helpers, le_fused.hip: the fused tail must reproduce its arithmetic bit for bit)
__device__ __forceinline__ f32x16 tail_bias_tile(const float *b, int lh)
{
    f32x16 a;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4 *>(b + 8 * g + 4 * lh);
        a[4 * g + 0] = v.x; a[4 * g + 1] = v.y; a[4 * g + 2] = v.z; a[4 * g + 3] = v.w;
    }
    return a;
}
__device__ __forceinline__ f16x8 tail_lrelu_pack(const f32x16 &a, int s)
{
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (f16)a[8 * s + j];
    return __builtin_elementwise_max(o, o * (f16)0.1f);
}


template <int NW>   // waves = 16-channel tiles: Cout = 16 * NW
__global__ __launch_bounds__(64 * NW, 1) void conv3x3s2_preg_kernel(ConvParams p)   // 1 block per CU: LDS-limited anyway
{
    constexpr int NT = 64 * NW, COUT = 16 * NW;
    constexpr int OUT_ROWB = COUT * 2 + 16;
    static_assert(TH * TW * OUT_ROWB <= A_BYTES, "epilogue tile aliases a halo buffer");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int l_row = lane >> 3, l_slot = lane & 7;
    const int ntiles = p.tiles_x * p.tiles_y;

    // ---- this wave's filter rows: A fragments of all 18 k-steps (tap x 32-channel half) --------
    f16x8 wfr[18];
#pragma unroll
    for (int s = 0; s < 18; ++s)
        wfr[s] = *reinterpret_cast<const f16x8 *>(p.wpk + ((size_t)(s >> 1) * p.CoutPad + wave * 16 + l15) * 64 + (s & 1) * 32 + kg * 8);
    const float4 sc = *reinterpret_cast<const float4 *>(p.scale + wave * 16 + 4 * kg);
    const float4 sh = *reinterpret_cast<const float4 *>(p.shift + wave * 16 + 4 * kg);
    const float aslope = act_slope(p.act);

    // ---- halo staging: LDS row q = hy * 33 + col, col = even columns first; chunk swizzle = q & 7 = l_row
    auto issue_tile = [&](int t, int buf) {
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int iy0 = 2 * ty * TH - 1, ix0 = 2 * tx * TW - 1;
        const dma_rsrc_t ra = dma_rsrc(p.src0, (unsigned)p.Hi * (unsigned)p.Wi * (unsigned)p.s0_stride * 2u);
        for (int piece = wave; piece < A_PIECES; piece += NW) {
            const int q = piece * 8 + l_row;
            const int hy = q / HWD, col = q - hy * HWD;
            const int hx = col < NEVEN ? 2 * col : 2 * (col - NEVEN) + 1;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = (q < NPIX) & ((unsigned)iy < (unsigned)p.Hi) & ((unsigned)ix < (unsigned)p.Wi);
            const unsigned off = ((unsigned)(iy * p.Wi + ix) * (unsigned)p.s0_stride + (unsigned)((l_slot ^ l_row) << 3)) * 2u;
            dma16(ra, smem + buf * A_BYTES + piece * 1024, ok ? off : DMA_OOB);       // out of the image: zeros
        }
    };

    // per-lane read offsets: LDS row = c + l15 for a compile-time c; the swizzle needs (c + l15) & 7
    // (the second 32-channel half is the same offset with bit 6 flipped)
    int xo[8];
#pragma unroll
    for (int c7 = 0; c7 < 8; ++c7) xo[c7] = l15 * PIXB + ((kg ^ ((c7 + l15) & 7)) << 4);

    // ---- the fused CondNet2 tail (NW = 12 only): its weights ride in LDS behind the two halo buffers
    const bool tail = NW == 12 && p.tail_w != nullptr && !p.tail_s;
    if (NW == 12 && tail) {
        for (int e = tid; e < 12 * 64; e += NT) reinterpret_cast<f16x8 *>(smem + TAIL_OFF)[e] = reinterpret_cast<const f16x8 *>(p.tail_w)[e];
        if (tid < 96) reinterpret_cast<float *>(smem + TAIL_OFF + TAIL_WB)[tid] = p.tail_b[tid];
    }

    // ---- prologue: tile 0 landed, tile 1 in flight ---------------------------------------------
    int t = blockIdx.x;
    const int step = gridDim.x;
    if (t < ntiles) issue_tile(t, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + step < ntiles) issue_tile(t + step, 1);

    for (int buf = 0; t < ntiles; t += step, buf ^= 1) {
        const char *a = smem + buf * A_BYTES;
        f32x4 acc[TH];
#pragma unroll
        for (int i = 0; i < TH; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

        // 144 MFMAs: k-step s = (tap, half) outer, output row inner; activation reads 4 MFMAs ahead
        auto ldx = [&](int m) -> f16x8 {
            const int s = m >> 3, pt = m & 7;
            const int tap = s >> 1, ks = s & 1, ky = tap / 3, kx = tap % 3;
            const int c = (2 * pt + ky) * HWD + (kx & 1) * NEVEN + (kx >> 1);
            return *reinterpret_cast<const f16x8 *>(a + c * PIXB + (xo[c & 7] ^ (ks << 6)));
        };
        f16x8 xq[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) xq[m] = ldx(m);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 18; ++s) {
#pragma unroll
            for (int pt = 0; pt < 8; ++pt) {
                const int m = s * 8 + pt;
                const f16x8 x = xq[m & 3];
                if (m + 4 < 144) xq[m & 3] = ldx(m + 4);
                __builtin_amdgcn_sched_barrier(0);
                acc[pt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfr[s], x, acc[pt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue: everyone is done reading this halo buffer -> it becomes the output tile
        __syncthreads();
        char *so = smem + buf * A_BYTES;
        int eln = lane, etid = tid;                            // (opaque copies: nothing of the epilogue's addressing is hoisted over the MFMA stream)
        asm volatile("" : "+v"(eln), "+v"(etid));
        const int el15 = eln & 15, ekg = eln >> 4;
#pragma unroll
        for (int pt = 0; pt < TH; ++pt) {
            f16x4 o;
            o[0] = (f16)act_fast(acc[pt][0] * sc.x + sh.x, aslope);
            o[1] = (f16)act_fast(acc[pt][1] * sc.y + sh.y, aslope);
            o[2] = (f16)act_fast(acc[pt][2] * sc.z + sh.z, aslope);
            o[3] = (f16)act_fast(acc[pt][3] * sc.w + sh.w, aslope);
            *reinterpret_cast<f16x4 *>(so + (pt * TW + el15) * OUT_ROWB + (wave * 16 + 4 * ekg) * 2) = o;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // next tile's halo landed (and older stores retired)
        __syncthreads();
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        constexpr int CPP = COUT / 8;
        if (NW == 4 && p.tail_s) {
            // CondNet3.4 (1x1 64 -> 16, no activation) on the staged tile: each of the four waves takes 32 pixels; the 64-channel map has
            // no other reader and is not stored.  conv_igemm's arithmetic: four 16-channel k-steps from zero, then acc * scale + shift
            const int l31 = eln & 31, lh = eln >> 5;
            const int qq = 32 * wave + l31;
            const char *px = so + qq * OUT_ROWB + 16 * lh;
            f32x16 o;
#pragma unroll
            for (int k = 0; k < 16; ++k) o[k] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f16x8 w = *reinterpret_cast<const f16x8 *>(p.tail_w + l31 * 64 + 16 * s + 8 * lh);
                o = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, *reinterpret_cast<const f16x8 *>(px + 32 * s), o, 0, 0, 0);
            }
            const int oy = oy0 + qq / TW, ox = ox0 + qq % TW;
            if (oy < p.Ho && ox < p.Wo) {
                f16x4 lo, hi;
                const float4 s0 = *reinterpret_cast<const float4 *>(p.tail_s + 4 * lh), s1 = *reinterpret_cast<const float4 *>(p.tail_s + 8 + 4 * lh);
                const float4 b0 = *reinterpret_cast<const float4 *>(p.tail_b + 4 * lh), b1 = *reinterpret_cast<const float4 *>(p.tail_b + 8 + 4 * lh);
                // (packed converts, common.h cvt_h4: conv_igemm's epilogue rounds fp32 -> f16 as a step of its own)
                lo = cvt_h4(o[0] * s0.x + b0.x, o[1] * s0.y + b0.y, o[2] * s0.z + b0.z, o[3] * s0.w + b0.w);
                hi = cvt_h4(o[4] * s1.x + b1.x, o[5] * s1.y + b1.y, o[6] * s1.z + b1.z, o[7] * s1.w + b1.w);
                f16 *d = p.tail_out + ((size_t)oy * p.Wo + ox) * 16;
                *reinterpret_cast<f16x4 *>(d + 4 * lh) = lo;
                *reinterpret_cast<f16x4 *>(d + 8 + 4 * lh) = hi;
            }
        } else if (NW == 12 && tail) {
            if (wave < 4) {
                // waves 0..3: CondNet2.2 (1x1 64 -> 64, LeakyReLU) + CondNet2.4 (1x1 64 -> 16) on the 32 pixels 32 wave .. of the tile,
                // B fragments straight from the staged f16 tile (the values cond_tail_kernel would read back from HBM), its MFMA order
                // (per-lane addresses rebuilt from an opaque copy of the lane id: hoisted out of the tile loop they would be spilled --
                // the MFMA stream has no register to spare -- and a scratch reload drains the DMA queue)
                const int oln = eln;
                const int l31 = oln & 31, lh = oln >> 5;
                const int qq = 32 * wave + l31;
                const char *px = so + qq * OUT_ROWB + 16 * lh;
                f16x8 cur[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) cur[s] = *reinterpret_cast<const f16x8 *>(px + 32 * s);
                const f16x8 *tw = reinterpret_cast<const f16x8 *>(smem + TAIL_OFF);
                const float *tb = reinterpret_cast<const float *>(smem + TAIL_OFF + TAIL_WB);
                // (the two halves of the hidden layer one after the other: with the 72-VGPR filter bank live there is no room for both
                // accumulator tiles at once; each tile's own accumulation order is cond_tail_kernel's)
                f16x8 bf[4];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x16 h = tail_bias_tile(tb + 32 * mt, lh);
#pragma unroll
                    for (int s = 0; s < 4; ++s) h = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw[(4 * mt + s) * 64 + oln], cur[s], h, 0, 0, 0);
                    bf[2 * mt] = tail_lrelu_pack(h, 0);
                    bf[2 * mt + 1] = tail_lrelu_pack(h, 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                f32x16 o = tail_bias_tile(tb + 64, lh);
#pragma unroll
                for (int s = 0; s < 4; ++s) o = __builtin_amdgcn_mfma_f32_32x32x16_f16(tw[(8 + s) * 64 + oln], bf[s], o, 0, 0, 0);
                const int oy = oy0 + qq / TW, ox = ox0 + qq % TW;
                if (oy < p.Ho && ox < p.Wo) {
                    f16x4 lo, hi;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { lo[k] = (f16)o[k]; hi[k] = (f16)o[4 + k]; }
                    f16 *d = p.tail_out + ((size_t)oy * p.Wo + ox) * 16;
                    *reinterpret_cast<f16x4 *>(d + 4 * lh) = lo;
                    *reinterpret_cast<f16x4 *>(d + 8 + 4 * lh) = hi;
                }
            } else {
                // waves 4..11: channels 64..191 of the tile (the first 64 have no other reader)
                constexpr int CP2 = CPP - 8;
                for (int e = etid - 256; e < TH * TW * CP2; e += NT - 256) {
                    const int qq = e / CP2, c8 = 8 + e % CP2;
                    const int oy = oy0 + qq / TW, ox = ox0 + qq % TW;
                    if (oy < p.Ho && ox < p.Wo)
                        *reinterpret_cast<f16x8 *>(p.dst + ((size_t)oy * p.Wo + ox) * p.dstC + c8 * 8) =
                            *reinterpret_cast<const f16x8 *>(so + qq * OUT_ROWB + c8 * 16);
                }
            }
        } else {
            for (int e = etid; e < TH * TW * CPP; e += NT) {
                const int qq = e / CPP, c8 = e % CPP;
                const int oy = oy0 + qq / TW, ox = ox0 + qq % TW;
                if (oy < p.Ho && ox < p.Wo)
                    *reinterpret_cast<f16x8 *>(p.dst + ((size_t)oy * p.Wo + ox) * p.dstC + c8 * 8) =
                        *reinterpret_cast<const f16x8 *>(so + qq * OUT_ROWB + c8 * 16);
            }
        }
        __syncthreads();                                       // output tile read out: the buffer is free again
        if (t + 2 * step < ntiles) issue_tile(t + 2 * step, buf);
    }
}

template <int NW>
hipError_t launch_s2(ConvParams p, int n_cu, hipStream_t s)
{
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv3x3s2_preg_kernel<NW>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM + (NW == 12 ? TAIL_SMEM : 0));
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const int ntiles = p.tiles_x * p.tiles_y;
    hipLaunchKernelGGL(kern, dim3(ntiles < n_cu ? ntiles : n_cu), dim3(64 * NW), SMEM + (NW == 12 ? TAIL_SMEM : 0), s, p);
    return hipGetLastError();
}

}  // namespace

// 3x3, stride 2, pad 1, Cin = 64 (src0 only, pixel stride p.s0_stride), Cout == CoutPad in {64, 192}, NHWC store.
hipError_t conv3x3s2_preg_launch(ConvParams p, int n_cu, hipStream_t s)
{
    if (p.c0 != 64 || p.c1 != 0 || p.mode != ST_NHWC || p.res1 || p.res2 || !p.zeros || p.s0_stride < 64 ||
        p.Cout != p.CoutPad || p.dstC < p.Cout || n_cu < 1 || (p.tail_w && (!p.tail_b || !p.tail_out || (p.tail_s ? p.CoutPad != 64 : p.CoutPad != 192))))
        return hipErrorInvalidValue;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    if (p.CoutPad == 64) return launch_s2<4>(p, n_cu, s);
    if (p.CoutPad == 192) return launch_s2<12>(p, n_cu, s);
    return hipErrorInvalidValue;
}
