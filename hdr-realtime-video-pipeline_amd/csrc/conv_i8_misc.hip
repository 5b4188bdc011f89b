// conv_i8_misc.hip -- conv1x1_i8, the small kernel beside conv3x3_pglds_i8.hip on the W8A8 HG path: the decoder's 1x1
// fuse convolutions conv6 .. conv9 (Hallucination_arch.py:118-133) on v_mfma_i32_16x16x64_i8.  K = the concatenation of
// two int8 tensors that share one quantiser, no activation; the output is re-quantised to the next layer's (signed-range)
// codes (or real values as f16 for an fp16 reader).  HBM-bound: 128 pixels x 128 output channels per block, 128-channel
// K chunks staged by LDS-DMA into two stages, one barrier per chunk, two blocks per CU.
// (The fp16 -> int8 boundary at the other end is conv2's store mode ST_NHWC_Q8, conv3x3_pglds.hip.)
#include "launchers.h"

namespace {


// 128 pixels x 128 output channels per block, two stages of {A 16 KiB, B 16 KiB}: 64 KiB, so TWO blocks share a CU and
// one block's loads and stores overlap the other's MFMAs (with one 144-KiB block per CU the phases ran back to back)
constexpr int TP = 128, BN = 128, CT = 128;           // pixels x output channels per block, K chunk (bytes per row)
constexpr int A_BYTES = TP * CT, B_BYTES = BN * CT;   // 16 KiB + 16 KiB per stage
constexpr int STAGE = A_BYTES + B_BYTES, NSTAGE = 2;
constexpr int SMEM = NSTAGE * STAGE;                  // 64 KiB

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool OUTF16>
__global__ __launch_bounds__(512) void conv1x1_i8_kernel(ConvI8Params p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int ntn = p.Cout / BN;
    const int n0 = (blockIdx.x % ntn) * BN;           // channel tiles of one pixel tile are neighbours: its K rows stay in L2
    const size_t px0 = (size_t)(blockIdx.x / ntn) * TP;
    const size_t npx = (size_t)p.Hi * p.Wi;
    const int nchunk0 = p.c0 / CT, nchunk = (p.c0 + p.c1) / CT;

    const int l_row = lane >> 3, l_slot = lane & 7;
    auto issue = [&](int cc, int stage) {
        const int8_t *src;
        int cs, coff;
        if (cc < nchunk0) { src = p.src0; cs = p.c0; coff = cc * CT; }
        else { src = p.src1; cs = p.c1; coff = (cc - nchunk0) * CT; }
        char *sa = smem + stage * STAGE, *sb = sa + A_BYTES;
        // LDS-DMA as buffer loads (common.h); pixels past the end of the tensor are zeros (their results are never stored)
        const dma_rsrc_t ra = dma_rsrc(src, (unsigned)npx * (unsigned)cs);
#pragma unroll
        for (int it = 0; it < 2; ++it) {              // 16 pieces of 8 pixel rows
            const int piece = wave * 2 + it;
            const int r = piece * 8 + l_row;
            const size_t px = px0 + r;
            dma16(ra, sa + piece * 1024, px < npx ? (unsigned)px * (unsigned)cs + (unsigned)(coff + ((l_slot ^ (r & 7)) << 4)) : DMA_OOB);
        }
        const dma_rsrc_t rb = dma_rsrc(p.wpk, (unsigned)nchunk * (unsigned)p.Cout * (unsigned)CT);
        const unsigned so = (unsigned)(cc * p.Cout + n0) * (unsigned)CT;
#pragma unroll
        for (int it = 0; it < 2; ++it) {              // 16 pieces of 8 weight rows
            const int piece = wave * 2 + it;
            const int n = piece * 8 + l_row;
            dma16(rb, sb + piece * 1024, (unsigned)(n * CT + ((l_slot ^ (n & 7)) << 4)), so);
        }
    };

    // wave tiling: 2 (channels) x 4 (pixel groups of 32), 4x2 tiles of 16x16 each
    const int wc = wave & 1, wp = wave >> 1;
    i32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = i32x4{0, 0, 0, 0};
    const int kw = l15 & 7;
    const int b_lane = (wc * 64 + l15) * CT, a_lane = (wp * 32 + l15) * CT;
    // f16 output (conv9: 64 real channels in a 128-wide tile): waves whose 64 channels are padding only stage data
    const bool live = n0 + wc * 64 < p.dstC;

    issue(0, 0);
    for (int cc = 0; cc < nchunk; ++cc) {
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();                 // chunk cc is in LDS for everyone; the other stage is no longer read
        if (cc + 1 < nchunk) issue(cc + 1, (cc + 1) & 1);
        const char *a = smem + (cc & 1) * STAGE + a_lane, *b = smem + (cc & 1) * STAGE + A_BYTES + b_lane;
        if (!live) continue;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            i32x4 wf[4], xf[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const i32x4 *>(b + i * 16 * CT + (((ks * 4 + kg) ^ kw) << 4));
#pragma unroll
            for (int j = 0; j < 2; ++j) xf[j] = *reinterpret_cast<const i32x4 *>(a + j * 16 * CT + (((ks * 4 + kg) ^ kw) << 4));
#pragma unroll
            for (int m = 0; m < 8; ++m)
                acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[m >> 1], xf[m & 1], acc[m >> 1][m & 1], 0, 0, 0);
        }
    }
    __syncthreads();                                   // LDS is free: wave-private strips for the 16-byte stores

    // lane: pixel wp*32 + j*16 + l15, channels wc*64 + i*16 + 4*kg + {0..3}
    const int cw = n0 + wc * 64 + 4 * kg;
    if constexpr (OUTF16) {
        // real-valued f16 output for an fp16 consumer: no activation, no re-quantisation
        if (!live) return;
        constexpr int SPH = 144;
        char *sth = smem + wave * (32 * SPH);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 sc = *reinterpret_cast<const float4 *>(p.scale + cw + i * 16);
            const float4 sh = *reinterpret_cast<const float4 *>(p.shift + cw + i * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f16x4 o;
                o[0] = (f16)((float)acc[i][j][0] * sc.x + sh.x);
                o[1] = (f16)((float)acc[i][j][1] * sc.y + sh.y);
                o[2] = (f16)((float)acc[i][j][2] * sc.z + sh.z);
                o[3] = (f16)((float)acc[i][j][3] * sc.w + sh.w);
                *reinterpret_cast<f16x4 *>(sth + (j * 16 + l15) * SPH + (i * 16 + 4 * kg) * 2) = o;
            }
        }
        const int h_px = lane >> 3, h_chunk = lane & 7;
        f16 *dsth = reinterpret_cast<f16 *>(p.dst);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const f16x8 v = *reinterpret_cast<const f16x8 *>(sth + (rr * 8 + h_px) * SPH + h_chunk * 16);
            const size_t px = px0 + wp * 32 + rr * 8 + h_px;
            if (px < npx) *reinterpret_cast<f16x8 *>(dsth + px * p.dstC + n0 + wc * 64 + h_chunk * 8) = v;
        }
        return;
    }
    if (!live) return;
    constexpr int SP = 80;
    char *stg = smem + wave * 2560;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 sc = *reinterpret_cast<const float4 *>(p.scale + cw + i * 16);
        float4 sh = *reinterpret_cast<const float4 *>(p.shift + cw + i * 16);
        sh.x += 128.f; sh.y += 128.f; sh.z += 128.f; sh.w += 128.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // int8 code clamp(rint(acc * scale + shift), -128, 127) as the u8 code 128 higher: v_cvt_pk_u8_f32 rounds to
            // nearest even, saturates and packs (common.h quant4), xor 0x80 per byte brings it back to int8
            unsigned pk = 0;
            pk = __builtin_amdgcn_cvt_pk_u8_f32((float)acc[i][j][0] * sc.x + sh.x, 0, pk);
            pk = __builtin_amdgcn_cvt_pk_u8_f32((float)acc[i][j][1] * sc.y + sh.y, 1, pk);
            pk = __builtin_amdgcn_cvt_pk_u8_f32((float)acc[i][j][2] * sc.z + sh.z, 2, pk);
            pk = __builtin_amdgcn_cvt_pk_u8_f32((float)acc[i][j][3] * sc.w + sh.w, 3, pk);
            *reinterpret_cast<unsigned *>(stg + (j * 16 + l15) * SP + i * 16 + 4 * kg) = pk ^ 0x80808080u;
        }
    }
    const int s_px = lane >> 2, s_chunk = lane & 3;
    int8_t *dst = reinterpret_cast<int8_t *>(p.dst);
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const i32x4 v = *reinterpret_cast<const i32x4 *>(stg + (rr * 16 + s_px) * SP + s_chunk * 16);
        const size_t px = px0 + wp * 32 + rr * 16 + s_px;
        if (px < npx) *reinterpret_cast<i32x4 *>(dst + px * p.dstC + n0 + wc * 64 + s_chunk * 16) = v;
    }
}

}  // namespace

// 1x1 on int8 codes, NHWC, src0 [+ src1] channels multiples of 128, Cout multiple of 128; dst int8 codes [Hi*Wi][dstC], or
// (out_f16) real values as f16; dstC = the real channel count (Cout, or Cout - 64: the last tile's upper half is padding).
hipError_t conv1x1_i8_launch(ConvI8Params p, hipStream_t stream)
{
    if ((p.c0 % CT) || (p.c1 % CT) || p.c0 + p.c1 < CT || (p.Cout % BN) || !p.padline || p.mode != ST_NHWC ||
        (p.dstC % 64) || (p.dstC < p.Cout && p.dstC + 64 != p.Cout))
        return hipErrorInvalidValue;
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_i8_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv1x1_i8_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const size_t npx = (size_t)p.Hi * p.Wi;
    const int grid = (int)((npx + TP - 1) / TP) * (p.Cout / BN);
    if (p.out_f16) hipLaunchKernelGGL(conv1x1_i8_kernel<true>, dim3(grid), dim3(512), SMEM, stream, p);
    else hipLaunchKernelGGL(conv1x1_i8_kernel<false>, dim3(grid), dim3(512), SMEM, stream, p);
    return hipGetLastError();
}
