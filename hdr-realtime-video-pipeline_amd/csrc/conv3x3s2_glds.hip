// conv3x3s2_glds.hip -- 3x3 / stride-2 convolution from 64 channels (gfx950): the first layers of
// LE's CondNet2/3/4 (HDRUNet3T1_arch.py:47-55), merged into ONE launch with 192 output channels
// that reads the 64-channel full-resolution condition map once, and the 64->64 second layers.
//
// Same pipeline as conv3x3_glds.hip (LDS-DMA staging, 3-slot weight ring two taps ahead, counted
// s_waitcnt vmcnt + one raw s_barrier per tap).  Cin = 64 is one channel chunk, so the (17 x 33
// pixel) input halo of an 8x16 output tile is staged once and the K loop is the 9 taps.
#include "launchers.h"

namespace {

constexpr int TH = 8, TW = 16;
constexpr int HH = 2 * TH + 1, HWD = 2 * TW + 1, NPIX = HH * HWD;   // 17 x 33 = 561 halo pixels
constexpr int PIXB = 128;
constexpr int A_PER_WAVE = 9, A_BYTES = 8 * A_PER_WAVE * 1024;      // 72 KiB (561 px + dummy tail)

__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ void glds16(const void *g, void *lds)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds, 16, 0, 0);
}

template <int NCT>   // 32-channel tiles per wave; the block computes COUT_T = 64*NCT channels
struct S2 {
    static constexpr int COUT_T = 64 * NCT;
    static constexpr int B_BYTES = COUT_T * PIXB;
    static constexpr int B_PER_WAVE = COUT_T / 64;          // 1-KiB pieces per wave per tap
    static constexpr int SMEM = A_BYTES + 3 * B_BYTES;
    static constexpr int OUT_ROWB = COUT_T * 2 + 16;
    static_assert(TH * TW * OUT_ROWB <= A_BYTES, "epilogue tile aliases the halo buffer");
};

template <int NCT>
__global__ __launch_bounds__(512) void conv3x3s2_glds_kernel(ConvParams p)
{
    using C = S2<NCT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem, *sB = smem + A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int l_row = lane >> 3, l_slot = lane & 7;

    const int nwg = gridDim.x;
    int t;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int ty = t / p.tiles_x, tx = t % p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;

    auto issue_B = [&](int tap, int slot) {
        const f16 *base = p.wpk + (size_t)tap * p.CoutPad * 64;
#pragma unroll
        for (int k = 0; k < C::B_PER_WAVE; ++k) {
            const int piece = wave * C::B_PER_WAVE + k;
            const int n = piece * 8 + l_row;
            glds16(base + (size_t)n * 64 + ((l_slot ^ swz64(n)) << 3), sB + slot * C::B_BYTES + piece * 1024);
        }
    };
    // prologue: whole halo tile + weights of taps 0 and 1
#pragma unroll
    for (int it = 0; it < A_PER_WAVE; ++it) {
        const int piece = wave + it * 8;
        const int hp = piece * 8 + l_row;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = hp < NPIX && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const f16 *g = ok ? p.src0 + ((size_t)iy * p.Wi + ix) * p.s0_stride + ((l_slot ^ swz64(hp)) << 3) : p.zeros + (l_slot << 3);
        glds16(g, sA + piece * 1024);
    }
    issue_B(0, 0);
    issue_B(1, 1);
    if (C::B_PER_WAVE == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // waves: 4 pixel tiles (2 output rows x 16) x 2 channel groups of NCT tiles
    const int pt = wave & 3, cg = wave >> 2;
    const int q = pt * 32 + l31;
    const int hp_base = (2 * (q / TW)) * HWD + 2 * (q % TW);
    f32x16 acc[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap + 2 < 9) issue_B(tap + 2, (tap + 2) % 3);
        const char *b = sB + (tap % 3) * C::B_BYTES;
        const int hp = hp_base + (tap / 3) * HWD + (tap % 3);
        // fragments one k-step ahead of their MFMAs, read/MFMA order pinned (see conv3x3_glds.hip)
        f16x8 xf[2], wf[2][NCT];
        auto ldfrag = [&](int ks) {
            const int chunk = ks * 2 + lh;
            xf[ks & 1] = *reinterpret_cast<const f16x8 *>(sA + hp * PIXB + ((chunk ^ swz64(hp)) << 4));
#pragma unroll
            for (int i = 0; i < NCT; ++i) {
                const int n = (cg * NCT + i) * 32 + l31;
                wf[ks & 1][i] = *reinterpret_cast<const f16x8 *>(b + n * PIXB + ((chunk ^ swz64(n)) << 4));
            }
        };
        ldfrag(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks + 1 < 4) ldfrag(ks + 1);
#pragma unroll
            for (int i = 0; i < NCT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks & 1][i], xf[ks & 1], acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 1 + NCT, 0);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + NCT, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NCT, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, NCT, 0);
        if (tap + 2 < 9) {
            if (C::B_PER_WAVE == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue through LDS (aliases the halo buffer)
    const float aslope = act_slope(p.act);
    char *so = smem;
#pragma unroll
    for (int i = 0; i < NCT; ++i)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int cl = (cg * NCT + i) * 32 + 8 * qd + 4 * lh;
            const float4 sc = *reinterpret_cast<const float4 *>(p.scale + cl);
            const float4 sh = *reinterpret_cast<const float4 *>(p.shift + cl);
            f16x4 o;
            o[0] = (f16)act_fast(acc[i][4 * qd + 0] * sc.x + sh.x, aslope);
            o[1] = (f16)act_fast(acc[i][4 * qd + 1] * sc.y + sh.y, aslope);
            o[2] = (f16)act_fast(acc[i][4 * qd + 2] * sc.z + sh.z, aslope);
            o[3] = (f16)act_fast(acc[i][4 * qd + 3] * sc.w + sh.w, aslope);
            *reinterpret_cast<f16x4 *>(so + q * C::OUT_ROWB + cl * 2) = o;
        }
    __syncthreads();
    constexpr int CPP = C::COUT_T / 8;
    for (int e = tid; e < TH * TW * CPP; e += 512) {
        const int qq = e / CPP, c8 = e % CPP;
        const int oy = oy0 + qq / TW, ox = ox0 + qq % TW;
        if (oy < p.Ho && ox < p.Wo && c8 * 8 < p.Cout)
            *reinterpret_cast<f16x8 *>(p.dst + ((size_t)oy * p.Wo + ox) * p.dstC + c8 * 8) =
                *reinterpret_cast<const f16x8 *>(so + qq * C::OUT_ROWB + c8 * 16);
    }
}

template <int NCT>
hipError_t launch_s2(ConvParams p, hipStream_t s)
{
    using C = S2<NCT>;
    static bool attr_set = false;
    auto kern = conv3x3s2_glds_kernel<NCT>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(p.tiles_x * p.tiles_y), dim3(512), C::SMEM, s, p);
    return hipGetLastError();
}

}  // namespace

// 3x3, stride 2, pad 1, Cin = 64 (src0 only, pixel stride p.s0_stride), CoutPad in {64, 192}, NHWC store.
hipError_t conv3x3s2_glds_launch(ConvParams p, hipStream_t s)
{
    if (p.c0 != 64 || p.c1 != 0 || p.mode != ST_NHWC || p.res1 || p.res2 || !p.zeros || p.s0_stride < 64) return hipErrorInvalidValue;
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    if (p.CoutPad == 64) return launch_s2<1>(p, s);
    if (p.CoutPad == 192) return launch_s2<3>(p, s);
    return hipErrorInvalidValue;
}
