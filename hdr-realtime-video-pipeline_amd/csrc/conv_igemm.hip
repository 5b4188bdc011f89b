// conv_igemm.hip -- implicit-GEMM convolution on CDNA4 matrix cores (gfx950).
//
// Replaces every nn.Conv2d with Cin >= 16 on the hot path (PyTorch/MIOpen calls in the
// reference: HDRUNet3T1_arch.py:14-55, arch_util.py:78-84, Hallucination_arch.py:59-89).
//
//   D[co][pixel] = sum_{tap, ci} W[co][tap][ci] * X[pixel @ tap][ci]
//
// GEMM view: M = output channels (MFMA A operand = packed weights), N = output pixels of an
// 8x16 tile (MFMA B operand = activations), K = taps x input channels, walked as
// (channel chunk of CIN_T) x (tap).  The activation tile INCLUDING ITS HALO is staged once per
// channel chunk into LDS and re-read for all KS*KS taps at shifted addresses, so HBM/L2 sees
// each input element once per tile (+halo), not 9 times.  Activations are NHWC fp16: a
// pixel's CIN_T channels are one contiguous 64/128-byte run, i.e. exactly the 8-wide k
// fragments v_mfma_f32_32x32x16_f16 wants.  LDS rows are XOR-swizzled per pixel so the
// ds_read_b128 fragment reads of 32 consecutive pixels spread over all banks.
// Accumulation fp32; per-channel scale/shift (bias, folded BatchNorm), activation, residual
// adds, PixelShuffle(2), MaxPool2d(2) and the planar 3-channel head are fused into the
// LDS-staged epilogue, whose global stores are 16 B per lane and fully coalesced.
#include "launchers.h"

namespace {

constexpr int TH = 8, TW = 16;  // output pixels per workgroup tile: 8 rows x 16 cols = 128

template <int CIN_T>
__device__ __forceinline__ int swz(int row)
{
    // 16-byte chunk swizzle: rows that share a 256-byte LDS bank row get distinct chunk slots.
    constexpr int NCH = CIN_T / 8;   // chunks per row
    constexpr int PPR = 16 / NCH;    // rows per 256-byte bank row
    return (row / PPR) % NCH;
}

template <int CIN_T, int BN, int KS, int S>
struct Cfg {
    static constexpr int HH = (TH - 1) * S + KS;
    static constexpr int HW = (TW - 1) * S + KS;
    static constexpr int NPIX = HH * HW;
    static constexpr int NCH = CIN_T / 8;
    static constexpr int PIXB = CIN_T * 2;
    static constexpr int A_BYTES = ((NPIX * PIXB + 255) / 256) * 256;
    static constexpr int B_BYTES = BN * PIXB;
    static constexpr int A_LD = (NPIX * NCH + 255) / 256;  // 16-B loads per thread for the halo
    static constexpr int B_LD = (BN * NCH + 255) / 256;
    static constexpr int OUT_ROWB = BN * 2 + 16;
    static constexpr int OUT_BYTES = TH * TW * OUT_ROWB;
    static constexpr int MAIN_BYTES = 2 * A_BYTES + 2 * B_BYTES;
    static constexpr int SMEM = MAIN_BYTES > OUT_BYTES ? MAIN_BYTES : OUT_BYTES;
    // wave grid: WC waves along channels x WP waves along pixels, MT x NT 32x32 tiles per wave
    static constexpr int WC = (BN >= 64) ? 2 : 1;
    static constexpr int WP = 4 / WC;
    static constexpr int MT = BN / 32 / WC;
    static constexpr int NT = 4 / WP;
};

template <int CIN_T, int BN, int KS, int S>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvParams p)
{
    using C = Cfg<CIN_T, BN, KS, S>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *sA = smem;
    char *sB = smem + 2 * C::A_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-aware, bijective remap of the 1-D grid: the 8 XCDs each get a contiguous run of tiles
    // so neighbouring tiles (shared halo, shared weights) hit the same L2.
    const int nwg = gridDim.x;
    int t;
    {
        const int b = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = b & 7;
        t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const int ntn = p.CoutPad / BN;
    const int nt_i = t % ntn;
    const int sp = t / ntn;
    const int ty = sp / p.tiles_x, tx = sp % p.tiles_x;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int n0 = nt_i * BN;
    constexpr int PAD = KS / 2;
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;

    const int Cin = p.c0 + p.c1;
    const int nchunk = Cin / CIN_T;
    const int nchunk0 = p.c0 / CIN_T;
    constexpr int NTAP = KS * KS;
    const int nit = nchunk * NTAP;

    const int wc = wave % C::WC, wp = wave / C::WC;

    f32x16 acc[C::MT][C::NT];
#pragma unroll
    for (int i = 0; i < C::MT; ++i)
#pragma unroll
        for (int j = 0; j < C::NT; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    uint4 ra[C::A_LD];
    uint4 rb[C::B_LD];

    auto load_A = [&](int cc) {
        const f16 *src;
        int cs, coff;
        if (cc < nchunk0) { src = p.src0; cs = p.s0_stride; coff = cc * CIN_T; }
        else { src = p.src1; cs = p.s1_stride; coff = (cc - nchunk0) * CIN_T; }
#pragma unroll
        for (int it = 0; it < C::A_LD; ++it) {
            const int e = tid + it * 256;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (e < C::NPIX * C::NCH) {
                const int hp = e / C::NCH, c = e % C::NCH;
                const int hy = hp / C::HW, hx = hp % C::HW;
                const int iy = iy0 + hy, ix = ix0 + hx;
                if (iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi)
                    v = *reinterpret_cast<const uint4 *>(src + ((size_t)iy * p.Wi + ix) * cs + coff + c * 8);
            }
            ra[it] = v;
        }
    };
    auto store_A = [&](int buf) {
        char *dst = sA + buf * C::A_BYTES;
#pragma unroll
        for (int it = 0; it < C::A_LD; ++it) {
            const int e = tid + it * 256;
            if (e < C::NPIX * C::NCH) {
                const int hp = e / C::NCH, c = e % C::NCH;
                *reinterpret_cast<uint4 *>(dst + hp * C::PIXB + ((c ^ swz<CIN_T>(hp)) << 4)) = ra[it];
            }
        }
    };
    auto load_B = [&](int it_i) {
        const int cc = it_i / NTAP, tap = it_i % NTAP;
        const f16 *src = p.wpk + ((size_t)(tap * nchunk + cc) * p.CoutPad + n0) * CIN_T;
#pragma unroll
        for (int it = 0; it < C::B_LD; ++it) {
            const int e = tid + it * 256;
            if (e < BN * C::NCH) rb[it] = *reinterpret_cast<const uint4 *>(src + (size_t)e * 8);
        }
    };
    auto store_B = [&](int buf) {
        char *dst = sB + buf * C::B_BYTES;
#pragma unroll
        for (int it = 0; it < C::B_LD; ++it) {
            const int e = tid + it * 256;
            if (e < BN * C::NCH) {
                const int n = e / C::NCH, c = e % C::NCH;
                *reinterpret_cast<uint4 *>(dst + n * C::PIXB + ((c ^ swz<CIN_T>(n)) << 4)) = rb[it];
            }
        }
    };

    // per-lane halo-pixel base index (tap 0) for each of this wave's pixel tiles
    int hp_base[C::NT];
#pragma unroll
    for (int j = 0; j < C::NT; ++j) {
        const int q = (wp * C::NT + j) * 32 + l31;   // tile-local output pixel 0..127
        hp_base[j] = (q / TW) * S * C::HW + (q % TW) * S;
    }
    int wrow[C::MT];
#pragma unroll
    for (int i = 0; i < C::MT; ++i) wrow[i] = (wc * C::MT + i) * 32 + l31;

    // prologue
    load_A(0);
    load_B(0);
    store_A(0);
    store_B(0);
    __syncthreads();

    int abuf = 0;
    for (int it_i = 0; it_i < nit; ++it_i) {
        const int tap = it_i % NTAP;
        const bool has_next = it_i + 1 < nit;
        const bool next_new_chunk = has_next && (tap == NTAP - 1);
        if (has_next) {
            load_B(it_i + 1);
            if (next_new_chunk) load_A((it_i + 1) / NTAP);
        }
        {
            const char *a = sA + abuf * C::A_BYTES;
            const char *b = sB + (it_i & 1) * C::B_BYTES;
            const int ky = tap / KS, kx = tap % KS;
            const int tap_off = ky * C::HW + kx;
#pragma unroll
            for (int ks = 0; ks < CIN_T / 16; ++ks) {
                const int chunk = ks * 2 + lh;
                f16x8 wf[C::MT], xf[C::NT];
#pragma unroll
                for (int i = 0; i < C::MT; ++i)
                    wf[i] = *reinterpret_cast<const f16x8 *>(b + wrow[i] * C::PIXB + ((chunk ^ swz<CIN_T>(wrow[i])) << 4));
#pragma unroll
                for (int j = 0; j < C::NT; ++j) {
                    const int hp = hp_base[j] + tap_off;
                    xf[j] = *reinterpret_cast<const f16x8 *>(a + hp * C::PIXB + ((chunk ^ swz<CIN_T>(hp)) << 4));
                }
#pragma unroll
                for (int i = 0; i < C::MT; ++i)
#pragma unroll
                    for (int j = 0; j < C::NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (has_next) {
            store_B((it_i + 1) & 1);
            if (next_new_chunk) { store_A(abuf ^ 1); }
        }
        __syncthreads();
        if (next_new_chunk) abuf ^= 1;
    }

    // ---------------------------------------------------------------- epilogue (LDS staged)
    // acc[i][j][r]: channel = (wc*MT+i)*32 + (r&3) + 8*(r>>2) + 4*lh, pixel = (wp*NT+j)*32 + l31
    const float aslope = act_slope(p.act);
    char *so = smem;
#pragma unroll
    for (int i = 0; i < C::MT; ++i) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int cl = (wc * C::MT + i) * 32 + 8 * qd + 4 * lh;   // tile-local channel of r&3==0
            const float4 sc = *reinterpret_cast<const float4 *>(p.scale + n0 + cl);
            const float4 sh = *reinterpret_cast<const float4 *>(p.shift + n0 + cl);
#pragma unroll
            for (int j = 0; j < C::NT; ++j) {
                const int q = (wp * C::NT + j) * 32 + l31;
                f16x4 o;
                o[0] = (f16)act_fast(acc[i][j][4 * qd + 0] * sc.x + sh.x, aslope);
                o[1] = (f16)act_fast(acc[i][j][4 * qd + 1] * sc.y + sh.y, aslope);
                o[2] = (f16)act_fast(acc[i][j][4 * qd + 2] * sc.z + sh.z, aslope);
                o[3] = (f16)act_fast(acc[i][j][4 * qd + 3] * sc.w + sh.w, aslope);
                *reinterpret_cast<f16x4 *>(so + q * C::OUT_ROWB + cl * 2) = o;
            }
        }
    }
    __syncthreads();

    constexpr int CPP = BN / 8;  // 16-byte chunks per pixel row
    auto add_res = [&](f16x8 v, const f16 *r, size_t off) {
        const f16x8 rv = *reinterpret_cast<const f16x8 *>(r + off);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (f16)((float)v[k] + (float)rv[k]);
        return v;
    };

    if (p.mode == ST_NHWC) {
        for (int e = tid; e < TH * TW * CPP; e += 256) {
            const int q = e / CPP, c8 = e % CPP;
            const int oy = oy0 + q / TW, ox = ox0 + q % TW;
            const int ch = n0 + c8 * 8;
            if (oy < p.Ho && ox < p.Wo && ch < p.Cout) {
                f16x8 v = *reinterpret_cast<const f16x8 *>(so + q * C::OUT_ROWB + c8 * 16);
                const size_t off = ((size_t)oy * p.Wo + ox) * p.dstC + ch;
                if (p.res1) v = add_res(v, p.res1, off);
                if (p.res2) v = add_res(v, p.res2, off);
                *reinterpret_cast<f16x8 *>(p.dst + off) = v;
            }
        }
    } else if (p.mode == ST_PS) {
        // packed output channel n' = sub*Cps + c, sub = 2*i + j  ->  dst[2*oy+i][2*ox+j][c]
        const int cps = p.dstC;
        for (int e = tid; e < TH * TW * CPP; e += 256) {
            const int q = e / CPP, c8 = e % CPP;
            const int oy = oy0 + q / TW, ox = ox0 + q % TW;
            const int ch = n0 + c8 * 8;
            if (oy < p.Ho && ox < p.Wo && ch < p.Cout) {
                const int sub = ch / cps, c = ch % cps;
                const int Y = 2 * oy + (sub >> 1), X = 2 * ox + (sub & 1);
                if (Y < p.Hd && X < p.Wd) {
                    f16x8 v = *reinterpret_cast<const f16x8 *>(so + q * C::OUT_ROWB + c8 * 16);
                    const size_t off = ((size_t)Y * p.Wd + X) * cps + c;
                    if (p.res1) v = add_res(v, p.res1, off);
                    if (p.res2) v = add_res(v, p.res2, off);
                    *reinterpret_cast<f16x8 *>(p.dst + off) = v;
                }
            }
        }
    } else if (p.mode == ST_POOL) {
        if (p.dst_full) {
            for (int e = tid; e < TH * TW * CPP; e += 256) {
                const int q = e / CPP, c8 = e % CPP;
                const int oy = oy0 + q / TW, ox = ox0 + q % TW;
                const int ch = n0 + c8 * 8;
                if (oy < p.Ho && ox < p.Wo && ch < p.Cout)
                    *reinterpret_cast<f16x8 *>(p.dst_full + ((size_t)oy * p.Wo + ox) * p.Cout + ch) =
                        *reinterpret_cast<const f16x8 *>(so + q * C::OUT_ROWB + c8 * 16);
            }
        }
        for (int e = tid; e < (TH / 2) * (TW / 2) * CPP; e += 256) {
            const int pq = e / CPP, c8 = e % CPP;
            const int py = pq / (TW / 2), px = pq % (TW / 2);
            const int oy = oy0 / 2 + py, ox = ox0 / 2 + px;
            const int ch = n0 + c8 * 8;
            if (oy < p.Hd && ox < p.Wd && ch < p.Cout) {
                const int q00 = (2 * py) * TW + 2 * px;
                f16x8 v = *reinterpret_cast<const f16x8 *>(so + q00 * C::OUT_ROWB + c8 * 16);
                const f16x8 v1 = *reinterpret_cast<const f16x8 *>(so + (q00 + 1) * C::OUT_ROWB + c8 * 16);
                const f16x8 v2 = *reinterpret_cast<const f16x8 *>(so + (q00 + TW) * C::OUT_ROWB + c8 * 16);
                const f16x8 v3 = *reinterpret_cast<const f16x8 *>(so + (q00 + TW + 1) * C::OUT_ROWB + c8 * 16);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    f16 m = v[k] > v1[k] ? v[k] : v1[k];
                    const f16 m2 = v2[k] > v3[k] ? v2[k] : v3[k];
                    v[k] = m > m2 ? m : m2;
                }
                *reinterpret_cast<f16x8 *>(p.dst + ((size_t)oy * p.Wd + ox) * p.dstC + ch) = v;
            }
        }
    } else {  // ST_PLANAR3: channels 0..2 -> planar f16 [3][Hd][Wd] (+ planar residual)
        for (int e = tid; e < TH * TW * 3; e += 256) {
            const int ch = e / (TH * TW), q = e % (TH * TW);
            const int oy = oy0 + q / TW, ox = ox0 + q % TW;
            if (oy < p.Hd && ox < p.Wd && oy < p.Ho && ox < p.Wo) {
                float v = (float)*reinterpret_cast<const f16 *>(so + q * C::OUT_ROWB + ch * 2);
                const size_t off = (size_t)ch * p.Hd * p.Wd + (size_t)oy * p.Wd + ox;
                if (p.res_planar) v += (float)p.res_planar[off];
                p.dst_planar[off] = (f16)v;
            }
        }
    }
}

template <int CIN_T, int BN, int KS, int S>
hipError_t launch_cfg(const ConvParams &p, hipStream_t stream)
{
    using C = Cfg<CIN_T, BN, KS, S>;
    static DevOnce attr_once;   // hipFuncSetAttribute is per (function, device)
    auto kern = conv_igemm_kernel<CIN_T, BN, KS, S>;
    if (attr_once.need()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM);
        if (e != hipSuccess) return e;
        attr_once.done();
    }
    const int grid = p.tiles_x * p.tiles_y * (p.CoutPad / BN);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), C::SMEM, stream, p);
    return hipGetLastError();
}

}  // namespace

// Host-side dispatcher.  cin_t in {32,64}, bn in {32,64,128}, ks in {1,3}, stride in {1,2}.
// Returns hipErrorInvalidValue for a combination that is not instantiated.
hipError_t conv_igemm_launch(ConvParams p, int cin_t, int bn, int ks, int stride, hipStream_t stream)
{
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
#define HDRTV_CASE(CT, BN_, KS_, S_) \
    if (cin_t == CT && bn == BN_ && ks == KS_ && stride == S_) return launch_cfg<CT, BN_, KS_, S_>(p, stream);
    HDRTV_CASE(32, 32, 3, 2)
    HDRTV_CASE(64, 32, 1, 1)
    HDRTV_CASE(64, 64, 1, 1)
#undef HDRTV_CASE
    return hipErrorInvalidValue;
}
