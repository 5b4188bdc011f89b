// common.h -- shared device/host declarations for libhdrtv_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define HDRTV_WAVE 64

// Epilogue activation codes
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU01 = 2 };
// Conv store modes
enum { ST_NHWC = 0, ST_PS = 1, ST_POOL = 2, ST_PLANAR3 = 3, ST_PS_DOT3 = 4 };

__device__ __forceinline__ float act_apply(float v, int act)
{
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_LRELU01) return v >= 0.f ? v : v * 0.1f;
    return v;
}
// Branch-free form for hot epilogues: act(v) = max(v, slope*v) with slope = 1 (none), 0 (ReLU),
// 0.1 (LeakyReLU 0.1).  A runtime `act` switch per element costs two scalar branches each.
__host__ __device__ __forceinline__ float act_slope(int act) { return act == ACT_RELU ? 0.f : (act == ACT_LRELU01 ? 0.1f : 1.f); }
__device__ __forceinline__ float act_fast(float v, float slope) { return fmaxf(v, slope * v); }
// Four fp32 values -> f16x4 through the PACKED convert (v_cvt_pk_f16_f32: each value rounded fp32 -> f16, nearest even).  A scalar
// (f16) cast of an FMA's result lets hipcc fuse the two into v_fma_mixlo_f16, which rounds the exact product-sum ONCE, to f16: a few
// values per million then differ from the two-step form by one f16 step -- and a kernel and its twin must not differ (round 5: the
// per-layer SFT convs against the fused row kernels, CondNet3.4's fused tail against conv_igemm, once the SLP vectoriser no longer
// turned the scalar casts into packed converts by itself).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x4 cvt_h4(float a, float b, float c, float d)
{
    const f16x2_t lo = __builtin_convertvector(f32x2_t{a, b}, f16x2_t), hi = __builtin_convertvector(f32x2_t{c, d}, f16x2_t);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}

// Four activations -> four int8 codes of the reference's u8 activation quantiser (W8A8Conv2d.forward,
// hdrtvnet_torch.py:353-356): q = clamp(rint((x - x_zero) / x_scale), 0, 255) evaluated as one FMA (inv = 1 / x_scale,
// zoff = -x_zero / x_scale; symmetric layers: zoff = 128), code = q - 128, byte k of the result = value k.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
// u8 codes clamp(rint(x * inv + zoff), 0, 255) of four values, as int8 codes q - 128.  v_cvt_pk_u8_f32 rounds to nearest
// even and saturates to [0, 255] by itself: a separate v_rndne + v_med3 in front of it changes no result (tools/
// cvt_pk_u8_probe.hip compares both forms on the GPU over every half-integer tie from -8 to 262, +-inf, +-1e9 and 200 000
// values in between: 0 differ), so the quantiser is one FMA + one convert per value.
__device__ __forceinline__ unsigned quant4(float a, float b, float c, float d, float inv, float zoff)
{
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(a, inv, zoff), 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(b, inv, zoff), 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(c, inv, zoff), 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(d, inv, zoff), 3, w);
    return w ^ 0x80808080u;                       // u8 code q -> int8 code q - 128
}

// the same for values that already are code + offset (a chain's constants have 1 / x_scale and the zero point folded in)
__device__ __forceinline__ unsigned quant4u(float a, float b, float c, float d)
{
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, w);
    w = __builtin_amdgcn_cvt_pk_u8_f32(d, 3, w);
    return w ^ 0x80808080u;
}

// ---- LDS-DMA as a BUFFER load (buffer_load_dwordx4 ... lds).  Not global_load_lds: that one is a FLAT-encoded instruction,
// which hipcc's waitcnt pass books as "may access LDS and memory"; from then on it never counts -- every wait it inserts is
// lgkmcnt(0) / vmcnt(0).  With the buffer form LDS reads get counted waits.  A lane whose byte offset lies outside the
// resource's num_records writes ZEROS to LDS (tools/lds_dma_oob_probe.hip), so image borders need no zero line: DMA_OOB.
// dma16<OFF>: OFF is an immediate (<= 4095) added to both the memory and the LDS address; voff per lane, soff wave-uniform.
typedef __amdgpu_buffer_rsrc_t dma_rsrc_t;
constexpr unsigned DMA_OOB = 0x80000000u;                // beyond every tensor here (all < 2 GiB)
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const void *base, unsigned bytes = 0x80000000u)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), (short)0, (int)bytes, 0x00020000);
}
template <int OFF = 0> __device__ __forceinline__ void dma16(dma_rsrc_t r, void *lds, unsigned voff, unsigned soff = 0)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds, 16, voff, soff, OFF, 0);
}

// K-dimension permutation that lets a 32x32 MFMA accumulator tile be re-used, packed to f16,
// as the B operand of the next MFMA (cdna_hip_programming.md section 3, "An accumulator tile as
// the next MFMA's operand"): operand slot p (0..15) of a 16-wide k-step holds logical k
// perm16[p].  Weight packers apply it to the K (input-channel) axis of chained layers.
__host__ __device__ __forceinline__ int acc_kperm16(int p)
{
    return ((p >> 3) << 2) | (((p >> 2) & 1) << 3) | (p & 3);
}

// ------------------------------------------------------------------------------------------
// Parameter block of the implicit-GEMM convolution (conv_igemm.hip).
struct ConvParams {
    const f16 *src0;   // NHWC, c0 channels
    const f16 *src1;   // NHWC, c1 channels (channel concat after src0) or nullptr
    int c0, c1;        // multiples of the kernel's CIN_T
    int s0_stride, s1_stride;   // elements per pixel of src0 / src1 (>= c0 / c1: channel slices of wider tensors)
    int Hi, Wi;        // input spatial size
    int Ho, Wo;        // conv output spatial size
    const f16 *wpk;    // [KS*KS][Cin/CIN_T][CoutPad][CIN_T]
    const float *scale;  // [CoutPad] per-channel multiplier (folded BatchNorm, else 1)
    const float *shift;  // [CoutPad] per-channel offset (bias, folded)
    int CoutPad;       // multiple of BN
    int Cout;          // real output channels (<= CoutPad)
    int act;
    int mode;
    f16 *dst;          // NHWC [Hd][Wd][dstC]           (ST_NHWC, ST_PS, ST_POOL)
    f16 *dst_full;     // ST_POOL only: optional un-pooled NHWC copy [Ho][Wo][Cout]
    int dstC;
    int Hd, Wd;        // dst spatial size (crop bound for ST_PS; Ho/2,Wo/2 for ST_POOL)
    const f16 *res1;   // optional residuals, same layout/shape as dst
    const f16 *res2;
    f16 *dst_planar;       // ST_PLANAR3: f16 [3][Hd][Wd]
    const f16 *res_planar; // ST_PLANAR3: residual planes
    int tiles_x, tiles_y;
    const f16 *zeros;      // >= 256 B of zeros (source of out-of-image halo pixels for LDS-DMA staging)
    const float *dotw;     // ST_PS_DOT3: [3][dstC] weights of the 1x1 conv fused behind the pixel shuffle
    float *dst_dot;        // ST_PS_DOT3: f32 [Hd][Wd][4] partial sums (x,y,z used)
    void *trash;           // conv_pglds: >= 2 KiB scratch that out-of-image lanes store to (never read)
    int nt_slow;           // conv_pglds tile order: 0 = Cout-tile fastest (an XCD shares halos), 1 = Cout-tile slowest (shares a weight slab)    // conv3x3s2_preg<192> only: CondNet2.{2,4} (cond_tail_kernel's chain, le_fused.hip) computed from output channels 0..63 while the
    // tile is in LDS -- those channels are then not stored.  tail_w: its 12 fragments, tail_b: [64] + [32] bias, tail_out: NHWC 16
    // conv3x3s2_preg<64>: tail_s != null selects the other fused tail -- ONE 1x1 layer without activation (CondNet3.4, conv_igemm's
    // arithmetic: K order 16-channel steps from a zero accumulator, then acc * tail_s + tail_b) on ALL 64 output channels, which
    // are then not stored at all.  tail_w: that layer's packed weights [32][64]
    const f16 *tail_w;
    const float *tail_b;
    const float *tail_s;
    f16 *tail_out;
};

// Parameter block of the int8 HG convolutions (conv3x3_pglds_i8.hip, conv_i8_misc.hip).  Activations are int8 codes
// c = q - 128 of the reference's u8 quantiser q = round((x - x_zero) / x_scale), NHWC, pixel stride = channel count.
struct ConvI8Params {
    const int8_t *src0, *src1;   // src1: channel concat after src0, or nullptr
    int c0, c1;                  // multiples of 128
    int Hi, Wi, Ho, Wo;
    const int8_t *wpk;           // [KS*KS][Cin/128][Cout][128]
    const float *scale, *shift;  // [Cout]: out = acc * scale + shift, in output codes (int8 dst) or real units (f16 dst)
    int Cout;                    // multiple of 128
    int mode;                    // ST_NHWC / ST_PS / ST_POOL
    int out_f16;                 // real-valued output instead of int8 codes: 1x1 -> f16 dst; 3x3 ST_PS_DOT3 -> dst_dot
    const float *dotw;           // ST_PS_DOT3: [3][dstC] weights of the 1x1 conv fused behind the pixel shuffle
    float *dst_dot;              // ST_PS_DOT3: f32 [Hd][Wd][4] partial sums (x,y,z used)
    void *dst;
    int dstC, Hd, Wd;
    int tiles_x, tiles_y;
    const int8_t *padline;       // 128 B of the input tensor's zero-point code (k - 128): out-of-image halo pixels
    void *trash;                 // >= 2 KiB scratch that out-of-image lanes store to (never read)
    // Float zero point (x_zero not an integer multiple of x_scale): padline holds code 0 -- padded taps add nothing to the
    // sum -- and delta[16 border classes][Cout] is added to `shift` for pixels whose 3x3 window leaves the image (the constant
    // the missing taps would otherwise contribute; class = (row bits) << 2 | (column bits), bit0 first tap outside, bit1 last)
    const float *delta;
    const int *delta_acc;        // the same constant in accumulator units (rounded), for the ST_PS_DOT3 epilogue
    float lo_clamp;              // lowest output code: -128, or the code of 0.0 behind a ReLU whose reader has x_zero < 0
};

// Parameter block of the persistent 32-channel conv (conv32p.hip): 3x3, stride 1, Cin = 32.
struct Conv32Params {
    const f16 *src;        // NHWC 32
    const f16 *cond;       // NHWC 16 when an SFT layer is fused in front, else nullptr
    const f16 *sft_wfrag;  // SFT A-fragments [3][64 lanes][8]: hidden stack (scale16|shift16), scale-out, shift-out
    const float *sft_bias; // [32 hidden] [32 scale-out] [32 shift-out]
    int H, W;              // input == conv output spatial size
    const f16 *wpk;        // [9][CoutPad][32]
    const float *scale, *shift;
    int CoutPad, Cout, act, mode;
    f16 *dst;
    int dstC, Hd, Wd;
    const f16 *res1, *res2;
    f16 *dst_planar;
    const f16 *res_planar;
    const f16 *zeros;
    f16 *dump;             // diagnostic builds (make STAMP=1): per-phase cycle sums
    char *trash;           // conv32s: >= 8 KiB write-only scratch that masked-off lanes store to (keeps store counts exact)
    // conv32s, HR_conv1 only: conv_first fused in front (c3_img != nullptr; src is then unused): the planar 3-channel image,
    // and conv_first's three A fragments [3 kernel rows][64 lanes][8] (pack_c3's K order, bias in the centre tap's 4th channel)
    const f16 *c3_img, *c3_wfrag;
    int tiles_x, tiles_y;
    // W8A8 layer (wpk8 != nullptr): int8 weights [9][CoutPad][32] with the K axis in code-tile order
    // (byte 16h + 4qd + k = input channel 8qd + 4h + k), scale[CoutPad], shift[16 border classes][CoutPad], and the
    // input quantiser as u8 code = clamp(rint(x * q_inv + q_zoff), 0, 255), int8 code = u8 ^ q_flip per byte
    const int8_t *wpk8;
    float q_inv, q_zoff;
    // W8A8 SFT convs in front (sq_wfrag != nullptr, needs wpk8 and cond): three int8 A fragments [3][64 lanes][16 B]
    // (both first layers block-diagonal over K = 32; scale-out; shift-out), 6 x 32 dequantisation constants
    // [set][lane half][16] and the quantisers: sq_inv / sq_zoff of the condition map per branch (0 scale, 1 shift),
    // sq_hzoff of the hidden activations per branch (their 1 / x_scale is folded into the constants)
    const int8_t *sq_wfrag;
    const float *sq_const;
    float sq_inv[2], sq_zoff[2], sq_hzoff[2];
};

// Parameter block of the W8A8 LE convolutions outside conv32p (conv_q8.hip).
struct ConvQ8Params {
    const void *src;       // NHWC: f16 (quantised while it is staged) or int8 codes q - 128 of THIS layer's quantiser
    int src_i8;
    int Cin;               // 32 or 64
    int src_stride;        // elements per pixel of src (>= Cin: a channel slice of a wider tensor)
    int Hi, Wi, Ho, Wo, ks, stride;
    const int8_t *wpk8;    // [ks*ks][CoutPad][Cin], natural channel order
    const float *scale;    // [CoutPad]: x_scale * w_scale[n]
    const float *shift;    // [16 border classes][CoutPad]: bias + w_scale[n] * (128 x_scale + x_zero) * sum of in-image taps
    int CoutPad, Cout, act;
    float q_inv, q_zoff;   // input quantiser: u8 code = clamp(rint(x * q_inv + q_zoff), 0, 255)
    void *dst;             // NHWC f16, or int8 codes of the READING layer's quantiser (oq_inv, oq_zoff)
    int dst_i8;
    float oq_inv, oq_zoff;
    int dstC;
};

// 1..3 W8A8 layers (3x3, stride 2, 64 -> 64) reading one f16 NHWC tensor through their own quantisers (conv_q8.hip)
struct ConvQ8Group {
    const int8_t *wpk8;    // [9][64][64]
    const float *scale, *shift;
    float q_inv, q_zoff;
    void *dst;             // NHWC 64: f16, or int8 codes of the reading layer's quantiser
    int dst_i8;
    float oq_inv, oq_zoff;
    int act;
};
struct ConvQ8MultiParams {
    const f16 *src;
    int src_stride, Hi, Wi, Ho, Wo, ngroups;
    ConvQ8Group g[3];
};

// Activation fake-quantiser of a W8A8 layer as the reference's W8A8Conv2d.forward applies it to its input (hdrtvnet_torch.py:
// 351-358): q = clamp(rint(x * inv + zoff), 0, 255), x' = f16(q * scale + zero)  (le_rows.hip applies it in registers)
struct FqParam { float inv, zoff, scale, zero; };

// Parameter block of the row-streaming fused ResBlock_with_SFT (le_rows.hip): y = x + conv2(sft2(relu(conv1(sft1(x, c))), c))
struct RowsRbParams {
    const f16 *x;          // NHWC 32 [H][W]
    const f16 *cond;       // NHWC 16 [H][W]
    const f16 *w1, *w2;    // [9][32][32] (pack_conv, CIN_T = 32)
    const float *b1, *b2;  // [32] bias
    const f16 *sft1_wfrag, *sft2_wfrag;     // pack_sft fragments
    const float *sft1_bias, *sft2_bias;
    f16 *dst;              // NHWC 32 [H][W]
    char *trash;           // >= 8 KiB write-only scratch for masked-off lanes (keeps store counts exact)
    void *dump;            // diagnostic builds (make STAMP=1): per-phase cycle sums
    int H, W;
    int nstrips, rows_per_seg;              // set by the launcher
    // W8A8 layers as fake-quant on the fp16 kernel (weights = the dequantised int8 weights): bit 0 conv1's input, 1 conv2's,
    // 2 sft1's four convs, 3 sft2's.  fq_s*: cond -> scale branch, cond -> shift branch, scale hidden, shift hidden
    int fq;
    FqParam fq_c1, fq_c2, fq_s1[4], fq_s2[4];
};

// The same chains with EVERY layer W8A8 on int8 MFMA (le_rows_i8.hip): a conv is conv32s<.., i8>'s operand set (pack_conv32_i8:
// wpk8 [9][32][32] int8, byte 16 h + 4 qd + k of a row = input channel 8 qd + 4 h + k; scale [32]; shift [16 border classes][32];
// the input quantiser q_inv / q_zoff), an SFT layer its SQ set (pack_sft: three int8 A fragments, 192 dequantisation constants,
// the two condition quantisers and the two hidden-layer zero offsets)
struct RowsConvI8 { const int8_t *wpk8; const float *scale, *shift; float q_inv, q_zoff; };
struct RowsSftI8 { const int8_t *wfrag; const float *konst; float inv[2], zoff[2], hzoff[2]; };
struct RowsRbI8Params {
    const f16 *x, *cond;   // NHWC 32 / 16 [H][W]
    RowsConvI8 c1, c2;
    RowsSftI8 s1, s2;
    float slope1;          // act_slope of conv1's activation (ReLU: 0), a runtime value as in conv32s
    f16 *dst;
    char *trash;
    int H, W;
    int nstrips, rows_per_seg;
};

// ... the full-resolution tail with up_conv3 (wpk8 [9][128][32], PixelShuffle row order; scale [128]; shift [16][128]), SFT_layer2,
// HR_conv2 and conv_last all W8A8
struct RowsTailI8Params {
    const f16 *u, *fea0, *cond, *res_planar;
    f16 *dst_planar;
    RowsConvI8 up, hr, last;
    RowsSftI8 s;
    float slope_relu;      // act_slope(ACT_RELU), a runtime value as in conv32p / conv32s
    char *trash;
    int H, W;
    int nstrips, rows_per_seg;
};

// ... the full-resolution head with conv_first (cf: conv_c3_q8's operands -- two int8 A fragments per lane, K = (kernel row | kernel
// column, channel)), SFT_layer1, HR_conv1 and down_conv1 (dn: its weights in pack_conv32_i8's K order, "LE.down_conv1#rows8") all W8A8
struct RowsHeadI8Params {
    const f16 *img;        // f16 [3][H][W]
    const f16 *cond;       // NHWC 16 [H][W]
    f16 *fea0, *fea1;      // NHWC 32 [H][W] / [H/2][W/2]
    RowsConvI8 cf, hr, dn;
    RowsSftI8 s;
    float slope_relu;
    char *trash;
    int H, W;
    int nstrips, rows_per_seg;
};

// Parameter block of the row-streaming fused tail of the LE net (le_rows.hip):
// out = res + conv_last(relu(HR_conv2(sft(relu(shuffle(up_conv(u))) + skip, cond))))
struct RowsTailParams {
    const f16 *u;          // NHWC 32 [H/2][W/2]: the half-resolution trunk output
    const f16 *fea0;       // NHWC 32 [H][W]: the skip
    const f16 *cond;       // NHWC 16 [H][W]
    const f16 *res_planar; // f16 [3][H][W]: the long skip (agcm output)
    f16 *dst_planar;       // f16 [3][H][W]
    const f16 *w_up;       // [9][128][32], rows in PixelShuffle order (pack_conv ps_cps = 32)
    const float *b_up;     // [128]
    const f16 *sft_wfrag;
    const float *sft_bias;
    const f16 *w_hr, *w_last;               // [9][32][32]
    const float *b_hr, *b_last;             // [32]
    char *trash;           // >= 8 KiB write-only scratch for masked-off lanes
    void *dump;            // diagnostic builds (make STAMP=1)
    int H, W;
    int nstrips, rows_per_seg;              // set by the launcher
    // fake-quant (see RowsRbParams): bit 0 up_conv's input (u), 1 HR_conv2's (the modulated sum), 2 conv_last's, 3 the SFT layer's convs
    int fq;
    FqParam fq_u, fq_y, fq_z, fq_s[4];
};

// Parameter block of the row-streaming fused head of the LE net (le_rows.hip):
// fea0 = relu(HR_conv1(sft(relu(conv_first(img)), cond))), fea1 = relu(down_conv1(fea0))
struct RowsHeadParams {
    const f16 *img;        // f16 [3][H][W]
    const f16 *cond;       // NHWC 16 [H][W]
    const f16 *c3_wfrag;   // conv_first: pack_c3 fragments [3 kernel rows][64 lanes][8], bias in the K axis
    const f16 *sft_wfrag;
    const float *sft_bias;
    const f16 *w_hr, *w_down;               // [9][32][32]
    const float *b_hr, *b_down;             // [32]
    f16 *fea0;             // NHWC 32 [H][W]
    f16 *fea1;             // NHWC 32 [H/2][W/2]
    char *trash;
    void *dump;
    int H, W;
    int nstrips, rows_per_seg;
    // fake-quant (see RowsRbParams): bit 0 conv_first's input (the image), 1 HR_conv1's, 2 down_conv1's, 3 the SFT layer's convs
    int fq;
    FqParam fq_img, fq_y, fq_f, fq_s[4];
};

// precision="fp32" (fp32_ops.hip): one generic convolution over planar CHW fp32 tensors
struct F32ConvParams {
    const float *x0, *x1;            // input planes; x1 = the second half of a channel concat (torch.cat((a, b), 1)) or null
    int c0, c1;
    int Hi, Wi, Ho, Wo;
    const float *w;                  // [cout group][cin][tap][cot]
    const float *bias;               // padded to the group size, like every per-channel vector below
    const float *bn_s, *bn_t;        // BatchNorm2d(eval) as scale / shift, or null
    const float *gfm_s, *gfm_t;      // GFM modulation v*s + t + v, or null
    const float *res;                // residual [cout][Ho][Wo] added after the activation, or null
    float *y;
    int cout, cot, pad, act;         // act: 0 none, 1 ReLU, 2 LeakyReLU(slope)
    float slope;
    int ps;                          // store through PixelShuffle(2): y is [cout/4][2Ho][2Wo]
    int narrow_below;                // developer A/B: workgroups per CU below which a layer runs 8 channels per lane (0 = default)
    int no_mfma;                     // developer A/B (variant f32_mfma = 0): every layer on the vector-FMA kernel
};
struct F32GfmParams {
    const float *w[6], *b[6];        // cond_scale_{first,HR,last}, cond_shift_{first,HR,last}: Linear(6 -> n)
    int n[6];
    const float *fea;                // the classifier's 6-vector
    float *out;                      // [6][64]
};

// Letterbox (letterbox.hip): u8 BGR [sh][sw][3] -> u8 BGR [dh][dw][3], resized region [y0, y0+new_h) x [x0, x0+new_w)
enum { LB_COPY = 0, LB_AREA_INT = 1, LB_AREA_FRAC = 2, LB_CUBIC = 3 };
struct LetterboxParams {
    const uint8_t *src;
    uint8_t *dst;
    int sh, sw, dh, dw, new_w, new_h, x0, y0, mode;
    int ix, iy;                      // LB_AREA_INT: integer shrink factors
    // LB_AREA_FRAC: run starts [new+1], source indices and float weights per run entry.  LB_CUBIC: xsrc/ysrc = first
    // of four source taps per destination index, xc/yc = four 11-bit fixed-point coefficients per index.
    const int *xbeg, *ybeg, *xsrc, *ysrc;
    const float *xw, *yw;
    const int *xc, *yc;
};

// Objective metrics (metrics.hip): two unit-range images [3][H][W], per-workgroup partial sums {squared error, SSIM, dE-ITP}
struct MetricsParams {
    const void *a, *b;
    int is_f32, H, W;
    float peak_nits;
    double *partials;   // [workgroups][3]
};
