// launchers.h -- host-callable launch wrappers defined in the .hip kernel files.
#pragma once
#include "common.h"

// One-time per-kernel setup (hipFuncSetAttribute) is a property of (function, device), not of the process: a process
// that holds contexts on several devices must repeat it on each.  Races are benign (the setup is idempotent).
struct DevOnce {
    unsigned char set[64] = {};
    static int cur() { int d = 0; return hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64 ? d : 0; }
    bool need() const { return !set[cur()]; }
    void done() { set[cur()] = 1; }
};

hipError_t conv_igemm_launch(ConvParams p, int cin_t, int bn, int ks, int stride, hipStream_t stream);
hipError_t conv_glds1_launch(ConvParams p, hipStream_t stream, int n_cu = 0, bool old_form = false);   // n_cu >= 8: the persistent form
hipError_t conv_pglds_launch(ConvParams p, int n_cu, hipStream_t stream);
hipError_t conv_prw_launch(ConvParams p, int th, int n_cu, hipStream_t stream);   // Cout % 256 == 0, modes NHWC / PS / POOL; th = 16 | 8
hipError_t conv_pglds_i8_launch(ConvI8Params p, int n_cu, hipStream_t stream);
hipError_t conv_prw_i8_launch(ConvI8Params p, int th, int n_cu, hipStream_t stream);   // Cin % 128 == 0, Cout % 256 == 0, int8 out
hipError_t conv1x1_i8_launch(ConvI8Params p, hipStream_t stream);
hipError_t conv32p_launch(Conv32Params p, int n_cu, hipStream_t stream, int nw = 0);
hipError_t conv32s_launch(Conv32Params p, int n_cu, hipStream_t stream, bool nosplit = false);   // the single-pass (CoutPad == 32) layers
hipError_t le_rb_rows_launch(RowsRbParams p, int n_cu, hipStream_t stream);   // fused ResBlock_with_SFT, row-streaming (le_rows.hip)
hipError_t le_rb_rows_i8_launch(RowsRbI8Params p, int n_cu, hipStream_t stream);   // ... every layer W8A8 on int8 MFMA (le_rows_i8.hip)
hipError_t le_tail_rows_i8_launch(RowsTailI8Params p, int n_cu, hipStream_t stream);
hipError_t le_head_rows_i8_launch(RowsHeadI8Params p, int n_cu, hipStream_t stream);
hipError_t le_tail_rows_launch(RowsTailParams p, int n_cu, hipStream_t stream);   // up_conv3 .. conv_last in one launch (le_rows.hip)
hipError_t le_head_rows_launch(RowsHeadParams p, int n_cu, hipStream_t stream);   // conv_first .. down_conv1 in one launch (le_rows.hip)
hipError_t conv_t16_launch(ConvParams p, hipStream_t stream, int n_cu);
hipError_t conv_q8_launch(ConvQ8Params p, hipStream_t stream, int n_cu);
hipError_t conv_q8_multi_launch(ConvQ8MultiParams p, int n_cu, hipStream_t stream);
hipError_t conv3x3s2_preg_launch(ConvParams p, int n_cu, hipStream_t stream);

hipError_t pre_unpack_launch(const uint8_t *bgr, f16 *out, int H, int W, hipStream_t s);
hipError_t cond_resize_launch(const f16 *in, f16 *out, int H, int W, int Ho, int Wo, const float *wx, const int *xmn,
                              const int *xns, const float *wy, const int *ymn, const int *yns, hipStream_t s);
hipError_t pre_fused_launch(const uint8_t *bgr, f16 *out, f16 *cond, int H, int W, int Ho, int Wo, const float *wx, const int *xmn,
                            const int *xns, const float *wy, const int *ymn, const int *yns, int mode, hipStream_t s);
hipError_t post_u8_launch(const void *in, int is_f32, int H, int W, uint8_t *bgr, hipStream_t s);
hipError_t post_rgb48_launch(const void *in, int is_f32, int H, int W, uint16_t *rgb, int pq, float peak, hipStream_t s,
                             const float *pq_bnd = nullptr);

// activation fake-quantiser of a W8A8 layer evaluated in fp32 (classifier convs, Linear heads): on = 0 passes x through
struct FakeQ { int on; float inv, zoff, scale, zero; };
hipError_t cls_block_launch(const void *in, int in_f16, int Ci, int Hi, int Wi, const float *nmean, const float *nrstd,
                            const float *ngamma, const float *nbeta, const float *Wt, const float *bias, int Co, float *out,
                            int Ho, int Wo, float *part, hipStream_t s, const FakeQ *qin = nullptr, const FakeQ *qstat = nullptr,
                            int *nblk = nullptr);
hipError_t cls_stats_launch(const float *part, int C, int nblk, int n, float eps, float *mean, float *rstd, hipStream_t s);
struct AgcmFoldArgs {
    const float *mean5;
    const float *w20, *b20;
    const float *ws[3], *bs[3], *wt[3], *bt[3];
    const float *w1, *b1, *w2, *b2, *w3, *b3;
};
hipError_t agcm_fold_launch(const AgcmFoldArgs &a, f16 *frags, float *biasbuf, hipStream_t s);
hipError_t agcm_mlp_launch(const f16 *in, f16 *out, size_t npix, const f16 *frags, const float *biasbuf, hipStream_t s);

hipError_t conv_c3_launch(const f16 *in, int H, int W, const f16 *wfrag, const float *scale, const float *shift, int cout,
                          int act, f16 *out, f16 *out_pool, int n_cu, hipStream_t s, float pool_q_inv = 0.f, float pool_q_zero = 0.f,
                          const f16 *w2frag = nullptr, float *part2 = nullptr);   // HG.conv1: + conv10's second half per pixel (f32 [H][W][4])
// LE.conv_first as a W8A8 layer: int8 A fragments [2][64 lanes][16 B] (K = (ky | kx4, c4)), scale[32], shift[16 border classes][32]
hipError_t conv_c3_q8_launch(const f16 *in, int H, int W, const int8_t *wq, const float *scale, const float *shift, float q_inv,
                             float q_zoff, int act, f16 *out, int n_cu, hipStream_t s);
// pool_q_inv > 0: out_pool holds int8 codes clamp(rint(v * pool_q_inv + pool_q_zero), -128, 127), COUT bytes per pixel
// last layer of a fused chain as a W8A8 layer: int8 weight fragments [2][64 lanes][16 B], ss = scale[32] | shift[32]
struct QLastArgs {
    const int8_t *wq;
    const float *ss;
    float q_inv, q_zoff;
};
hipError_t le_cond_trunk_launch(const f16 *img, int H, int W, const f16 *wfrag, const float *bias, f16 *cond, f16 *cond1,
                                int n_cu, hipStream_t s, const QLastArgs *q6 = nullptr);
hipError_t cond_tail_launch(const f16 *x, int x_stride, size_t npx, const f16 *wfrag, const float *bias, f16 *out, int n_cu,
                            hipStream_t s, const QLastArgs *q2 = nullptr);
// ---- fully quantised (W8A8) per-pixel chains (le_chain_q8.hip); layouts documented there
struct TrunkQ8Args {
    const int8_t *wfrag;           // [20][64 lanes][16 B]
    const float *consts;           // [1664]
    float q1_inv, q1_zoff, q4_inv;
    float zoff[5];
};
hipError_t le_cond_trunk_q8_launch(const f16 *img, int H, int W, const TrunkQ8Args &a, f16 *cond, f16 *cond1, int n_cu, hipStream_t s);
hipError_t cond_tail_q8_launch(const int8_t *x, size_t npx, const int8_t *wfrag, const float *consts, float zoff2, f16 *out, int n_cu,
                               hipStream_t s);
hipError_t agcm_mlp_q8_launch(const f16 *in, f16 *out, size_t npix, const int8_t *wfrag, const float *consts, float q1_inv, float q1_zoff,
                              float zoff2, float zoff3, hipStream_t s);
hipError_t planar3_to_q8_launch(const f16 *in, size_t npix, float inv, float zoff, int8_t *out, hipStream_t s);
struct AgcmFoldQ8Args {
    FakeQ q20, qlin[6];            // model.20; cond_scale_{first,HR,last}, cond_shift_{first,HR,last}
    const float *P, *Q;            // [3][64] each: x_scale * w_scale[m]; w_scale[m] * (128 x_scale + x_zero) * sum(w) + bias[m]
    float inv2, inv3;              // 1 / x_scale of HRconv and conv_last
    float *consts;                 // out: 320 per-frame constants of agcm_mlp_q8
};
hipError_t agcm_fold_q8_launch(const AgcmFoldArgs &a, const AgcmFoldQ8Args &q, float *biasbuf, hipStream_t s);
hipError_t hg_prep_launch(const f16 *base, int H, int W, int Hp, int Wp, f16 *img_pad, uint8_t *mask, float r, float thresh,
                          hipStream_t s);
struct HgFinalFusedArgs {
    const f16 *img;
    const uint8_t *mask;
    const float *part;
    const f16 *wfrag;
    const float *scale, *shift, *b10, *wl, *bl;
    void *out;
    int out_f32, H, W, Hp, Wp;
};
hipError_t hg_final_fused_launch(const HgFinalFusedArgs &a, int n_cu, hipStream_t s);
// the same tail when conv_c3_launch(..., w2frag, part2) has left conv10's second half per pixel: a per-pixel kernel (a.wfrag, scale, shift unused)
hipError_t hg_final_light_launch(const HgFinalFusedArgs &a, const float *part2, hipStream_t s);
hipError_t letterbox_launch(const LetterboxParams &p, hipStream_t s);
hipError_t metrics_launch(const MetricsParams &p, hipStream_t s);
int metrics_blocks(int H, int W);

// precision="fp32" (fp32_ops.hip)
hipError_t conv_f32_launch(const F32ConvParams &p, int ks, int stride, int n_cu, hipStream_t s);
bool conv_f32_on_mfma(const F32ConvParams &p, int ks, int n_cu);      // whether conv_f32_launch runs a stride-1 layer on the fp32 matrix pipe
hipError_t ew_f32_launch(int op, const float *a, const float *b, const float *c, float *y, size_t n, hipStream_t s);
hipError_t avgpool3s2_leaky_f32_launch(const float *x, float *y, int C, int H, int W, float slope, hipStream_t s);
hipError_t instnorm_f32_launch(float *x, const float *gamma, const float *beta, int C, int n, float eps, hipStream_t s);
hipError_t plane_mean_f32_launch(const float *x, float *y, int C, int n, hipStream_t s);
hipError_t gfm_heads_f32_launch(const F32GfmParams &p, hipStream_t s);
hipError_t maxpool2_f32_launch(const float *x, float *y, int C, int H, int W, hipStream_t s);
hipError_t window_f32_launch(const float *x, float *y, int C, int H, int W, int Ho, int Wo, int mode, hipStream_t s);
hipError_t hg_mask_f32_launch(const float *base, float *mask, size_t npix, float r, hipStream_t s);
hipError_t hg_blend_f32_launch(const float *t, const float *img, const float *mask, float *out, int H, int W, int Hp, int Wp,
                               hipStream_t s);
hipError_t pre_f32_launch(const uint8_t *bgr, float *rgb, float *cond, int H, int W, int Ho, int Wo, const float *wx,
                          const int *xmn, const int *xns, const float *wy, const int *ymn, const int *yns, int mode, hipStream_t s);
