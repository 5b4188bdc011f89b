/*
 * hdrtv_mi355x.h -- C ABI of libhdrtv_mi355x.so: the MI355X-native (gfx950) SDR->HDR
 * per-frame inference path of HDRTVNet++ (AGCM -> LE -> HG) with its pre/post stages.
 *
 * The reference (DanHelmy/hdr-realtime-video-pipeline) has no FFI: its AMD backend is the
 * Python class HDRTVNetTorch (src/models/hdrtvnet_torch.py:1513).  Each entry point below
 * names the reference method/lines it replaces; the Python mirror that binds them is
 * hdr-realtime-video-pipeline_amd/hdrtv_mi355x/processor.py.
 *
 * Conventions: plain pointers and sizes only.  Every "dev" pointer is device memory
 * (hipMalloc / torch CUDA tensor .data_ptr()); `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  All calls are stream-ordered and never synchronise, except
 * hdrtv_create / hdrtv_reserve / hdrtv_destroy and the hdrtv_ring_* host-side calls.
 * Return 0 on success, a negative HDRTV_E* code otherwise; hdrtv_last_error() describes it.
 * Not re-entrant per context (like the reference: one processor per worker thread).
 */
#ifndef HDRTV_MI355X_H
#define HDRTV_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HDRTV_OK 0
#define HDRTV_EINVAL (-1)   /* bad argument / shape                      */
#define HDRTV_EWEIGHTS (-2) /* weight pack malformed or tensor missing   */
#define HDRTV_EHIP (-3)     /* a HIP runtime call failed                 */
#define HDRTV_ENOMEM (-4)
#define HDRTV_ESTATE (-5)   /* call order (e.g. infer before reserve)    */

typedef struct hdrtv_ctx hdrtv_ctx;

/* Output element type of hdrtv_infer's `out` and input type of the post kernels. */
#define HDRTV_F16 0
#define HDRTV_F32 1

/* Library / build info: "hdrtv_mi355x <ver> gfx950 ...".  Never NULL. */
const char *hdrtv_version(void);

/* Replaces HDRTVNetTorch.__init__/_load_model (hdrtvnet_torch.py:1532-1673, 2044-2169).
 * hr_pack: HDRW1 weight pack (hdrtv_mi355x/weights.py) holding the 264 AGCM+LE tensors of
 * HR.pt under their state_dict names.  hg_pack: HDRW1 pack of Hallucination_Generator's
 * state_dict (conv*.0.*, conv*.1.* BatchNorm, Up_conv*.0.*, conv6..conv10, conv_last) or
 * NULL/0 for the no-HG model (reference: use_hg=False).  Weights are repacked on the host
 * into MFMA operand layouts and uploaded to `device_id`. */
int hdrtv_create(const void *hr_pack, size_t hr_bytes, const void *hg_pack, size_t hg_bytes,
                 int device_id, hdrtv_ctx **out);

/* The same with the compute precision of HDRTVNetTorch(precision=...) (hdrtvnet_torch.py:1694-1712).
 * HDRTV_PREC_F16 (= hdrtv_create): f16 storage / fp32 accumulate on the MFMA kernels; tensors at the boundary are f16
 * (the HG head's output f32).  HDRTV_PREC_F32: the reference's fp32 preset -- the same graph on planar fp32 tensors with
 * vector-FMA kernels (csrc/fp32_ops.hip; no matrix path exists for fp32 on gfx950, so roughly 1/30 of the f16 frame rate):
 * hdrtv_preprocess writes f32 `rgb` / `cond`, hdrtv_infer takes them and writes f32 `out` / `agcm_out` (out_dtype must
 * be HDRTV_F32), the post kernels take dtype HDRTV_F32.  An INT8 checkpoint is refused (HDRTV_EWEIGHTS). */
#define HDRTV_PREC_F16 0
#define HDRTV_PREC_F32 1
int hdrtv_create_ex(const void *hr_pack, size_t hr_bytes, const void *hg_pack, size_t hg_bytes,
                    int device_id, int precision, hdrtv_ctx **out);
int hdrtv_destroy(hdrtv_ctx *ctx);

/* 1 if the context was created with HG weights. */
int hdrtv_has_hg(const hdrtv_ctx *ctx);

/* How hdrtv_preprocess derives the 0.25x condition map (hdrtvnet_torch.py:2262-2294): 0 = antialiased bicubic (default),
 * 1 = bilinear, the reference's fast_condition_resize=True / HDRTVNET_FAST_COND_RESIZE, 2 = zeros, HDRTVNET_ZERO_COND. */
int hdrtv_set_cond_mode(hdrtv_ctx *ctx, int mode);

/* HG_Composite(mask_r=0.75) (HG_Composite_arch.py:21, 78-84): the highlight mask is max_c(base) > r + 0.1 * (1 - r).
 * The reference fixes r at construction; 0 <= r < 1.  Takes effect at the next hdrtv_infer. */
int hdrtv_set_hg_mask_r(hdrtv_ctx *ctx, float r);

/* Replaces HDRTVNetTorch._ensure_buffers (hdrtvnet_torch.py:2198-2233): (re)allocates the
 * internal activation workspace for H x W frames.  No-op when the size is unchanged.
 * Synchronises the device when it reallocates. */
int hdrtv_reserve(hdrtv_ctx *ctx, int H, int W);

/* Replaces the device half of HDRTVNetTorch.preprocess (hdrtvnet_torch.py:2255-2294).
 * dev_bgr_hwc : u8  [H][W][3] BGR (the frame after the pinned H2D copy)
 * dev_rgb_chw : f16 [3][H][W]     RGB, fp16(float(u8) * fp32(1/255))
 * dev_cond    : f16 [3][H/4][W/4] 0.25x antialiased bicubic (a=-0.5), fp32 accumulate */
int hdrtv_preprocess(hdrtv_ctx *ctx, void *stream, const uint8_t *dev_bgr_hwc, int H, int W,
                     void *dev_rgb_chw, void *dev_cond);

/* Replaces HDRTVNetTorch.infer -> model((tensor, cond)) (hdrtvnet_torch.py:2301-2346;
 * Ensemble_AGCM_LE_arch.py:889-897, HG_Composite_arch.py:86-107).
 * dev_rgb_chw, dev_cond: as produced by hdrtv_preprocess.
 * dev_out      : [3][H][W] planar RGB; out_dtype HDRTV_F16 without HG.  With HG the reference's
 *                result is fp32 (the fp32 highlight mask promotes it): pass HDRTV_F32, or
 *                HDRTV_F16 to have it rounded once at the end.
 * dev_agcm_out : f16 [3][H][W], second element of the reference's result tuple; may be NULL. */
int hdrtv_infer(hdrtv_ctx *ctx, void *stream, const void *dev_rgb_chw, const void *dev_cond, int H,
                int W, void *dev_out, int out_dtype, void *dev_agcm_out);

/* Frames in flight (no reference counterpart: the reference's worker processes one frame at a time and hides its copies
 * behind streams, gui_pipeline_worker_feeders.py:125-249).  A context holds `lanes` activation workspaces (1 or 2, default
 * 1; weights are shared); hdrtv_infer_lane(ctx, l, stream_l, ...) is hdrtv_infer on workspace l, so calls with different
 * lanes on different streams may overlap on the device: the tail of one frame's kernel fills with the next frame's
 * workgroups (+3 .. 4.5 % frames/s at 3840x2160 with two lanes, +11 % at 1920x1080; a frame's own latency doubles).
 * Results do not depend on the lane (tests/test_gpu_lanes.py).  HDRTV_EINVAL for more than two lanes (three kernels running
 * at once is where rare wrong tiles were seen) and for two on an fp32 context.  hdrtv_set_lanes drops the
 * reservation when the count changes (call hdrtv_reserve again; it synchronises the device); hdrtv_infer is lane 0.  Calls on one context are made by one host thread at a time,
 * as before: lanes make the DEVICE work concurrent, not the entry points re-entrant.  hdrtv_get_tap addresses lane 0. */
int hdrtv_set_lanes(hdrtv_ctx *ctx, int lanes);
int hdrtv_get_lanes(const hdrtv_ctx *ctx);
int hdrtv_infer_lane(hdrtv_ctx *ctx, int lane, void *stream, const void *dev_rgb_chw, const void *dev_cond, int H,
                     int W, void *dev_out, int out_dtype, void *dev_agcm_out);

/* Replaces HDRTVNetTorch.postprocess's device half (hdrtvnet_torch.py:2357-2361):
 * trunc(clamp(x,0,1)*255 + 0.5) in the tensor's dtype semantics, RGB planar -> u8 [H][W][3] BGR. */
int hdrtv_post_u8(hdrtv_ctx *ctx, void *stream, const void *dev_out, int dtype, int H, int W,
                  uint8_t *dev_bgr_hwc);

/* Replaces _tensor_to_rgb48_bytes' GPU branch (gui_pipeline_worker_feeders.py:223-227):
 * fp32(x) -> clamp(0,1) -> *65535 -> +0.5 (two fp32 roundings) -> trunc u16, planar ->
 * [H][W][3] RGB little-endian (mpv "rgb48le").  dst is device memory (e.g. a ring slot's dev_ptr;
 * hdrtv_ring_commit then moves it to the slot's pinned host buffer as the reference's host.copy_ does, :228). */
int hdrtv_post_rgb48(hdrtv_ctx *ctx, void *stream, const void *dev_out, int dtype, int H, int W,
                     uint16_t *dst);

/* North-star display variant with no counterpart on the reference's playback path
 * (SURVEY.md 8a-14/15): treats x as linear-light BT.709 with 1.0 = peak_nits, applies the
 * BT.709->BT.2020 matrix (ITU-R BT.2087), the ST.2084 PQ OETF
 * (gui_objective_metrics.py:486-491 constants) and the u16 quantiser, same output layout. */
int hdrtv_post_pq_rgb48(hdrtv_ctx *ctx, void *stream, const void *dev_out, int dtype, int H, int W,
                        float peak_nits, uint16_t *dst);

/* The host step in front of preprocess, on the device: _letterbox_bgr
 * (src/gui_scaling.py:228-244), i.e. cv2.resize preserving the aspect ratio -- INTER_AREA when shrinking,
 * INTER_CUBIC when enlarging -- centred on a black [dh][dw] canvas.  src / dst are device u8 BGR HWC.
 * Parity with cv2 itself is UNPINNED (OpenCV is not part of the reference tree); the arithmetic is the
 * restatement in oracle/letterbox_oracle.py, which the GPU tests hold this entry point to bit for bit. */
int hdrtv_letterbox_u8(hdrtv_ctx *ctx, void *stream, const uint8_t *dev_src_bgr, int sh, int sw,
                       uint8_t *dev_dst_bgr, int dh, int dw);

/* Objective metrics of the reference's metrics dict between two device images [3][H][W]
 * (R, G, B planes, unit range, f16 or f32): out3 = { PSNR dB, SSIM, dE-ITP } as _psnr_bgr / _ssim_bgr /
 * _delta_e_itp_bgr compute them (src/gui_objective_metrics.py:438-528; peak_nits = HDRTVNET_OBJECTIVE_HDR_PEAK_NITS,
 * 1000).  Synchronises `stream` (the three numbers come back to the host).  Parity UNPINNED: that module needs
 * cv2; the arithmetic is oracle/metrics_oracle.py's restatement. */
int hdrtv_metrics(hdrtv_ctx *ctx, void *stream, const void *dev_a, const void *dev_b, int dtype, int H, int W,
                  float peak_nits, double *out3);

/* ---- pinned host RGB48 ring: replaces _pinned_u16_host_ring / _acquire_pinned_u16_slot /
 * _PinnedMpvFrame (gui_pipeline_worker_feeders.py:38-70, 125-170).  `slots` in [2,8]
 * (HDRTVNET_FEEDER_GPU_RGB48_RING_FRAMES).  A slot cycles free -> acquired -> (kernel writes its device
 * buffer; commit = hipMemcpyAsync to the pinned host buffer + event) -> waited -> released. */
int hdrtv_ring_create(hdrtv_ctx *ctx, int slots, int H, int W);
/* Returns the slot index (>=0), its pinned host buffer and its device staging buffer, or HDRTV_ESTATE when no
 * slot frees up within timeout_ms (reference: 250 ms, feeders.py:166). */
int hdrtv_ring_acquire(hdrtv_ctx *ctx, int timeout_ms, uint16_t **host_ptr, uint16_t **dev_ptr);
/* After hdrtv_post_rgb48 into the slot's dev_ptr: enqueues the device -> pinned-host copy on `stream` and records
 * the slot's ready event behind it. */
int hdrtv_ring_commit(hdrtv_ctx *ctx, int slot, void *stream);
/* Blocks the calling host thread until the slot's contents are complete (wait_ready). */
int hdrtv_ring_wait(hdrtv_ctx *ctx, int slot);
/* Marks the slot free again (payload.release()). */
int hdrtv_ring_release(hdrtv_ctx *ctx, int slot);
int hdrtv_ring_destroy(hdrtv_ctx *ctx);

/* ---- introspection for parity tests and profiling (no reference counterpart) ----------
 * Looks up an internal activation by name after hdrtv_infer (e.g. "le.cond1", "hg.conv4_2").
 * layout: 0 = NHWC f16, 1 = planar CHW f16, 2 = planar CHW f32, 3 = f32 vector. */
int hdrtv_get_tap(hdrtv_ctx *ctx, const char *name, void **dev_ptr, int *C, int *H, int *W,
                  int *layout);
/* Number of kernel launches one hdrtv_infer issues at the reserved size, and the algorithmic
 * multiply-accumulates of Conv2d/Linear layers per frame. */
int hdrtv_infer_stats(hdrtv_ctx *ctx, int *launches, double *macs);

/* Per-launch timing of hdrtv_infer with HIP events on the launch stream (bench.py roofline).
 * While enabled, every hdrtv_infer records one event after each kernel launch; interval i is
 * launch i (plus its launch gap).  hdrtv_profile_get(ctx, -1, ...) returns the number of launches
 * recorded by the last hdrtv_infer; index i returns the layer name, the kernel (template
 * instance) name, its duration in ms, its algorithmic MACs and its algorithmic bytes
 * (activations in + out + residuals + weights, each counted once). */
int hdrtv_profile_enable(hdrtv_ctx *ctx, int on);
int hdrtv_profile_get(hdrtv_ctx *ctx, int i, const char **layer, const char **kernel, float *ms, double *macs,
                      double *bytes);

/* Developer / test switch with no reference counterpart: selects which of several equivalent kernels or schedules a
 * layer runs on (e.g. "le_rows": 1 = the fused row-streaming LE kernels, 0 = one launch per layer).  The table is
 * filled at hdrtv_create (defaults, then the creating process's HDRTV_VARIANTS="name=value,..."); the launch path never
 * reads the environment.  Takes effect at the next hdrtv_infer; HDRTV_EINVAL for an unknown name. */
int hdrtv_set_variant(hdrtv_ctx *ctx, const char *name, int value);
int hdrtv_get_variant(hdrtv_ctx *ctx, const char *name, int *value);

const char *hdrtv_last_error(const hdrtv_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* HDRTV_MI355X_H */
