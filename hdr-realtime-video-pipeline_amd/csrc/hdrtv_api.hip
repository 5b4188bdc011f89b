// hdrtv_api.hip -- the exported C ABI of libhdrtv_mi355x.so (include/hdrtv_mi355x.h): argument checks, the context's life cycle,
// the RGB48 ring, letterbox / metrics / PQ tables.  Packing, workspace and launch sequencing are their own translation units
// (api.h).  No kernel lives in this file.
#include "api.h"

using namespace hdrtv_host;
// =========================================================================== exported C ABI
extern "C" {

#ifndef HDRTV_BUILD_ID
#define HDRTV_BUILD_ID "unstamped"
#endif
// "... build <id>": <id> = the first 12 hex digits of the SHA-1 over the library's sources (csrc/Makefile)
const char *hdrtv_version(void) { return "hdrtv_mi355x 0.1 gfx950 (MFMA f16 implicit-GEMM, hand-written HIP) build " HDRTV_BUILD_ID; }

int hdrtv_create(const void *hr_pack, size_t hr_bytes, const void *hg_pack, size_t hg_bytes, int device_id, hdrtv_ctx **out)
{
    return hdrtv_create_ex(hr_pack, hr_bytes, hg_pack, hg_bytes, device_id, HDRTV_PREC_F16, out);
}

int hdrtv_create_ex(const void *hr_pack, size_t hr_bytes, const void *hg_pack, size_t hg_bytes, int device_id, int precision,
                    hdrtv_ctx **out)
{
    if (!out) return HDRTV_EINVAL;
    *out = nullptr;
    hdrtv_ctx *c = new hdrtv_ctx();
    *out = c;   // returned even on failure so hdrtv_last_error() can be read; caller still destroys it
    c->device = device_id;
    if (precision != HDRTV_PREC_F16 && precision != HDRTV_PREC_F32) return fail(c, HDRTV_EINVAL, "bad precision %d", precision);
    c->fp32 = precision == HDRTV_PREC_F32;
    if (!hr_pack || hr_bytes == 0) return fail(c, HDRTV_EINVAL, "hr_pack is required");
    Pack hr, hg;
    if (!hr.parse(hr_pack, hr_bytes, c->err)) return HDRTV_EWEIGHTS;
    c->has_hg = hg_pack != nullptr && hg_bytes > 0;
    if (c->has_hg && !hg.parse(hg_pack, hg_bytes, c->err)) return HDRTV_EWEIGHTS;
    if (c->fp32 ? !build_weights_f32(c, hr, c->has_hg ? &hg : nullptr) : !build_weights(c, hr, c->has_hg ? &hg : nullptr))
        return HDRTV_EWEIGHTS;
    int ndev = 0;
    HIPCHK(c, hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) return fail(c, HDRTV_EINVAL, "device %d not available (%d devices)", device_id, ndev);
    HIPCHK(c, hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(c, HDRTV_EINVAL, "device %d is %s; this library is built for gfx950 only", device_id, prop.gcnArchName);
    c->dev_ncu = c->n_cu = prop.multiProcessorCount;
    // test switch: a huge value gives every persistent kernel one tile per workgroup (tests/test_gpu_parity.py
    // compares that schedule bit for bit with the real one)
    variants_init(c);
    if (c->var.at("force_ncu") > 0) c->n_cu = c->var.at("force_ncu");
    if (hipMalloc((void **)&c->wts.dev, c->wts.size + 256) != hipSuccess) {
        c->wts.dev = nullptr;
        return fail(c, HDRTV_ENOMEM, "weight allocation failed");
    }
    HIPCHK(c, hipMemcpy(c->wts.dev, c->wts.host.data(), c->wts.host.size(), hipMemcpyHostToDevice));
    c->wts.host.clear();
    c->wts.host.shrink_to_fit();
    return HDRTV_OK;
}

int hdrtv_destroy(hdrtv_ctx *c)
{
    if (!c) return HDRTV_OK;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    hdrtv_ring_destroy(c);
    for (hipEvent_t ev : c->prof_ev) (void)hipEventDestroy(ev);
    if (c->pq_bnd) (void)hipFree(c->pq_bnd);
    if (c->lb_dev) (void)hipFree(c->lb_dev);
    if (c->mt_dev) (void)hipFree(c->mt_dev);
    free_workspaces(c);
    if (c->wts.dev) (void)hipFree(c->wts.dev);
    delete c;
    return HDRTV_OK;
}

int hdrtv_has_hg(const hdrtv_ctx *c) { return c && c->has_hg ? 1 : 0; }

int hdrtv_set_cond_mode(hdrtv_ctx *c, int mode)
{
    if (!c) return HDRTV_EINVAL;
    if (mode < 0 || mode > 2) return fail(c, HDRTV_EINVAL, "cond mode must be 0 (bicubic-aa), 1 (bilinear) or 2 (zero)");
    c->cond_mode = mode;
    return HDRTV_OK;
}

int hdrtv_set_variant(hdrtv_ctx *c, const char *name, int value)
{
    if (!c || !name) return HDRTV_EINVAL;
    auto it = c->var.find(name);
    if (it == c->var.end()) return fail(c, HDRTV_EINVAL, "unknown variant %s", name);
    if (!variant_allowed(name, value)) return fail(c, HDRTV_EINVAL, "variant %s = %d needs the A/B library (make AB=1)", name, value);
    const std::string n(name);
    // the row kernels take any segment height >= 1 (tests/test_gpu_le_rows.py: test_short_segments_*); 0 or less is a typo
    if (n == "le_rows_min" && (value < 1 || value > 4096)) return fail(c, HDRTV_EINVAL, "variant le_rows_min must be 1 .. 4096");
    if (n == "force_ncu") {                 // sizes grids at launch time only (no workspace depends on it): takes effect at once
        if (value < 0) return fail(c, HDRTV_EINVAL, "variant force_ncu must be >= 0");
        c->n_cu = value > 0 ? value : c->dev_ncu;
    }
    it->second = value;
    return HDRTV_OK;
}

int hdrtv_get_variant(hdrtv_ctx *c, const char *name, int *value)
{
    if (!c || !name || !value) return HDRTV_EINVAL;
    auto it = c->var.find(name);
    if (it == c->var.end()) return fail(c, HDRTV_EINVAL, "unknown variant %s", name);
    *value = it->second;
    return HDRTV_OK;
}

int hdrtv_set_hg_mask_r(hdrtv_ctx *c, float r)
{
    if (!c) return HDRTV_EINVAL;
    if (!(r >= 0.f && r < 1.f)) return fail(c, HDRTV_EINVAL, "mask_r must be in [0, 1)");
    c->mask_r = r;
    return HDRTV_OK;
}

int hdrtv_reserve(hdrtv_ctx *c, int H, int W)
{
    if (!c) return HDRTV_EINVAL;
    if (!c->wts.dev) return fail(c, HDRTV_ESTATE, "context not initialised");
    return do_reserve(c, H, W);
}

int hdrtv_preprocess(hdrtv_ctx *c, void *stream, const uint8_t *bgr, int H, int W, void *rgb, void *cond)
{
    if (!c || !bgr || !rgb || !cond) return fail(c, HDRTV_EINVAL, "null argument");
    if (c->H != H || c->W != W || !c->ws.dev) return fail(c, HDRTV_ESTATE, "call hdrtv_reserve(%d,%d) first", H, W);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    Seq q{c, s};
    const Shapes sh = shapes_for(H, W);
    if (c->fp32) {                       // precision="fp32": rgb / cond are fp32 tensors
        q.chk(pre_f32_launch(bgr, (float *)rgb, (float *)cond, H, W, sh.h4, sh.w4, wsp<float>(c, "aa.wx"), wsp<int>(c, "aa.xmn"),
                             wsp<int>(c, "aa.xns"), wsp<float>(c, "aa.wy"), wsp<int>(c, "aa.ymn"), wsp<int>(c, "aa.yns"),
                             c->cond_mode, s), "pre_f32");
        return q.rc;
    }
    const bool split = c->var.at("pre_split") != 0;                       // developer A/B: the two-kernel form
    if (H / 4 >= 1 && W / 4 >= 1 && !(split && c->cond_mode == 0)) {
        q.chk(pre_fused_launch(bgr, (f16 *)rgb, (f16 *)cond, H, W, sh.h4, sh.w4, wsp<float>(c, "aa.wx"), wsp<int>(c, "aa.xmn"),
                               wsp<int>(c, "aa.xns"), wsp<float>(c, "aa.wy"), wsp<int>(c, "aa.ymn"), wsp<int>(c, "aa.yns"), c->cond_mode, s),
              "pre_fused");
        return q.rc;
    }
    q.chk(pre_unpack_launch(bgr, (f16 *)rgb, H, W, s), "pre_unpack");
    q.chk(cond_resize_launch((const f16 *)rgb, (f16 *)cond, H, W, sh.h4, sh.w4, wsp<float>(c, "aa.wx"), wsp<int>(c, "aa.xmn"),
                             wsp<int>(c, "aa.xns"), wsp<float>(c, "aa.wy"), wsp<int>(c, "aa.ymn"), wsp<int>(c, "aa.yns"), s),
          "cond_resize");
    return q.rc;
}

int hdrtv_set_lanes(hdrtv_ctx *c, int lanes)
{
    if (!c) return HDRTV_EINVAL;
    // One or two.  Two is where the gain is (three and four frames in flight were slower than two), and three was where the trouble
    // was: with THREE kernels running at once, about one int8 frame in 500 had a few tiles of hg.conv2 wrong (its workgroups' waves
    // 4 .. 7, several workgroups at once; 13 of 4500 lane-frames with eight hardware queues, 0 of 3000 with GPU_MAX_HW_QUEUES=2, 0 of
    // 4000 with two lanes: tools/dbg/lane_stress2.py, NOTEBOOK.md round 5 section 8), and the one fp16 frame ever seen disturbed
    // (360 RGB48 values) ran with three lanes.  The fp32 preset stays on one lane: its vector kernels keep the packed-f32 arithmetic
    // (csrc/Makefile) that failed in pre_fused beside another stream's MFMA waves.
    // (HDRTV_LANES_ANY=1 in the environment of the calling process lifts both limits, up to 4: tools/dbg/lane_stress*.py)
    const char *any = getenv("HDRTV_LANES_ANY");
    const bool lifted = any && any[0] == '1';
    if (lanes < 1 || lanes > (lifted ? 4 : 2)) return fail(c, HDRTV_EINVAL, "lanes must be 1 or 2");
    if (lanes > 1 && c->fp32 && !lifted) return fail(c, HDRTV_EINVAL, "the fp32 preset runs one frame at a time (one lane)");
    if (lanes == c->lanes) return HDRTV_OK;
    if (c->ws.dev) {                      // the reservation goes with the old lane count
        HIPCHK(c, hipSetDevice(c->device));
        HIPCHK(c, hipDeviceSynchronize());
        free_workspaces(c);
    }
    c->lanes = lanes;
    return HDRTV_OK;
}

int hdrtv_get_lanes(const hdrtv_ctx *c) { return c ? c->lanes : 0; }

int hdrtv_infer(hdrtv_ctx *c, void *stream, const void *rgb, const void *cond, int H, int W, void *out, int out_dtype,
                void *agcm_out)
{
    return hdrtv_infer_lane(c, 0, stream, rgb, cond, H, W, out, out_dtype, agcm_out);
}

int hdrtv_infer_lane(hdrtv_ctx *c, int lane, void *stream, const void *rgb, const void *cond, int H, int W, void *out, int out_dtype,
                     void *agcm_out)
{
    if (!c || !rgb || !cond || !out) return fail(c, HDRTV_EINVAL, "null argument");
    if (c->H != H || c->W != W || !c->ws.dev) return fail(c, HDRTV_ESTATE, "call hdrtv_reserve(%d,%d) first", H, W);
    if (lane < 0 || lane >= c->lanes) return fail(c, HDRTV_EINVAL, "lane %d of %d (hdrtv_set_lanes)", lane, c->lanes);
    // every workspace tensor of this call resolves inside the lane's buffer; taps and hdrtv_preprocess's tables stay on lane 0
    struct LaneScope {
        hdrtv_ctx *c;
        LaneScope(hdrtv_ctx *c_, int l) : c(c_) { c->ws.dev = c->lane_ws[(size_t)l]; }
        ~LaneScope() { c->ws.dev = c->lane_ws[0]; }
    } lane_scope(c, lane);
    if (out_dtype != HDRTV_F16 && out_dtype != HDRTV_F32) return fail(c, HDRTV_EINVAL, "bad out_dtype");
    if (c->fp32 && out_dtype != HDRTV_F32) return fail(c, HDRTV_EINVAL, "an fp32 context takes and returns f32 tensors");
    if (!c->fp32 && !c->has_hg && out_dtype != HDRTV_F16) return fail(c, HDRTV_EINVAL, "the no-HG model returns f16");
    HIPCHK(c, hipSetDevice(c->device));
    Seq q{c, (hipStream_t)stream};
    c->launches = 0;
    c->macs = 0.0;
    c->prof.clear();
    q.mark();
    if (c->fp32) return run_f32(c, q, false, H, W, (const float *)rgb, (const float *)cond, (float *)out, (float *)agcm_out);
    f16 *agcm = agcm_out ? (f16 *)agcm_out : wsp<f16>(c, "agcm.out");
    if (run_agcm(c, q, (const f16 *)rgb, (const f16 *)cond, agcm) != HDRTV_OK) return q.rc;
    f16 *le_out = c->has_hg ? wsp<f16>(c, "le.out") : (f16 *)out;
    if (run_le(c, q, agcm, le_out) != HDRTV_OK) return q.rc;
    if (c->has_hg && run_hg(c, q, le_out, out, out_dtype == HDRTV_F32) != HDRTV_OK) return q.rc;
    return q.rc;
}

int hdrtv_post_u8(hdrtv_ctx *c, void *stream, const void *in, int dtype, int H, int W, uint8_t *bgr)
{
    if (!c || !in || !bgr || H <= 0 || W <= 0) return fail(c, HDRTV_EINVAL, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = post_u8_launch(in, dtype == HDRTV_F32, H, W, bgr, (hipStream_t)stream);
    return e == hipSuccess ? HDRTV_OK : fail(c, HDRTV_EHIP, "post_u8: %s", hipGetErrorString(e));
}

int hdrtv_post_rgb48(hdrtv_ctx *c, void *stream, const void *in, int dtype, int H, int W, uint16_t *dst)
{
    if (!c || !in || !dst || H <= 0 || W <= 0) return fail(c, HDRTV_EINVAL, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    hipError_t e = post_rgb48_launch(in, dtype == HDRTV_F32, H, W, dst, 0, 0.f, (hipStream_t)stream);
    return e == hipSuccess ? HDRTV_OK : fail(c, HDRTV_EHIP, "post_rgb48: %s", hipGetErrorString(e));
}

// ST.2084 OETF in double precision (constants gui_objective_metrics.py:63-67) -> u16 code, floor(pq * 65535 + 0.5): the
// definition the oracle's orc_pq_code states.  bnd[v] = the smallest fp32 y in [0, 1] with code(y) >= v.
static int pq_code64(float y)
{
    const double yp = std::pow((double)y, 0.1593017578125);
    const double v = std::pow((0.8359375 + 18.8515625 * yp) / (1.0 + 18.6875 * yp), 78.84375);
    const double q = std::floor(v * 65535.0 + 0.5);
    return (int)(q < 0.0 ? 0.0 : (q > 65535.0 ? 65535.0 : q));
}
static void pq_boundaries(std::vector<float> &bnd)
{
    bnd.assign(65536, 0.f);
    for (int v = 1; v <= 65535; ++v) {
        // analytic inverse (EOTF) as the first guess, then walk the fp32 grid to the exact boundary
        const double p = ((double)v - 0.5) / 65535.0, t = std::pow(p, 1.0 / 78.84375);
        const double num = std::max(t - 0.8359375, 0.0), den = 18.8515625 - 18.6875 * t;
        float f = (float)std::min(1.0, std::max(0.0, std::pow(num / den, 1.0 / 0.1593017578125)));
        while (f > 0.f && pq_code64(std::nextafterf(f, -1.f)) >= v) f = std::nextafterf(f, -1.f);
        while (f < 1.f && pq_code64(f) < v) f = std::nextafterf(f, 2.f);
        bnd[v] = f;
    }
    // first-guess table of the kernel (prepost.hip pq_code): exact codes at the fp32 values ((127 - 27) * 64 + i) << 17
    const int base = (127 - 27) << 6, n = 27 * 64 + 2;
    bnd.resize(65536 + n);
    for (int i = 0; i < n; ++i) {
        const uint32_t bits = (uint32_t)(base + i) << 17;
        float y;
        memcpy(&y, &bits, 4);
        bnd[65536 + i] = (float)pq_code64(y > 1.f ? 1.f : y);
    }
}

int hdrtv_post_pq_rgb48(hdrtv_ctx *c, void *stream, const void *in, int dtype, int H, int W, float peak_nits, uint16_t *dst)
{
    if (!c || !in || !dst || H <= 0 || W <= 0 || !(peak_nits > 0.f)) return fail(c, HDRTV_EINVAL, "bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->pq_bnd) {
        std::vector<float> bnd;
        pq_boundaries(bnd);
        if (hipMalloc((void **)&c->pq_bnd, bnd.size() * 4) != hipSuccess) { c->pq_bnd = nullptr; return fail(c, HDRTV_ENOMEM, "post_pq_rgb48: table allocation failed"); }
        HIPCHK(c, hipMemcpy(c->pq_bnd, bnd.data(), bnd.size() * 4, hipMemcpyHostToDevice));
    }
    hipError_t e = post_rgb48_launch(in, dtype == HDRTV_F32, H, W, dst, 1, peak_nits, (hipStream_t)stream, c->pq_bnd);
    return e == hipSuccess ? HDRTV_OK : fail(c, HDRTV_EHIP, "post_pq_rgb48: %s", hipGetErrorString(e));
}

// ------------------------------------------------------------------------------- letterbox
// Geometry and tables of _letterbox_bgr (gui_scaling.py:228-244) as restated in oracle/letterbox_oracle.py.
static bool letterbox_setup(hdrtv_ctx *c, int sh, int sw, int dh, int dw)
{
    if (c->lb_key[0] == sh && c->lb_key[1] == sw && c->lb_key[2] == dh && c->lb_key[3] == dw) return true;
    LetterboxParams &L = c->lb;
    memset(&L, 0, sizeof L);
    L.sh = sh; L.sw = sw; L.dh = dh; L.dw = dw;
    const double scale = std::min(dw / (double)std::max(sw, 1), dh / (double)std::max(sh, 1));
    L.new_w = std::max(1, (int)std::nearbyint(sw * scale));          // Python round(): half to even
    L.new_h = std::max(1, (int)std::nearbyint(sh * scale));
    L.x0 = (dw - L.new_w) / 2; L.y0 = (dh - L.new_h) / 2;
    std::vector<int> ints;
    std::vector<float> flts;
    size_t o_xbeg = 0, o_ybeg = 0, o_xsrc = 0, o_ysrc = 0, o_xc = 0, o_yc = 0, o_xw = 0, o_yw = 0;
    if (L.new_w == sw && L.new_h == sh) {
        L.mode = LB_COPY;
    } else if (scale < 1.0) {
        const double sx = sw / (double)L.new_w, sy = sh / (double)L.new_h;
        const int ix = (int)std::nearbyint(sx), iy = (int)std::nearbyint(sy);
        if (std::fabs(sx - ix) < DBL_EPSILON && std::fabs(sy - iy) < DBL_EPSILON) {
            L.mode = LB_AREA_INT; L.ix = ix; L.iy = iy;
        } else {
            L.mode = LB_AREA_FRAC;
            auto tab = [&](int ssize, int dsize, size_t &o_beg, size_t &o_src, size_t &o_w) {
                const double sc = ssize / (double)dsize;
                std::vector<int> beg(dsize + 1), src;
                std::vector<float> w;
                for (int dx = 0; dx < dsize; ++dx) {
                    beg[dx] = (int)src.size();
                    const double f1 = dx * sc, f2 = f1 + sc, cell = std::min(sc, ssize - f1);
                    int s1 = (int)std::ceil(f1), s2 = (int)std::floor(f2);
                    s2 = std::min(s2, ssize - 1); s1 = std::min(s1, s2);
                    if (s1 - f1 > 1e-3) { src.push_back(s1 - 1); w.push_back((float)((s1 - f1) / cell)); }
                    for (int q = s1; q < s2; ++q) { src.push_back(q); w.push_back((float)(1.0 / cell)); }
                    if (f2 - s2 > 1e-3) { src.push_back(s2); w.push_back((float)(std::min(std::min(f2 - s2, 1.0), cell) / cell)); }
                }
                beg[dsize] = (int)src.size();
                o_beg = ints.size(); ints.insert(ints.end(), beg.begin(), beg.end());
                o_src = ints.size(); ints.insert(ints.end(), src.begin(), src.end());
                o_w = flts.size(); flts.insert(flts.end(), w.begin(), w.end());
            };
            tab(sw, L.new_w, o_xbeg, o_xsrc, o_xw);
            tab(sh, L.new_h, o_ybeg, o_ysrc, o_yw);
        }
    } else {
        L.mode = LB_CUBIC;
        auto tab = [&](int ssize, int dsize, size_t &o_src, size_t &o_c) {
            const double sc = ssize / (double)dsize;
            std::vector<int> src(dsize), co(4 * (size_t)dsize);
            const float A = -0.75f;
            for (int d = 0; d < dsize; ++d) {
                float fx = (float)((d + 0.5) * sc - 0.5);
                const int s0 = (int)std::floor(fx);
                fx = fx - (float)s0;
                src[d] = s0 - 1;
                volatile float c0 = ((A * (fx + 1.f) - 5.f * A) * (fx + 1.f) + 8.f * A) * (fx + 1.f) - 4.f * A;
                volatile float c1 = ((A + 2.f) * fx - (A + 3.f)) * fx * fx + 1.f;
                volatile float c2 = ((A + 2.f) * (1.f - fx) - (A + 3.f)) * (1.f - fx) * (1.f - fx) + 1.f;
                volatile float c3 = 1.f - c0 - c1 - c2;
                const float cs[4] = {c0, c1, c2, c3};
                for (int k = 0; k < 4; ++k) {
                    const float r = std::nearbyint(cs[k] * 2048.f);
                    co[4 * (size_t)d + k] = (int)std::max(-32768.f, std::min(32767.f, r));
                }
            }
            o_src = ints.size(); ints.insert(ints.end(), src.begin(), src.end());
            o_c = ints.size(); ints.insert(ints.end(), co.begin(), co.end());
        };
        tab(sw, L.new_w, o_xsrc, o_xc);
        tab(sh, L.new_h, o_ysrc, o_yc);
    }
    const size_t bytes = ints.size() * 4 + flts.size() * 4 + 16;
    if (bytes > c->lb_cap) {
        if (c->lb_dev) (void)hipFree(c->lb_dev);
        c->lb_dev = nullptr; c->lb_cap = 0;
        if (hipMalloc(&c->lb_dev, bytes) != hipSuccess) { c->lb_dev = nullptr; return false; }
        c->lb_cap = bytes;
    }
    int *di = (int *)c->lb_dev;
    float *df = (float *)(di + ints.size());
    if (!ints.empty() && hipMemcpy(di, ints.data(), ints.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return false;
    if (!flts.empty() && hipMemcpy(df, flts.data(), flts.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return false;
    L.xbeg = di + o_xbeg; L.ybeg = di + o_ybeg; L.xsrc = di + o_xsrc; L.ysrc = di + o_ysrc;
    L.xc = di + o_xc; L.yc = di + o_yc; L.xw = df + o_xw; L.yw = df + o_yw;
    c->lb_key[0] = sh; c->lb_key[1] = sw; c->lb_key[2] = dh; c->lb_key[3] = dw;
    return true;
}

int hdrtv_letterbox_u8(hdrtv_ctx *c, void *stream, const uint8_t *src_bgr, int sh, int sw, uint8_t *dst_bgr, int dh, int dw)
{
    if (!c || !src_bgr || !dst_bgr || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0) return fail(c, HDRTV_EINVAL, "letterbox: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    if (!letterbox_setup(c, sh, sw, dh, dw)) {
        c->lb_key[0] = 0;
        return fail(c, HDRTV_ENOMEM, "letterbox: table allocation failed");
    }
    LetterboxParams p = c->lb;
    p.src = src_bgr; p.dst = dst_bgr;
    hipError_t e = letterbox_launch(p, (hipStream_t)stream);
    return e == hipSuccess ? HDRTV_OK : fail(c, HDRTV_EHIP, "letterbox: %s", hipGetErrorString(e));
}

// --------------------------------------------------------------------------------- metrics
int hdrtv_metrics(hdrtv_ctx *c, void *stream, const void *a, const void *b, int dtype, int H, int W, float peak_nits,
                  double *out3)
{
    if (!c || !a || !b || !out3 || H <= 0 || W <= 0 || !(peak_nits > 0.f)) return fail(c, HDRTV_EINVAL, "metrics: bad argument");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t nblk = (size_t)metrics_blocks(H, W);
    if (nblk > c->mt_cap) {
        if (c->mt_dev) (void)hipFree(c->mt_dev);
        c->mt_dev = nullptr; c->mt_cap = 0;
        if (hipMalloc((void **)&c->mt_dev, nblk * 3 * sizeof(double)) != hipSuccess) {
            c->mt_dev = nullptr;
            return fail(c, HDRTV_ENOMEM, "metrics: allocation failed");
        }
        c->mt_cap = nblk;
    }
    MetricsParams p;
    p.a = a; p.b = b; p.is_f32 = dtype == HDRTV_F32; p.H = H; p.W = W; p.peak_nits = peak_nits; p.partials = c->mt_dev;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = metrics_launch(p, s);
    if (e != hipSuccess) return fail(c, HDRTV_EHIP, "metrics: %s", hipGetErrorString(e));
    std::vector<double> host(nblk * 3);
    HIPCHK(c, hipMemcpyAsync(host.data(), c->mt_dev, host.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    double se = 0.0, ss = 0.0, de = 0.0;
    for (size_t i = 0; i < nblk; ++i) { se += host[3 * i]; ss += host[3 * i + 1]; de += host[3 * i + 2]; }
    const double npx = (double)H * W;
    const double mse = se / (3.0 * npx);
    out3[0] = mse <= 1e-12 ? 99.0 : 10.0 * std::log10(1.0 / mse);      // _psnr_bgr
    out3[1] = ss / (3.0 * npx);                                         // _ssim_bgr
    out3[2] = de / npx;                                                 // _delta_e_itp
    return HDRTV_OK;
}

// ------------------------------------------------------------------------------------ ring
int hdrtv_ring_create(hdrtv_ctx *c, int slots, int H, int W)
{
    if (!c || slots < 2 || slots > 8 || H <= 0 || W <= 0) return fail(c, HDRTV_EINVAL, "ring: slots must be 2..8");
    hdrtv_ring_destroy(c);
    HIPCHK(c, hipSetDevice(c->device));
    std::lock_guard<std::mutex> lk(c->ring_mu);
    const size_t bytes = (size_t)H * W * 3 * 2;
    c->ring.resize(slots);
    hipError_t e = hipSuccess;
    for (auto &sl : c->ring) {
        // The post kernel writes device memory and hdrtv_ring_commit moves it with hipMemcpyAsync: a kernel storing
        // straight into mapped host memory holds CUs for ~2 ms per 4K frame at PCIe speed, which the persistent
        // one-workgroup-per-CU convolutions of the next frame then wait for (measured: 75 -> 82 frames/s end to end)
        if (e == hipSuccess) e = hipHostMalloc((void **)&sl.host, bytes, hipHostMallocPortable);
        if (e == hipSuccess) e = hipMalloc((void **)&sl.dev, bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming);
        sl.state = 0;
    }
    if (e != hipSuccess) {                 // free the slots that were created before the failure
        for (auto &sl : c->ring) {
            if (sl.ev) (void)hipEventDestroy(sl.ev);
            if (sl.host) (void)hipHostFree(sl.host);
            if (sl.dev) (void)hipFree(sl.dev);
        }
        c->ring.clear();
        return fail(c, HDRTV_EHIP, "ring: allocation failed: %s", hipGetErrorString(e));
    }
    c->ring_next = 0; c->ring_H = H; c->ring_W = W;
    return HDRTV_OK;
}

int hdrtv_ring_acquire(hdrtv_ctx *c, int timeout_ms, uint16_t **host_ptr, uint16_t **dev_ptr)
{
    if (!c) return HDRTV_EINVAL;
    std::unique_lock<std::mutex> lk(c->ring_mu);
    if (c->ring.empty()) return fail(c, HDRTV_ESTATE, "ring not created");
    const int n = (int)c->ring.size();
    auto pick = [&]() -> int {
        for (int o = 0; o < n; ++o) {
            const int i = (c->ring_next + o) % n;
            if (c->ring[i].state == 0) return i;
        }
        return -1;
    };
    int i = pick();
    if (i < 0) {
        // (hdrtv_ring_destroy may run while this waits: an emptied ring ends the wait with an error)
        const bool got = c->ring_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms),
                                             [&] { return c->ring.empty() || (i = pick()) >= 0; });
        if (c->ring.empty()) return fail(c, HDRTV_ESTATE, "ring destroyed while waiting for a slot");
        if (!got) return fail(c, HDRTV_ESTATE, "no free ring slot within %d ms", timeout_ms);
    }
    c->ring[i].state = 1;
    c->ring_next = (i + 1) % n;
    if (host_ptr) *host_ptr = c->ring[i].host;
    if (dev_ptr) *dev_ptr = c->ring[i].dev;
    return i;
}

// Slot life cycle: 0 free -> (acquire) 1 acquired -> (commit) 2 committed -> (release) 0.  Every entry point looks its slot up
// and checks its state under ring_mu (the consumer thread calls wait / release while the producer acquires and commits, and
// hdrtv_ring_destroy may run between them): a call in the wrong state is HDRTV_ESTATE, not stale pixels.
int hdrtv_ring_commit(hdrtv_ctx *c, int slot, void *stream)
{
    if (!c) return HDRTV_EINVAL;
    uint16_t *host = nullptr, *dev = nullptr;
    hipEvent_t ev = nullptr;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(c->ring_mu);
        if (slot < 0 || slot >= (int)c->ring.size()) return fail(c, HDRTV_EINVAL, "bad ring slot");
        if (c->ring[slot].state != 1) return fail(c, HDRTV_ESTATE, "ring slot %d is not acquired (state %d)", slot, c->ring[slot].state);
        host = c->ring[slot].host; dev = c->ring[slot].dev; ev = c->ring[slot].ev;
        bytes = (size_t)c->ring_H * c->ring_W * 6;
        // enqueue under the lock: hdrtv_ring_destroy cannot free the buffers between the look-up and the copy (both calls
        // only enqueue; neither blocks on the device)
        hipError_t e = hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
        if (e == hipSuccess) e = hipEventRecord(ev, (hipStream_t)stream);
        if (e != hipSuccess) return fail(c, HDRTV_EHIP, "ring commit failed: %s", hipGetErrorString(e));
        c->ring[slot].state = 2;
        c->ring[slot].landed = false;
    }
    return HDRTV_OK;
}

// Blocks on a committed slot's event OUTSIDE ring_mu.  The event stays alive meanwhile: the call is counted in ring_waiters,
// and hdrtv_ring_destroy (also reached through hdrtv_ring_create) waits for that count to drop to zero before it frees anything.
static int ring_sync_slot(hdrtv_ctx *c, int slot, std::unique_lock<std::mutex> &lk, const char *what)
{
    hipEvent_t ev = c->ring[slot].ev;
    ++c->ring_waiters;
    lk.unlock();
    const hipError_t e = hipEventSynchronize(ev);
    lk.lock();
    --c->ring_waiters;
    c->ring_cv.notify_all();
    if (e != hipSuccess) return fail(c, HDRTV_EHIP, "ring %s failed: %s", what, hipGetErrorString(e));
    if (slot < (int)c->ring.size() && c->ring[slot].ev == ev && c->ring[slot].state == 2) c->ring[slot].landed = true;
    return HDRTV_OK;
}

int hdrtv_ring_wait(hdrtv_ctx *c, int slot)
{
    if (!c) return HDRTV_EINVAL;
    std::unique_lock<std::mutex> lk(c->ring_mu);
    if (slot < 0 || slot >= (int)c->ring.size()) return fail(c, HDRTV_EINVAL, "bad ring slot");
    // an acquired-but-uncommitted slot has no copy in flight: its event was never recorded (or belongs to the slot's
    // previous frame) and hipEventSynchronize would return at once on stale pixels
    if (c->ring[slot].state != 2) return fail(c, HDRTV_ESTATE, "ring slot %d is not committed (state %d)", slot, c->ring[slot].state);
    return ring_sync_slot(c, slot, lk, "wait");
}

int hdrtv_ring_release(hdrtv_ctx *c, int slot)
{
    if (!c) return HDRTV_EINVAL;
    {
        std::unique_lock<std::mutex> lk(c->ring_mu);
        if (slot < 0 || slot >= (int)c->ring.size()) return fail(c, HDRTV_EINVAL, "bad ring slot");
        if (c->ring[slot].state == 0) return fail(c, HDRTV_ESTATE, "ring slot %d is already free", slot);
        // a committed slot released without hdrtv_ring_wait: its device -> host copy may still be in flight, and a free slot
        // is re-acquired and overwritten by the next frame's post kernel -- the copy is waited for here
        if (c->ring[slot].state == 2 && !c->ring[slot].landed) {
            const hipEvent_t ev = c->ring[slot].ev;
            if (int rc = ring_sync_slot(c, slot, lk, "release")) return rc;
            if (slot >= (int)c->ring.size() || c->ring[slot].ev != ev) return fail(c, HDRTV_ESTATE, "ring destroyed during release");
        }
        c->ring[slot].state = 0;
        c->ring[slot].landed = false;
    }
    c->ring_cv.notify_all();
    return HDRTV_OK;
}

int hdrtv_ring_destroy(hdrtv_ctx *c)
{
    if (!c) return HDRTV_OK;
    std::unique_lock<std::mutex> lk(c->ring_mu);
    c->ring_cv.wait(lk, [&] { return c->ring_waiters == 0; });      // threads inside hipEventSynchronize on a slot's event
    for (auto &sl : c->ring) {
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
    }
    c->ring.clear();
    lk.unlock();
    c->ring_cv.notify_all();               // a producer blocked in hdrtv_ring_acquire sees the empty ring
    return HDRTV_OK;
}

int hdrtv_get_tap(hdrtv_ctx *c, const char *name, void **dev_ptr, int *C, int *H, int *W, int *layout)
{
    if (!c || !name) return HDRTV_EINVAL;
    auto it = c->t.find(name);
    if (it == c->t.end() || !c->ws.dev) return fail(c, HDRTV_EINVAL, "no tap named %s", name);
    if (dev_ptr) *dev_ptr = c->ws.dev + it->second.off;
    if (C) *C = it->second.C;
    if (H) *H = it->second.H;
    if (W) *W = it->second.W;
    if (layout) *layout = it->second.layout;
    return HDRTV_OK;
}

int hdrtv_infer_stats(hdrtv_ctx *c, int *launches, double *macs)
{
    if (!c) return HDRTV_EINVAL;
    if (launches) *launches = c->launches;
    if (macs) *macs = c->macs;
    return HDRTV_OK;
}

int hdrtv_profile_enable(hdrtv_ctx *c, int on)
{
    if (!c) return HDRTV_EINVAL;
    c->prof_on = on != 0;
    c->prof.clear();
    return HDRTV_OK;
}

int hdrtv_profile_get(hdrtv_ctx *c, int i, const char **layer, const char **kernel, float *ms, double *macs, double *bytes)
{
    if (!c) return HDRTV_EINVAL;
    const int n = (int)c->prof.size();
    if (i < 0) return n;
    if (i >= n || (size_t)i + 1 >= c->prof_ev.size()) return fail(c, HDRTV_EINVAL, "profile index out of range");
    hdrtv_ctx::ProfEntry &e = c->prof[i];
    HIPCHK(c, hipEventSynchronize(c->prof_ev[i + 1]));
    HIPCHK(c, hipEventElapsedTime(&e.ms, c->prof_ev[i], c->prof_ev[i + 1]));
    if (layer) *layer = e.layer.c_str();
    if (kernel) *kernel = e.kernel.c_str();
    if (ms) *ms = e.ms;
    if (macs) *macs = e.macs;
    if (bytes) *bytes = e.bytes;
    return HDRTV_OK;
}

// the message of the calling thread's view: copied out under err_mu (fail() may run on the producer and the consumer thread at once)
const char *hdrtv_last_error(const hdrtv_ctx *c)
{
    if (!c) return "null context";
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(c->err_mu);
    copy = c->err;
    return copy.c_str();
}

}  // extern "C"
