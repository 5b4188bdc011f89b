// api.h -- the host side of libhdrtv_mi355x.so: what its translation units share.
//   hdrtv_api.hip        the exported C ABI (include/hdrtv_mi355x.h), rings, letterbox / metrics / PQ tables
//   api_util.hip         fail(), the developer-variant table
//   api_pack.hip         state_dict -> MFMA operand layouts (build_weights); nothing else of it is visible outside
//   api_workspace.hip    per-resolution workspace (do_reserve), shapes
//   api_graph.hip        launch sequencing of hdrtv_infer: struct Seq, run_agcm / run_le / run_hg
//   fp32_graph.hip       precision="fp32": build_weights_f32, run_f32
// Kernels live in the other .hip files behind launchers.h; no kernel is launched from a header.
#pragma once
#include "../../include/hdrtv_mi355x.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "launchers.h"

namespace hdrtv_host {

// ------------------------------------------------------------------------------ weight pack
struct PackEntry {
    int dtype;  // 0 f32, 1 f16, 2 i8, 3 i64
    int ndim;
    int dims[4];
    const unsigned char *data;
    size_t nbytes;
    size_t numel() const
    {
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= (size_t)dims[i];
        return n;
    }
};

struct Pack {
    std::map<std::string, PackEntry> e;
    bool parse(const void *blob, size_t bytes, std::string &err)
    {
        const unsigned char *b = (const unsigned char *)blob;
        if (bytes < 16 || memcmp(b, "HDRW1\0\0\0", 8) != 0) { err = "not an HDRW1 weight pack"; return false; }
        uint32_t n;
        memcpy(&n, b + 8, 4);
        if (16 + (size_t)n * 136 > bytes) { err = "weight pack truncated (table)"; return false; }
        for (uint32_t i = 0; i < n; ++i) {
            const unsigned char *r = b + 16 + (size_t)i * 136;
            char name[97];
            memcpy(name, r, 96);
            name[96] = 0;
            PackEntry pe;
            uint32_t dt, nd, d[4];
            uint64_t off, nb;
            memcpy(&dt, r + 96, 4); memcpy(&nd, r + 100, 4); memcpy(d, r + 104, 16);
            memcpy(&off, r + 120, 8); memcpy(&nb, r + 128, 8);
            // the blob crosses the C ABI: every table field is checked before anything is read through it
            if (off > bytes || nb > bytes - off || nd > 4) { err = std::string("weight pack truncated: ") + name; return false; }
            if (dt > 3) { err = std::string("weight pack: bad dtype for ") + name; return false; }
            static const size_t esz[4] = {4, 2, 1, 8};
            uint64_t numel = 1;
            for (uint32_t k = 0; k < nd; ++k) {
                if (d[k] == 0 || d[k] > (1u << 28) || numel > (1ull << 40) / d[k]) { err = std::string("weight pack: bad shape for ") + name; return false; }
                numel *= d[k];
            }
            if (nb != numel * esz[dt]) { err = std::string("weight pack: size does not match shape for ") + name; return false; }
            pe.dtype = (int)dt; pe.ndim = (int)nd;
            for (int k = 0; k < 4; ++k) pe.dims[k] = (int)d[k];
            pe.data = b + off; pe.nbytes = nb;
            e[name] = pe;
        }
        return true;
    }
    // fetch as fp32 vector with an expected element count
    bool get(const std::string &name, size_t numel, std::vector<float> &out, std::string &err) const
    {
        auto it = e.find(name);
        if (it == e.end()) { err = "tensor missing from weight pack: " + name; return false; }
        const PackEntry &pe = it->second;
        if (pe.numel() != numel) { err = "bad shape for " + name; return false; }
        out.resize(numel);
        if (pe.dtype == 0) memcpy(out.data(), pe.data, numel * 4);
        else if (pe.dtype == 1) { const f16 *s = (const f16 *)pe.data; for (size_t i = 0; i < numel; ++i) out[i] = (float)s[i]; }
        else { err = "unsupported dtype for " + name; return false; }
        return true;
    }
    bool has(const std::string &name) const { return e.find(name) != e.end(); }
    // A layer's weight tensor as the reference's layer computes it: `<layer>.weight`, or for the INT8 runtime layers
    // (W8Conv2d / W8A8Conv2d / W8Linear / W8A8Linear, hdrtvnet_torch.py:233-410) weight_int8 * scale, the product
    // rounded once to f16 as `weight_int8.to(cd) * scale` is on a GPU (cd = fp16)
    bool getw(const std::string &layer, size_t numel, std::vector<float> &out, std::string &err, bool round_f16 = true) const
    {
        if (has(layer + ".weight")) return get(layer + ".weight", numel, out, err);
        std::vector<int8_t> q;
        if (!get_i8(layer + ".weight_int8", numel, q, err)) return false;
        const std::string sn = has(layer + ".w_scale") ? layer + ".w_scale" : layer + ".scale";
        auto it = e.find(sn);
        if (it == e.end()) { err = "INT8 layer without scale: " + layer; return false; }
        const size_t co = it->second.numel();
        std::vector<float> sc;
        if (co == 0 || numel % co || !get(sn, co, sc, err)) { if (err.empty()) err = "bad scale for " + layer; return false; }
        out.resize(numel);
        const size_t per = numel / co;
        // round_f16 = false: a W8A8 layer evaluated as fp32 fake-quant (AGCM classifier / Linear heads): weight_int8.float() * w_scale
        for (size_t i = 0; i < numel; ++i) out[i] = round_f16 ? (float)(f16)((float)q[i] * (float)(f16)sc[i / per]) : (float)q[i] * sc[i / per];
        return true;
    }
    bool is_w8a8(const std::string &layer) const { return has(layer + ".weight_int8") && has(layer + ".x_scale"); }
    bool get_i8(const std::string &name, size_t numel, std::vector<int8_t> &out, std::string &err) const
    {
        auto it = e.find(name);
        if (it == e.end()) { err = "tensor missing from weight pack: " + name; return false; }
        const PackEntry &pe = it->second;
        if (pe.numel() != numel || pe.dtype != 2) { err = "bad shape or dtype (want int8) for " + name; return false; }
        out.assign((const int8_t *)pe.data, (const int8_t *)pe.data + numel);
        return true;
    }
};

// --------------------------------------------------------------------------- device arenas
struct Arena {
    std::vector<unsigned char> host;   // staging (weights) -- empty for workspace arenas
    unsigned char *dev = nullptr;
    size_t size = 0;
    size_t reserve(size_t bytes)
    {
        const size_t off = (size + 255) & ~(size_t)255;
        size = off + bytes;
        return off;
    }
    size_t put(const void *src, size_t bytes)
    {
        const size_t off = reserve(bytes);
        if (host.size() < size) host.resize(size);
        memcpy(host.data() + off, src, bytes);
        return off;
    }
};

struct ConvLayer {
    size_t wpk = 0, scale = 0, shift = 0;   // weight-arena offsets
    int cin = 0, cout = 0, coutPad = 0, ks = 0, stride = 1, cin_t = 0, bn = 0;
};
struct ConvI8Layer {                        // W8A8 HG layer on int8 MFMA
    size_t wpk = 0, scale = 0, shift = 0, padline = 0, delta = 0, delta_acc = 0;
    bool has_delta = false;
    float lo_clamp = -128.f;
    int cin = 0, cout = 0, cout_real = 0, ks = 0, out_f16 = 0;   // cout: padded to a multiple of 128
};
struct C3Layer { size_t wfrag = 0, scale = 0, shift = 0; int cout = 0; };
// W8A8Conv2d's activation quantiser (hdrtvnet_torch.py:351-364) with a FLOAT zero point: value = scale * (code + off),
// int8 code = q - 128.  Symmetric layers (no x_zero buffer): q = round(x / scale) + 128, off = 0.
struct ActQf {
    float scale = 1.f, zero = 0.f;
    bool asym = true;
    float inv() const { return 1.f / scale; }
    float zoff() const { return asym ? -zero / scale : 128.f; }                           // u8 code = clamp(rint(x * inv + zoff), 0, 255)
    double soff() const { return asym ? 128.0 * (double)scale + (double)zero : 0.0; }     // scale * off
};
struct QLayer {                             // W8A8 layer of the HR network on int8 MFMA (conv32p<..,i8> / conv_q8)
    size_t wpk8 = 0, scale = 0, shift = 0;  // shift: [16 border classes][coutPad]
    ActQf q;
    int cin = 0, cout = 0, coutPad = 0, ks = 0, stride = 1;
};
struct QLastLayer { size_t wq = 0, ss = 0; ActQf q; bool on = false; };
struct SftLayer {
    size_t wfrag = 0, bias = 0;
    bool q = false;                         // all four 1x1 convs are W8A8: int8 fragments + dequantisation constants
    size_t qfrag = 0, qconst = 0;
    float inv[2] = {0, 0}, zoff[2] = {0, 0}, hzoff[2] = {0, 0};
    ActQf fq[4];                            // the input quantisers of scale_conv0, shift_conv0, scale_conv1, shift_conv1 (le_rows.hip's fake-quant form)
};

struct Tensor {
    size_t off = 0;
    int C = 0, H = 0, W = 0, layout = 0;   // 0 NHWC f16, 1 planar f16, 2 planar f32, 3 f32 vector, 4 u8 plane, 5 NHWC int8 codes
    size_t bytes() const
    {
        const size_t n = (size_t)C * H * W;
        return layout == 2 || layout == 3 ? n * 4 : (layout == 4 || layout == 5 ? n : n * 2);
    }
};

struct F32Layer {                            // precision="fp32": one conv layer of fp32_graph.hip
    size_t w = 0, b = 0, bn_s = 0, bn_t = 0;
    int cin = 0, cout = 0, cot = 32, ks = 1;
    bool bn = false;
};

struct RingSlot {
    uint16_t *host = nullptr, *dev = nullptr;
    // host: page-locked slot the consumer reads; dev: device staging buffer the post kernel writes (commit copies)
    hipEvent_t ev = nullptr;
    int state = 0;   // 0 free, 1 acquired, 2 committed
    bool landed = false;   // state 2: hdrtv_ring_wait has seen the copy land
};

}  // namespace hdrtv_host

struct hdrtv_ctx {
    // (the types of hdrtv_host are this struct's vocabulary: the public header only names it as an opaque struct)
    int device = 0;
    int n_cu = 256, dev_ncu = 256;        // n_cu: what grids are sized for (variant force_ncu); dev_ncu: the device's
    bool has_hg = false;
    bool fp32 = false;                    // hdrtv_create_ex(..., HDRTV_PREC_F32): the fp32 graph (fp32_graph.hip) on planar fp32 tensors
    std::map<std::string, hdrtv_host::F32Layer> conv32f;
    std::string err;
    mutable std::mutex err_mu;            // fail() runs on the producer and the consumer thread (ring entry points)
    hdrtv_host::Arena wts;
    std::map<std::string, hdrtv_host::ConvLayer> conv;
    std::map<std::string, hdrtv_host::C3Layer> c3;
    std::map<std::string, hdrtv_host::ConvI8Layer> conv8;
    float mask_r = 0.75f;                 // HG_Composite(mask_r=0.75), HG_Composite_arch.py:21
    int cond_mode = 0;                    // 0 AA-bicubic, 1 bilinear (fast_condition_resize), 2 zero (HDRTVNET_ZERO_COND)
    bool hg_i8 = false;                   // the HG pack is a W8A8 checkpoint: 15 layers run on int8 MFMA
    float hg_q0_inv = 0.f, hg_q0_zero = 0.f;   // quantiser of the fp16 -> int8 boundary (conv2's output)
    std::map<std::string, hdrtv_host::SftLayer> sft;
    bool hr_i8 = false;                   // the HR pack holds W8A8 layers: they run on int8 MFMA (predequantize off)
    std::map<std::string, hdrtv_host::QLayer> q32, q8;
    hdrtv_host::QLastLayer q_trunk6, q_tail2;         // CondNet1.4 / CondNet2.4 as the W8A8 last layer of their fused chains
    // fully quantised chains (le_chain_q8.hip) and the fp32 fake-quant of the AGCM classifier / Linear heads
    bool trunk_q8 = false, tail_q8 = false, agcm_q8 = false;
    size_t tq_frag = 0, tq_const = 0, tl_frag = 0, tl_const = 0, ag_frag = 0, ag_P = 0, ag_Q = 0;
    hdrtv_host::ActQf tq_q[6], tl_q[2], ag_q[3];
    FakeQ cls_q[6] = {}, lin_q[6] = {};
    std::map<std::string, size_t> f32v;   // raw fp32 vectors/matrices in the weight arena
    size_t zeros_off = 0;                 // 256 B of zeros in the weight arena
    size_t dump_off = 0;                  // 8 KiB write-only scratch (conv32p masked lanes)
    size_t trunk_wfrag = 0, trunk_bias = 0;   // fused LE condition trunk (le_fused.hip)
    size_t tail_wfrag = 0, tail_bias = 0;     // fused CondNet2 tail (le_fused.hip, cond_tail_kernel)
    size_t hgf_wfrag = 0, hg_w10a = 0;        // fused HG tail: conv1 + conv10(second half) fragments, conv10 first half
    // workspace
    int H = 0, W = 0;
    hdrtv_host::Arena ws;                 // ws.dev: the workspace of the lane being launched (lane 0 outside hdrtv_infer_lane)
    int lanes = 1;                        // hdrtv_set_lanes: one activation workspace per frame in flight
    std::vector<unsigned char *> lane_ws; // [lanes] after hdrtv_reserve; lane_ws[0] == ws.dev
    std::map<std::string, hdrtv_host::Tensor> t;
    int launches = 0;
    double macs = 0.0;
    // per-launch profile (hdrtv_profile_*): event i is recorded after launch i-1
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    struct ProfEntry { std::string layer, kernel; double macs, bytes; float ms; };
    std::vector<ProfEntry> prof;
    // letterbox tables (hdrtv_letterbox_u8): device copy for the last geometry
    int lb_key[4] = {0, 0, 0, 0};
    LetterboxParams lb{};
    void *lb_dev = nullptr;
    size_t lb_cap = 0;
    float *pq_bnd = nullptr;              // hdrtv_post_pq_rgb48: the 65536 code boundaries of the PQ quantiser (built on first use)
    // objective metrics partial sums
    double *mt_dev = nullptr;
    size_t mt_cap = 0;
    // ring
    std::vector<hdrtv_host::RingSlot> ring;
    int ring_next = 0, ring_H = 0, ring_W = 0, ring_waiters = 0;   // ring_waiters: threads blocked on a slot's event outside ring_mu
    std::mutex ring_mu;
    std::condition_variable ring_cv;
    // developer variant table (hdrtv_set_variant): which of several equivalent kernels / schedules a layer runs on.  Filled
    // once in hdrtv_create (defaults, then HDRTV_VARIANTS="name=value,..." of the creating process); never read from the
    // environment on the launch path.
    std::map<std::string, int> var;
};

namespace hdrtv_host {

// ---- api_util.hip: errors, developer variants
int fail(hdrtv_ctx *c, int code, const char *fmt, ...);
bool variant_allowed(const std::string &name, int value);
void variants_init(hdrtv_ctx *c);              // defaults, then HDRTV_VARIANTS of the creating process

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return hdrtv_host::fail(c, HDRTV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ---- api_pack.hip / fp32_graph.hip: the reference's state_dict -> operand layouts in c->wts (hdrtv_create)
bool build_weights(hdrtv_ctx *c, const Pack &hr, const Pack *hg);
bool build_weights_f32(hdrtv_ctx *c, const Pack &hr, const Pack *hg);

// ---- api_workspace.hip: the per-resolution workspace (hdrtv_reserve)
struct Shapes {
    int H, W, h4, w4;
    int ch[6], cw[6];        // classifier spatial sizes: [0]=cond, [i]=after block i
    int H1, W1, H2, W2, H3, W3;
    int Hp, Wp;
};
Shapes shapes_for(int H, int W);
Tensor &ws_add(hdrtv_ctx *c, const std::string &name, int C, int H, int W, int layout);
int do_reserve(hdrtv_ctx *c, int H, int W);
void free_workspaces(hdrtv_ctx *c);            // every lane's; leaves no reserved size behind
template <typename T>
T *wsp(hdrtv_ctx *c, const std::string &name)
{
    auto it = c->t.find(name);
    if (it == c->t.end()) { fprintf(stderr, "hdrtv: internal error, no workspace tensor %s\n", name.c_str()); abort(); }
    return reinterpret_cast<T *>(c->ws.dev + it->second.off);
}
template <typename T>
const T *wtp(hdrtv_ctx *c, size_t off) { return reinterpret_cast<const T *>(c->wts.dev + off); }

// ---- api_graph.hip: launch sequencing of one hdrtv_infer (AGCM, LE, HG); fp32_graph.hip: the fp32 preset's graph
struct Seq {
    hdrtv_ctx *c;
    hipStream_t s;
    int rc = HDRTV_OK;
    // conv3x3s2_preg<192> only, consumed by the next conv(): CondNet2's 1x1 tail fused behind its first 64 output channels
    const f16 *tail_w = nullptr;
    const float *tail_b = nullptr, *tail_s = nullptr;       // tail_s != null: conv3x3s2_preg<64>'s single-layer tail (CondNet3.4)
    f16 *tail_out = nullptr;
    bool ok() const { return rc == HDRTV_OK; }
    void mark()
    {
        if (!c->prof_on) return;
        const size_t i = c->prof.size();
        while (c->prof_ev.size() <= i) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) { c->prof_on = false; return; }
            c->prof_ev.push_back(ev);
        }
        (void)hipEventRecord(c->prof_ev[i], s);
    }
    // called after every launch: counts it, checks it and (profiling) closes its event interval
    void chk(hipError_t e, const char *what, const char *kernel = "", double macs = 0.0, double bytes = 0.0)
    {
        ++c->launches;
        c->macs += macs;
        if (e != hipSuccess && rc == HDRTV_OK) rc = fail(c, HDRTV_EHIP, "launch %s failed: %s", what, hipGetErrorString(e));
        if (c->prof_on) {
            c->prof.push_back({what, kernel, macs, bytes, 0.f});
            mark();
        }
    }
    // generic conv: src0 (+src1) -> dst
    void conv(const std::string &key, const f16 *src0, int c0, const f16 *src1, int c1, int Hi, int Wi, int act, int mode, f16 *dst, int dstC, int Hd, int Wd, const f16 *res1 = nullptr, const f16 *res2 = nullptr, f16 *dst_full = nullptr, f16 *dst_planar = nullptr, const f16 *res_planar = nullptr, const float *dotw = nullptr, float *dst_dot = nullptr, int s0_stride = 0);
    // W8A8 HG layer on int8 MFMA: 3x3 (conv3x3_pglds_i8.hip) or 1x1 (conv_i8_misc.hip)
    void conv8(const std::string &key, const int8_t *src0, int c0, const int8_t *src1, int c1, int Hi, int Wi, int mode, void *dst, int dstC, int Hd, int Wd, const float *dotw = nullptr, float *dst_dot = nullptr);
    // W8A8 LE layer on int8 MFMA (conv_q8.hip).  src: f16 NHWC (quantised on load) or this layer's int8 codes; dst: f16, or
    // (oq != nullptr) the int8 codes of the reading layer's quantiser *oq
    void convq8(const std::string &key, const void *src, bool src_i8, int src_stride, int Hi, int Wi, int act, void *dst, int dstC, const ActQf *oq);
    void c3(const std::string &key, const f16 *in, int H, int W, int act, f16 *out, f16 *out_pool, float pool_q_inv = 0.f, float pool_q_zero = 0.f, const f16 *w2frag = nullptr, float *part2 = nullptr);
    // persistent 32-channel 3x3 conv, optionally with the SFT layer `sft_key` fused in front (conv32p.hip)
    void conv32(const std::string &key, const f16 *src, const f16 *cond, const std::string &sft_key, int H, int W, int act, int mode, f16 *dst, int dstC, int Hd, int Wd, const f16 *res1 = nullptr, const f16 *res2 = nullptr, f16 *dst_planar = nullptr, const f16 *res_planar = nullptr, const f16 *c3_img = nullptr, const std::string &c3_key = "");
    // diagnostic builds (make STAMP=1) write per-phase cycle sums of launch number HDRTV_STAMP_LAUNCH (read once) here
    void *stamp_buf() const
    {
#ifdef HDRTV_STAMP
        static const int stamp_launch = [] { const char *e = getenv("HDRTV_STAMP_LAUNCH"); return e ? atoi(e) : -1; }();
        return (stamp_launch >= 0 && c->launches == stamp_launch) ? (void *)wsp<f16>(c, "dbg.stamps") : nullptr;
#else
        return nullptr;
#endif
    }
    // a conv inside a fused row kernel: its fp16 pack, or (W8A8 layer, variant le_rows_fq) the dequantised pack + its activation quantiser
    static FqParam fqp(const ActQf &q) { return FqParam{q.inv(), q.zoff(), q.scale, q.asym ? q.zero : -128.f * q.scale}; }
    const ConvLayer *rows_conv(const std::string &key, FqParam &fq, bool &on) const
    {
        on = false;
        auto it = c->conv.find(key);
        if (it != c->conv.end()) return &it->second;
        if (!c->var.at("le_rows_fq")) return nullptr;
        it = c->conv.find(key + "#fq");
        if (it == c->conv.end()) return nullptr;
        auto iq = c->q32.find(key);
        const ActQf *q = iq != c->q32.end() ? &iq->second.q : nullptr;
        if (!q) { auto i8 = c->q8.find(key); if (i8 != c->q8.end()) q = &i8->second.q; }
        if (!q) return nullptr;
        fq = fqp(*q);
        on = true;
        return &it->second;
    }
    // its SFT layer: fp16 convs, or all four W8A8 (fake-quant)
    bool rows_sft(const SftLayer &S, FqParam (&fq)[4], bool &on) const
    {
        on = S.q;
        if (S.q && !c->var.at("le_rows_fq")) return false;
        for (int i = 0; i < 4; ++i) fq[i] = fqp(S.fq[i]);
        return true;
    }
    // the row-streaming kernels (le_rows.hip) cut a map into 60-column strips x row segments, one workgroup each: worth it
    // when a segment is long against its 4 .. 6 warm-up rows
    bool rows_fit(int H, int W) const
    {
        const int nstrips = (W + 59) / 60, nseg = std::max(1, c->n_cu / nstrips);
        return W >= 60 && (H + nseg - 1) / nseg >= c->var.at("le_rows_min");
    }
    // ResBlock_with_SFT (arch_util.py:89-95): y = x + conv2(sft2(relu(conv1(sft1(x,c))),c))  [+ extra]; 2 launches
    void resblock(const std::string &base, const f16 *x, const f16 *cond, int H, int W, f16 *tb, f16 *y, const f16 *extra = nullptr);
};

int run_agcm(hdrtv_ctx *c, Seq &q, const f16 *rgb, const f16 *cond, f16 *agcm_out);
int run_le(hdrtv_ctx *c, Seq &q, const f16 *img, f16 *out_planar);
int run_hg(hdrtv_ctx *c, Seq &q, const f16 *base, void *out, int out_f32);
int run_f32(hdrtv_ctx *c, Seq &q, bool plan, int H, int W, const float *rgb, const float *cond, float *out, float *agcm_out);
int f32_plan(hdrtv_ctx *c, int H, int W);      // registers the fp32 graph's tensors (hdrtv_reserve)

}  // namespace hdrtv_host
