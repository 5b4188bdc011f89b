// hdrtv_api.hip -- C ABI of libhdrtv_mi355x.so (see include/hdrtv_mi355x.h).
// Host side only: weight-pack parsing, repacking into MFMA operand layouts, workspace
// management and the per-frame launch sequence.  No kernel lives in this file.
#include "../../include/hdrtv_mi355x.h"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "launchers.h"

namespace {

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

struct F32Layer {                            // precision="fp32": one conv layer of fp32_graph.inc
    size_t w = 0, b = 0, bn_s = 0, bn_t = 0;
    int cin = 0, cout = 0, cot = 32, ks = 1;
    bool bn = false;
};

struct RingSlot {
    uint16_t *host = nullptr, *dev = nullptr;
    // host: page-locked slot the consumer reads; dev: device staging buffer the post kernel writes (commit copies)
    hipEvent_t ev = nullptr;
    int state = 0;   // 0 free, 1 acquired, 2 committed
};

}  // namespace

struct hdrtv_ctx {
    int device = 0;
    int n_cu = 256;
    bool has_hg = false;
    bool fp32 = false;                    // hdrtv_create_ex(..., HDRTV_PREC_F32): the fp32 graph (fp32_graph.inc) on planar fp32 tensors
    std::map<std::string, F32Layer> conv32f;
    std::string err;
    Arena wts;
    std::map<std::string, ConvLayer> conv;
    std::map<std::string, C3Layer> c3;
    std::map<std::string, ConvI8Layer> conv8;
    float mask_r = 0.75f;                 // HG_Composite(mask_r=0.75), HG_Composite_arch.py:21
    int cond_mode = 0;                    // 0 AA-bicubic, 1 bilinear (fast_condition_resize), 2 zero (HDRTVNET_ZERO_COND)
    bool hg_i8 = false;                   // the HG pack is a W8A8 checkpoint: 15 layers run on int8 MFMA
    float hg_q0_inv = 0.f, hg_q0_zero = 0.f;   // quantiser of the fp16 -> int8 boundary (conv2's output)
    std::map<std::string, SftLayer> sft;
    bool hr_i8 = false;                   // the HR pack holds W8A8 layers: they run on int8 MFMA (predequantize off)
    std::map<std::string, QLayer> q32, q8;
    QLastLayer q_trunk6, q_tail2;         // CondNet1.4 / CondNet2.4 as the W8A8 last layer of their fused chains
    // fully quantised chains (le_chain_q8.hip) and the fp32 fake-quant of the AGCM classifier / Linear heads
    bool trunk_q8 = false, tail_q8 = false, agcm_q8 = false;
    size_t tq_frag = 0, tq_const = 0, tl_frag = 0, tl_const = 0, ag_frag = 0, ag_P = 0, ag_Q = 0;
    ActQf tq_q[6], tl_q[2], ag_q[3];
    FakeQ cls_q[6] = {}, lin_q[6] = {};
    std::map<std::string, size_t> f32v;   // raw fp32 vectors/matrices in the weight arena
    size_t zeros_off = 0;                 // 256 B of zeros in the weight arena
    size_t dump_off = 0;                  // 8 KiB write-only scratch (conv32p masked lanes)
    size_t trunk_wfrag = 0, trunk_bias = 0;   // fused LE condition trunk (le_fused.hip)
    size_t tail_wfrag = 0, tail_bias = 0;     // fused CondNet2 tail (le_fused.hip, cond_tail_kernel)
    size_t hgf_wfrag = 0, hg_w10a = 0;        // fused HG tail: conv1 + conv10(second half) fragments, conv10 first half
    // workspace
    int H = 0, W = 0;
    Arena ws;
    std::map<std::string, Tensor> t;
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
    std::vector<RingSlot> ring;
    int ring_next = 0, ring_H = 0, ring_W = 0;
    std::mutex ring_mu;
    std::condition_variable ring_cv;
    // developer variant table (hdrtv_set_variant): which of several equivalent kernels / schedules a layer runs on.  Filled
    // once in hdrtv_create (defaults, then HDRTV_VARIANTS="name=value,..." of the creating process); never read from the
    // environment on the launch path.
    std::map<std::string, int> var;
};

namespace {

int fail(hdrtv_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

// ---- developer variants: name -> default.  hdrtv_set_variant changes one on a context; HDRTV_VARIANTS="a=1,b=0" seeds them at
// hdrtv_create.  The launch path reads c->var only.
const std::pair<const char *, int> k_variants[] = {
    {"le_rows", 1},          // fused row-streaming LE kernels (le_rows.hip); 0 = the per-layer 16x16-tile kernels
    {"le_rows_min", 12},     // ... when a strip segment has at least this many rows (it pays 4 .. 6 warm-up rows)
    {"le_rows_fq", 1},       // ... also for W8A8 layers (fake-quant in registers, fp16 MFMA); 0 = those layers on the int8-MFMA per-layer kernels
    {"prw", 1},              // HG 3x3 convs on conv_prw: 0 never (conv_pglds), 1 the cheapest shape per layer, 2 / 3 16-row / 8-row tiles wherever it applies
    {"prw_i8", 1},           // int8 HG 3x3 convs on conv_prw_i8: 0 never, 1 only where the 8-row tiles win, 2 wherever "prw" selects it
    {"pglds_nt_slow", 3},    // conv_pglds tile order: 0 / 1 Cout-tile fastest / slowest, 2 slowest for Cout >= 512, 3 slowest for the Up convs
    {"no_t16", 0},           // 1: LE's stride-2 down-convs on the generic implicit-GEMM kernel
    {"conv32_old", 0},       // 1: single-pass LE convs on conv32p's two-barrier schedule instead of conv32s
    {"conv32_nosplit", 0},   // 1: conv32s without the conv / prep role split
    {"conv32_nw", 0},        // conv32p tile shape: 0 per layer, 8 16x16 tiles, 4 8x16 tiles x 2 workgroups per CU
    {"no_c3fuse", 0},        // 1: LE.conv_first as its own launch in front of HR_conv1
    {"no_c3q8", 0},          // 1: the W8A8 LE.conv_first through planar3_to_q8 + conv_q8 instead of conv_c3_q8
    {"glds1_old", 0},        // 1: HG 1x1 fuse convs on the non-persistent kernel
    {"final_recompute", 0},  // 1: HG tail recomputes conv1 (hg_final_fused) instead of reading conv1's per-pixel sums
    {"pre_split", 0},        // 1: preprocess as two kernels (unpack, condition resize)
    {"force_ncu", 0},        // > 0: pretend the device has this many CUs (persistent grids)
    {"f32_narrow_below", 0}, // precision="fp32": workgroups per CU below which conv_f32 runs 8 channels per lane (0 = 3)
};
// variants whose non-default settings select kernels that exist in the A/B library only (make AB=1 -> libhdrtv_mi355x_ab.so):
// superseded schedules kept as bit-identity yardsticks of the shipped ones
const char *const k_ab_only[] = {"conv32_old", "conv32_nosplit", "conv32_nw", "glds1_old", "final_recompute"};
bool variant_allowed(const std::string &name, int value)
{
#ifdef HDRTV_AB
    (void)name; (void)value;
    return true;
#else
    if (value == 0) return true;
    for (const char *n : k_ab_only) if (name == n) return false;
    return true;
#endif
}
void variants_init(hdrtv_ctx *c)
{
    for (const auto &kv : k_variants) c->var[kv.first] = kv.second;
    const char *e = getenv("HDRTV_VARIANTS");
    if (!e) return;
    std::string str(e);
    size_t pos = 0;
    while (pos < str.size()) {
        size_t nx = str.find(',', pos);
        if (nx == std::string::npos) nx = str.size();
        const std::string one = str.substr(pos, nx - pos);
        const size_t eq = one.find('=');
        if (eq != std::string::npos) {
            auto it = c->var.find(one.substr(0, eq));
            if (it != c->var.end() && variant_allowed(it->first, atoi(one.c_str() + eq + 1))) it->second = atoi(one.c_str() + eq + 1);
        }
        pos = nx + 1;
    }
}

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(c, HDRTV_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------------- weight repacking
// Implicit-GEMM conv: [Co][Ci][K][K] f32 -> wpk [K*K][Ci/CT][CoPad][CT] f16, per-channel scale/shift.
// ps_cps > 0: output channels are re-ordered for a fused PixelShuffle(2): packed row
// n' = sub*cps + c holds original channel 4*c + sub.
bool pack_conv(hdrtv_ctx *c, const Pack &pk, const std::string &key, const std::string &wname, int co, int ci, int ks,
               int stride, const std::string &bn_name, int ps_cps, int force_ct = 0)
{
    // wname may list several layers separated by '+': their output channels are concatenated
    std::vector<float> w, b;
    {
        size_t pos = 0;
        int parts = 1;
        for (char ch : wname) parts += ch == '+';
        const int co1 = co / parts;
        while (pos <= wname.size()) {
            size_t nx = wname.find('+', pos);
            if (nx == std::string::npos) nx = wname.size();
            const std::string one = wname.substr(pos, nx - pos);
            std::vector<float> w1, b1;
            if (!pk.getw(one, (size_t)co1 * ci * ks * ks, w1, c->err)) return false;
            if (!pk.get(one + ".bias", (size_t)co1, b1, c->err)) return false;
            w.insert(w.end(), w1.begin(), w1.end());
            b.insert(b.end(), b1.begin(), b1.end());
            pos = nx + 1;
        }
    }
    ConvLayer L;
    L.cin = ci; L.cout = co; L.ks = ks; L.stride = stride;
    L.coutPad = (co + 31) / 32 * 32;
    if (ks == 1 && stride == 1 && ci % 64 == 0 && ci >= 128 && co == 64) L.coutPad = 128;   // 1x1 64-out layers ride the 128-wide LDS-DMA kernel
    L.cin_t = force_ct ? force_ct : ((stride == 2 || ci == 32) ? 32 : 64);
    L.bn = L.coutPad >= 128 ? 128 : L.coutPad;
    if (force_ct) L.bn = L.coutPad;     // whole-Cout kernels (conv3x3s2_preg)
    if (ci % L.cin_t != 0 || L.coutPad % L.bn != 0) { c->err = "unsupported conv shape: " + wname; return false; }
    std::vector<float> scale(L.coutPad, 1.f), shift(L.coutPad, 0.f);
    std::vector<float> g, be, mu, var;
    const bool has_bn = !bn_name.empty();
    if (has_bn) {
        if (!pk.getw(bn_name, co, g, c->err) || !pk.get(bn_name + ".bias", co, be, c->err) ||
            !pk.get(bn_name + ".running_mean", co, mu, c->err) || !pk.get(bn_name + ".running_var", co, var, c->err))
            return false;
    }
    const int nch = ci / L.cin_t, ct = L.cin_t;
    std::vector<f16> wp((size_t)ks * ks * nch * L.coutPad * ct, (f16)0.f);
    for (int np = 0; np < co; ++np) {
        const int n = ps_cps > 0 ? 4 * (np % ps_cps) + np / ps_cps : np;
        if (has_bn) {
            const float s = g[n] / std::sqrt(var[n] + 1e-5f);
            scale[np] = s;
            shift[np] = (b[n] - mu[n]) * s + be[n];
        } else {
            shift[np] = b[n];
        }
        for (int k = 0; k < ci; ++k)
            for (int tap = 0; tap < ks * ks; ++tap)
                wp[(((size_t)tap * nch + k / ct) * L.coutPad + np) * ct + k % ct] = (f16)w[((size_t)n * ci + k) * ks * ks + tap];
    }
    L.wpk = c->wts.put(wp.data(), wp.size() * sizeof(f16));
    L.scale = c->wts.put(scale.data(), scale.size() * 4);
    L.shift = c->wts.put(shift.data(), shift.size() * 4);
    c->conv[key] = L;
    return true;
}

// W8A8 layer (W8A8Conv2d, hdrtvnet_torch.py:296-364, asymmetric) for the int8 kernels: weight_int8 [Co][Ci][K][K] ->
// wpk [K*K][Ci/128][Co][128]; activation codes are q - 128 with an integer zero point k = -x_zero / x_scale, so
//     y = x_scale * w_scale[n] * (acc + (128 - k) * sum(w_int8[n])) + bias[n]      (then BatchNorm, folded)
// and the epilogue's {scale, shift} map acc straight to the OUTPUT tensor's codes (out_scale, out_k) or, out_scale == 0,
// to real units for an fp16 consumer.
// activation quantiser of a W8A8 HG layer: value = scale * (q - kf), q the reference's u8 code, kf = -x_zero / x_scale.  An integer
// kf in 0..255 (k) is an exact code for 0.0: padding is then a constant line of that code and the whole layer is integer-exact;
// any other zero point (e.g. calibrate_w8a8's x_zero = running minimum) pads with code 128 (a zero in the centred sum) and
// corrects the pixels on the image border with a per-class constant (ConvI8Params.delta).
struct ActQ { float scale = 0.f; double kf = 0.0; int k = 0; bool integer = true; };
bool read_actq(hdrtv_ctx *c, const Pack &pk, const std::string &layer, ActQ &q)
{
    std::vector<float> xs, xz;
    if (!pk.get(layer + ".x_scale", 1, xs, c->err) || !pk.get(layer + ".x_zero", 1, xz, c->err)) return false;
    if (!(xs[0] > 0.f) || !std::isfinite(xs[0]) || !std::isfinite(xz[0])) { c->err = "W8A8 HG layer " + layer + ": bad x_scale / x_zero"; return false; }
    q.scale = xs[0];
    q.kf = -(double)xz[0] / (double)xs[0];
    q.k = (int)std::nearbyint(q.kf);
    q.integer = std::fabs(q.kf - q.k) <= 1e-3 && q.k >= 0 && q.k <= 255;
    if (q.integer) q.kf = q.k;
    return true;
}
bool pack_conv_i8(hdrtv_ctx *c, const Pack &pk, const std::string &key, const std::string &wname, int co, int ci, int ks,
                  const std::string &bn_name, int ps_cps, const ActQ &out, bool relu)
{
    std::vector<int8_t> w;
    std::vector<float> ws, b;
    ActQ in;
    if (!pk.get_i8(wname + ".weight_int8", (size_t)co * ci * ks * ks, w, c->err) || !pk.get(wname + ".w_scale", co, ws, c->err) ||
        !pk.get(wname + ".bias", co, b, c->err) || !read_actq(c, pk, wname, in))
        return false;
    const int coP = (co + 127) / 128 * 128;          // conv9: 64 real output channels in a 128-wide tile
    const bool c64 = ci == 64 && ks == 3;             // pixel-pair rows: 6 row-taps of 128 bytes (conv3x3_pglds_i8.hip, C64)
    if ((ci % 128 && !c64) || (co % 128 && ks != 1)) { c->err = "unsupported W8A8 conv shape: " + wname; return false; }
    std::vector<float> g, be, mu, var;
    const bool has_bn = !bn_name.empty();
    if (has_bn) {
        if (!pk.getw(bn_name, co, g, c->err) || !pk.get(bn_name + ".bias", co, be, c->err) ||
            !pk.get(bn_name + ".running_mean", co, mu, c->err) || !pk.get(bn_name + ".running_var", co, var, c->err))
            return false;
    }
    const int nch = c64 ? 1 : ci / 128, taps = ks * ks;
    std::vector<int8_t> wp((size_t)(c64 ? 6 : taps) * nch * coP * 128, (int8_t)0);
    std::vector<float> scale(coP, 0.f), shift(coP, 0.f), delta;
    std::vector<int> delta_acc;
    // Out-of-image halo pixels are ZEROS (what an LDS-DMA lane outside its buffer resource writes), i.e. code 0 = the value
    // x_scale * (128 - k), not 0.0: a 3x3 layer takes the padded taps' share back out through a per-channel constant for each of
    // the 16 border classes.  k = 128 needs none.  (Round 2 staged a line of code k - 128 for integer zero points instead.)
    const bool need_delta = ks == 3 && in.kf != 128.0;
    if (need_delta) { delta.assign((size_t)16 * coP, 0.f); delta_acc.assign((size_t)16 * coP, 0); }
    for (int np = 0; np < co; ++np) {
        const int n = ps_cps > 0 ? 4 * (np % ps_cps) + np / ps_cps : np;
        long wsum = 0;
        long tsum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < ci; ++k)
            for (int tap = 0; tap < taps; ++tap) {
                const int8_t v = w[((size_t)n * ci + k) * taps + tap];
                wsum += v;
                tsum[tap] += v;
                if (c64) {      // row-tap (ky, 0) = [w(ky,0) | w(ky,1)], row-tap (ky, 2) = [w(ky,2) | 0]
                    const int ky = tap / 3, kx = tap % 3;
                    wp[((size_t)(ky * 2 + (kx == 2)) * coP + np) * 128 + (kx == 1 ? 64 : 0) + k] = v;
                } else {
                    wp[(((size_t)tap * nch + k / 128) * coP + np) * 128 + k % 128] = v;
                }
            }
        const double a = (double)in.scale * (double)ws[n];
        double sc = a, sh = a * (128.0 - in.kf) * (double)wsum + (double)b[n], gs = 1.0;
        if (has_bn) {
            gs = (double)g[n] / std::sqrt((double)var[n] + 1e-5);
            sc *= gs;
            sh = (sh - (double)mu[n]) * gs + (double)be[n];
        }
        if (out.scale > 0.f) {      // to the codes (q - 128) of the consumer's quantiser
            sc /= (double)out.scale;
            sh = sh / (double)out.scale + out.kf - 128.0;
            gs /= (double)out.scale;
        }
        scale[np] = (float)sc;
        shift[np] = (float)sh;
        if (need_delta)             // padded taps hold code 0 = value scale * (128 - kf), not 0.0: take their share back out
            for (int cls = 1; cls < 16; ++cls) {
                const int cy = cls >> 2, cx = cls & 3;
                long miss = 0;
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap % 3;
                    if ((ky == 0 && (cy & 1)) || (ky == 2 && (cy & 2)) || (kx == 0 && (cx & 1)) || (kx == 2 && (cx & 2))) miss += tsum[tap];
                }
                delta[(size_t)cls * coP + np] = (float)(-a * (128.0 - in.kf) * (double)miss * gs);
                delta_acc[(size_t)cls * coP + np] = (int)std::nearbyint(-(128.0 - in.kf) * (double)miss);
            }
    }
    std::vector<int8_t> pad(128, (int8_t)0);
    ConvI8Layer L;
    if (need_delta) {
        L.delta = c->wts.put(delta.data(), delta.size() * 4);
        L.delta_acc = c->wts.put(delta_acc.data(), delta_acc.size() * 4);
        L.has_delta = true;
    }
    // behind a ReLU the smallest value is 0.0, whose code is above the bottom of the range when the reader's x_zero < 0
    if (relu && out.scale > 0.f) L.lo_clamp = (float)(std::min(255.0, std::max(0.0, std::nearbyint(out.kf))) - 128.0);
    L.cin = ci; L.cout = coP; L.cout_real = co; L.ks = ks; L.out_f16 = out.scale > 0.f ? 0 : 1;
    L.wpk = c->wts.put(wp.data(), wp.size());
    L.scale = c->wts.put(scale.data(), scale.size() * 4);
    L.shift = c->wts.put(shift.data(), shift.size() * 4);
    L.padline = c->wts.put(pad.data(), pad.size());
    c->conv8[key] = L;
    return true;
}


// ---------------------------------------------------------------- W8A8 layers of the HR network (AGCM + LE)
bool read_actqf(hdrtv_ctx *c, const Pack &pk, const std::string &layer, ActQf &q)
{
    std::vector<float> xs, xz;
    if (!pk.get(layer + ".x_scale", 1, xs, c->err)) return false;
    q.scale = xs[0];
    q.asym = pk.has(layer + ".x_zero");
    q.zero = 0.f;
    if (q.asym) {
        if (!pk.get(layer + ".x_zero", 1, xz, c->err)) return false;
        q.zero = xz[0];
    }
    if (!(q.scale > 0.f) || !std::isfinite(q.scale) || !std::isfinite(q.zero)) { c->err = "bad activation quantiser for " + layer; return false; }
    return true;
}
struct QRaw { std::vector<int8_t> w; std::vector<float> ws, b; ActQf q; };
bool read_qraw(hdrtv_ctx *c, const Pack &pk, const std::string &layer, int co, size_t per_co, QRaw &r)
{
    return pk.get_i8(layer + ".weight_int8", (size_t)co * per_co, r.w, c->err) && pk.get(layer + ".w_scale", co, r.ws, c->err) &&
           pk.get(layer + ".bias", co, r.b, c->err) && read_actqf(c, pk, layer, r.q);
}
// y[n] = x_scale * w_scale[n] * acc + w_scale[n] * (128 x_scale + x_zero) * sum(w_int8[n] over the in-image taps) + bias[n]:
// scale[coP] and shift[16][coP], class = (rows: bit0 first kernel row outside, bit1 last) << 2 | (columns likewise).
// Packed row np holds original output channel rowmap[np].
void q_tables(const QRaw &r, int co, int ci, int ks, int coP, const std::vector<int> &rowmap, std::vector<float> &scale,
              std::vector<float> &shift)
{
    scale.assign(coP, 0.f);
    shift.assign((size_t)16 * coP, 0.f);
    const int taps = ks * ks;
    for (int np = 0; np < co; ++np) {
        const int n = rowmap[np];
        std::vector<long> tsum(taps, 0);
        for (int k = 0; k < ci; ++k)
            for (int t = 0; t < taps; ++t) tsum[t] += r.w[((size_t)n * ci + k) * taps + t];
        scale[np] = (float)((double)r.q.scale * (double)r.ws[n]);
        for (int cls = 0; cls < 16; ++cls) {
            const int cy = cls >> 2, cx = cls & 3;
            long sum = 0;
            for (int t = 0; t < taps; ++t) {
                const int ky = t / ks, kx = t % ks;
                const bool miss = ks == 3 && ((ky == 0 && (cy & 1)) || (ky == 2 && (cy & 2)) || (kx == 0 && (cx & 1)) || (kx == 2 && (cx & 2)));
                if (!miss) sum += tsum[t];
            }
            shift[(size_t)cls * coP + np] = (float)((double)r.ws[n] * r.q.soff() * (double)sum + (double)r.b[n]);
        }
    }
}
// 3x3 / stride 1 / 32 input channels -> conv32p<.., i8>: wpk8 [9][coP][32], byte 16h + 4qd + k of a row = input channel
// 8qd + 4h + k (the order in which conv32p's per-tile pass produces a pixel's codes); ps: PixelShuffle row permutation
bool pack_conv32_i8(hdrtv_ctx *c, const Pack &pk, const std::string &key, int co, int ps_cps)
{
    QRaw r;
    if (!read_qraw(c, pk, key, co, 32 * 9, r)) return false;
    const int coP = (co + 31) / 32 * 32;
    std::vector<int> rowmap(co);
    for (int np = 0; np < co; ++np) rowmap[np] = ps_cps > 0 ? 4 * (np % ps_cps) + np / ps_cps : np;
    std::vector<int8_t> wp((size_t)9 * coP * 32, (int8_t)0);
    for (int np = 0; np < co; ++np)
        for (int tap = 0; tap < 9; ++tap)
            for (int h = 0; h < 2; ++h)
                for (int qd = 0; qd < 4; ++qd)
                    for (int k = 0; k < 4; ++k)
                        wp[((size_t)tap * coP + np) * 32 + 16 * h + 4 * qd + k] = r.w[((size_t)rowmap[np] * 32 + 8 * qd + 4 * h + k) * 9 + tap];
    std::vector<float> scale, shift;
    q_tables(r, co, 32, 3, coP, rowmap, scale, shift);
    QLayer L;
    L.q = r.q; L.cin = 32; L.cout = co; L.coutPad = coP; L.ks = 3; L.stride = 1;
    L.wpk8 = c->wts.put(wp.data(), wp.size());
    L.scale = c->wts.put(scale.data(), scale.size() * 4);
    L.shift = c->wts.put(shift.data(), shift.size() * 4);
    c->q32[key] = L;
    return true;
}
// any other W8A8 LE conv -> conv_q8: wpk8 [ks*ks][coP][ci], natural channel order
bool pack_conv_q8(hdrtv_ctx *c, const Pack &pk, const std::string &key, int co, int ci, int ks, int stride, int ci_real = 0)
{
    QRaw r;
    if (ci_real && ci_real != ci) {           // fewer real input channels than the kernel's 32-byte pixel: zero weights for the rest
        QRaw s;
        if (!read_qraw(c, pk, key, co, (size_t)ci_real * ks * ks, s)) return false;
        r = s;
        r.w.assign((size_t)co * ci * ks * ks, (int8_t)0);
        for (int n = 0; n < co; ++n)
            for (int k = 0; k < ci_real; ++k)
                for (int t = 0; t < ks * ks; ++t) r.w[((size_t)n * ci + k) * ks * ks + t] = s.w[((size_t)n * ci_real + k) * ks * ks + t];
    } else if (!read_qraw(c, pk, key, co, (size_t)ci * ks * ks, r)) {
        return false;
    }
    const int coP = (co + 31) / 32 * 32, taps = ks * ks;
    std::vector<int> rowmap(co);
    for (int np = 0; np < co; ++np) rowmap[np] = np;
    std::vector<int8_t> wp((size_t)taps * coP * ci, (int8_t)0);
    for (int n = 0; n < co; ++n)
        for (int k = 0; k < ci; ++k)
            for (int t = 0; t < taps; ++t) wp[((size_t)t * coP + n) * ci + k] = r.w[((size_t)n * ci + k) * taps + t];
    std::vector<float> scale, shift;
    q_tables(r, co, ci, ks, coP, rowmap, scale, shift);
    QLayer L;
    L.q = r.q; L.cin = ci; L.cout = co; L.coutPad = coP; L.ks = ks; L.stride = stride;
    L.wpk8 = c->wts.put(wp.data(), wp.size());
    L.scale = c->wts.put(scale.data(), scale.size() * 4);
    L.shift = c->wts.put(shift.data(), shift.size() * 4);
    c->q8[key] = L;
    return true;
}
// 1x1 64 -> 16 as the W8A8 last layer of a fused chain (le_fused.hip qlast_apply): two int8 A fragments whose byte j of
// lane (row, lh), MFMA m, is input channel 16s + (e < 4 ? 4lh + e : 8 + 4lh + e - 4) with s = 2m + j / 8, e = j % 8
bool pack_q_last(hdrtv_ctx *c, const Pack &pk, const std::string &layer, QLastLayer &out)
{
    QRaw r;
    if (!read_qraw(c, pk, layer, 16, 64, r)) return false;
    std::vector<int8_t> fr((size_t)2 * 64 * 16, (int8_t)0);
    for (int m = 0; m < 2; ++m)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 16; ++j) {
                const int row = lane & 31, lh = lane >> 5, s = 2 * m + j / 8, e = j % 8;
                const int ch = 16 * s + (e < 4 ? 4 * lh + e : 8 + 4 * lh + e - 4);
                if (row < 16) fr[((size_t)m * 64 + lane) * 16 + j] = r.w[(size_t)row * 64 + ch];
            }
    std::vector<int> rowmap(16);
    for (int i = 0; i < 16; ++i) rowmap[i] = i;
    std::vector<float> scale, shift;
    q_tables(r, 16, 64, 1, 32, rowmap, scale, shift);
    std::vector<float> ss(64, 0.f);
    for (int i = 0; i < 32; ++i) { ss[i] = scale[i]; ss[32 + i] = shift[i]; }
    out.q = r.q;
    out.wq = c->wts.put(fr.data(), fr.size());
    out.ss = c->wts.put(ss.data(), ss.size() * 4);
    out.on = true;
    return true;
}


// ---- fully quantised chains (le_chain_q8.hip)
// 1x1 layer [co][ci] -> int8 A fragments [co/32 (>= 1)][ci/32][64 lanes][16]: byte j of lane (row, lh), K-step kb, is input channel
// 32kb + 8(j/4) + 4lh + j%4 when the operand is the previous layer's accumulator tile (chained), 32kb + 16lh + j when it is read
// from an NHWC int8 tensor (natural)
void chain_frags(const QRaw &r, int co, int ci, bool chained, std::vector<int8_t> &out)
{
    const int nmt = (co + 31) / 32, nkb = ci / 32;
    const size_t base = out.size();
    out.resize(base + (size_t)nmt * nkb * 64 * 16, (int8_t)0);
    for (int mt = 0; mt < nmt; ++mt)
        for (int kb = 0; kb < nkb; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 16; ++j) {
                    const int row = 32 * mt + (lane & 31), lh = lane >> 5;
                    const int ch = 32 * kb + (chained ? 8 * (j >> 2) + 4 * lh + (j & 3) : 16 * lh + j);
                    if (row < co) out[base + (((size_t)mt * nkb + kb) * 64 + lane) * 16 + j] = r.w[(size_t)row * ci + ch];
                }
}
// dequantisation constants of a chained layer in register order [mt][lh][A16 | B16]; inv_next = 1 / x_scale of the layer that
// reads the result as codes (1 = keep real units); per_co = weights per output channel (taps included)
void chain_consts(const QRaw &r, int co, size_t per_co, double inv_next, std::vector<float> &out)
{
    const int nmt = (co + 31) / 32;
    for (int mt = 0; mt < nmt; ++mt)
        for (int lh = 0; lh < 2; ++lh)
            for (int ab = 0; ab < 2; ++ab)
                for (int j = 0; j < 16; ++j) {
                    const int row = 32 * mt + 8 * (j >> 2) + 4 * lh + (j & 3);
                    double v = 0.0;
                    if (row < co) {
                        long sum = 0;
                        for (size_t k = 0; k < per_co; ++k) sum += r.w[(size_t)row * per_co + k];
                        v = ab ? ((double)r.ws[row] * r.q.soff() * (double)sum + (double)r.b[row]) * inv_next
                               : (double)r.q.scale * (double)r.ws[row] * inv_next;
                    }
                    out.push_back((float)v);
                }
}
bool pack_trunk_q8(hdrtv_ctx *c, const Pack &pk)
{
    const char *names[6] = {"LE.cond_first.0", "LE.cond_first.2", "LE.cond_first.4", "LE.CondNet1.0", "LE.CondNet1.2", "LE.CondNet1.4"};
    QRaw r[6];
    for (int l = 0; l < 6; ++l)
        if (!read_qraw(c, pk, names[l], l == 5 ? 16 : 64, l == 0 ? 27 : 64, r[l])) return false;
    std::vector<int8_t> fr;
    // layer 1: k = (ky*3+kx)*3 + c = 16 lh + j
    fr.resize((size_t)2 * 64 * 16, (int8_t)0);
    for (int mt = 0; mt < 2; ++mt)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 16; ++j) {
                const int row = 32 * mt + (lane & 31), k = 16 * (lane >> 5) + j;
                if (k < 27) fr[((size_t)mt * 64 + lane) * 16 + j] = r[0].w[((size_t)row * 3 + k % 3) * 9 + k / 3];
            }
    for (int l = 1; l < 6; ++l) chain_frags(r[l], l == 5 ? 16 : 64, 64, true, fr);
    std::vector<float> K;
    const double inv2 = 1.0 / r[1].q.scale;
    // L1 A [mt][lh][16], then B [cls][mt][lh][16] from the border-class table
    std::vector<int> rowmap(64);
    for (int i = 0; i < 64; ++i) rowmap[i] = i;
    std::vector<float> sc, sh;
    q_tables(r[0], 64, 3, 3, 64, rowmap, sc, sh);
    for (int mt = 0; mt < 2; ++mt)
        for (int lh = 0; lh < 2; ++lh)
            for (int j = 0; j < 16; ++j) K.push_back((float)((double)sc[32 * mt + 8 * (j >> 2) + 4 * lh + (j & 3)] * inv2));
    for (int cls = 0; cls < 16; ++cls)
        for (int mt = 0; mt < 2; ++mt)
            for (int lh = 0; lh < 2; ++lh)
                for (int j = 0; j < 16; ++j) K.push_back((float)((double)sh[(size_t)cls * 64 + 32 * mt + 8 * (j >> 2) + 4 * lh + (j & 3)] * inv2));
    chain_consts(r[1], 64, 64, 1.0 / r[2].q.scale, K);
    chain_consts(r[2], 64, 64, 1.0, K);                     // `cond` is stored in real units (f16)
    chain_consts(r[3], 64, 64, 1.0 / r[4].q.scale, K);
    chain_consts(r[4], 64, 64, 1.0 / r[5].q.scale, K);
    chain_consts(r[5], 16, 64, 1.0, K);
    if (K.size() != 1664 || fr.size() != (size_t)20 * 1024) { c->err = "internal: trunk_q8 pack size"; return false; }
    for (int l = 0; l < 6; ++l) c->tq_q[l] = r[l].q;
    c->tq_frag = c->wts.put(fr.data(), fr.size());
    c->tq_const = c->wts.put(K.data(), K.size() * 4);
    c->trunk_q8 = true;
    return true;
}
bool pack_tail_q8(hdrtv_ctx *c, const Pack &pk)
{
    QRaw r[2];
    if (!read_qraw(c, pk, "LE.CondNet2.2", 64, 64, r[0]) || !read_qraw(c, pk, "LE.CondNet2.4", 16, 64, r[1])) return false;
    std::vector<int8_t> fr;
    chain_frags(r[0], 64, 64, false, fr);
    chain_frags(r[1], 16, 64, true, fr);
    std::vector<float> K;
    chain_consts(r[0], 64, 64, 1.0 / r[1].q.scale, K);
    chain_consts(r[1], 16, 64, 1.0, K);
    c->tl_q[0] = r[0].q; c->tl_q[1] = r[1].q;
    c->tl_frag = c->wts.put(fr.data(), fr.size());
    c->tl_const = c->wts.put(K.data(), K.size() * 4);
    c->tail_q8 = true;
    return true;
}
bool pack_agcm_q8(hdrtv_ctx *c, const Pack &pk)
{
    QRaw r[3];
    if (!read_qraw(c, pk, "AGCM.conv_first", 64, 3, r[0]) || !read_qraw(c, pk, "AGCM.HRconv", 64, 64, r[1]) ||
        !read_qraw(c, pk, "AGCM.conv_last", 3, 64, r[2]))
        return false;
    std::vector<int8_t> fr((size_t)2 * 64 * 16, (int8_t)0);
    for (int mt = 0; mt < 2; ++mt)
        for (int lane = 0; lane < 32; ++lane)            // lane half 0 only: bytes 0..2 = colour channels
            for (int j = 0; j < 3; ++j) fr[((size_t)mt * 64 + lane) * 16 + j] = r[0].w[(size_t)(32 * mt + lane) * 3 + j];
    chain_frags(r[1], 64, 64, true, fr);
    chain_frags(r[2], 3, 64, true, fr);
    std::vector<float> P(192, 0.f), Q(192, 0.f);
    const int co[3] = {64, 64, 3}, ci[3] = {3, 64, 64};
    for (int l = 0; l < 3; ++l)
        for (int m = 0; m < co[l]; ++m) {
            long sum = 0;
            for (int k = 0; k < ci[l]; ++k) sum += r[l].w[(size_t)m * ci[l] + k];
            P[l * 64 + m] = (float)((double)r[l].q.scale * r[l].ws[m]);
            Q[l * 64 + m] = (float)((double)r[l].ws[m] * r[l].q.soff() * (double)sum + (double)r[l].b[m]);
        }
    for (int l = 0; l < 3; ++l) c->ag_q[l] = r[l].q;
    c->ag_frag = c->wts.put(fr.data(), fr.size());
    c->ag_P = c->wts.put(P.data(), P.size() * 4);
    c->ag_Q = c->wts.put(Q.data(), Q.size() * 4);
    c->agcm_q8 = true;
    return true;
}
bool read_fakeq(hdrtv_ctx *c, const Pack &pk, const std::string &layer, FakeQ &f)
{
    f = FakeQ{0, 0.f, 0.f, 0.f, 0.f};
    if (!pk.is_w8a8(layer)) return true;
    ActQf q;
    if (!read_actqf(c, pk, layer, q)) return false;
    f.on = 1; f.inv = q.inv(); f.zoff = q.zoff(); f.scale = q.scale; f.zero = q.asym ? q.zero : -128.f * q.scale;
    return true;
}

// A-fragment element of the 3-channel 3x3 convs (le_hg_misc.hip): k-step ky, lane half lh, slot j = pixel kx = 2 lh + j / 4,
// channel j % 4; the 4th pixel and the 4th channel are padding
static inline float c3_welem(const std::vector<float> &w, int m, int ky, int lh, int j, const float *bias_k = nullptr)
{
    const int kx = 2 * lh + (j >> 2), ch = j & 3;
    if (bias_k && ky == 1 && kx == 1 && ch == 3) return bias_k[m];      // the staged pixels' 4th channel is 1 (le_hg_misc.hip)
    return (kx < 3 && ch < 3) ? w[((size_t)m * 3 + ch) * 9 + ky * 3 + kx] : 0.f;
}

// 3x3 conv from 3 planar channels: A fragments [MT][3 kernel rows][64 lanes][8]
bool pack_c3(hdrtv_ctx *c, const Pack &pk, const std::string &key, const std::string &wname, int co, const std::string &bn_name)
{
    std::vector<float> w, b;
    if (!pk.getw(wname, (size_t)co * 27, w, c->err) || !pk.get(wname + ".bias", co, b, c->err)) return false;
    std::vector<float> scale(co, 1.f), shift(co, 0.f), g, be, mu, var;
    if (!bn_name.empty()) {
        if (!pk.getw(bn_name, co, g, c->err) || !pk.get(bn_name + ".bias", co, be, c->err) ||
            !pk.get(bn_name + ".running_mean", co, mu, c->err) || !pk.get(bn_name + ".running_var", co, var, c->err))
            return false;
    }
    for (int n = 0; n < co; ++n) {
        if (!bn_name.empty()) {
            const float s = g[n] / std::sqrt(var[n] + 1e-5f);
            scale[n] = s;
            shift[n] = (b[n] - mu[n]) * s + be[n];
        }                                  // no BatchNorm: the bias rides in the K axis (c3_welem), scale 1 and shift 0
    }
    const float *bias_k = bn_name.empty() ? b.data() : nullptr;
    const int mt = co / 32;
    std::vector<f16> fr((size_t)mt * 3 * 64 * 8, (f16)0.f);
    for (int i = 0; i < mt; ++i)
        for (int ky = 0; ky < 3; ++ky)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    fr[(((size_t)i * 3 + ky) * 64 + lane) * 8 + j] = (f16)c3_welem(w, i * 32 + (lane & 31), ky, lane >> 5, j, bias_k);
    C3Layer L;
    L.cout = co;
    L.wfrag = c->wts.put(fr.data(), fr.size() * sizeof(f16));
    L.scale = c->wts.put(scale.data(), co * 4);
    L.shift = c->wts.put(shift.data(), co * 4);
    c->c3[key] = L;
    return true;
}

// LE.conv_first as a W8A8 layer -> conv_c3_q8 (le_hg_misc.hip): A fragments [2 MFMAs][64 lanes][16 bytes], byte j of lane
// (row n, half lh) = weight of pixel kx = j / 4, channel j % 4 in kernel row ky = lh (first MFMA) / 2 (second, lh = 0 only)
bool pack_c3_q8(hdrtv_ctx *c, const Pack &pk, const std::string &key)
{
    QRaw r;
    if (!read_qraw(c, pk, key, 32, 27, r)) return false;
    std::vector<int8_t> fr((size_t)2 * 64 * 16, (int8_t)0);
    for (int m = 0; m < 2; ++m)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 16; ++j) {
                const int n = lane & 31, lh = lane >> 5, kx = j >> 2, ch = j & 3;
                const int ky = m == 0 ? lh : (lh == 0 ? 2 : -1);
                if (ky >= 0 && kx < 3 && ch < 3) fr[((size_t)m * 64 + lane) * 16 + j] = r.w[((size_t)n * 3 + ch) * 9 + ky * 3 + kx];
            }
    std::vector<int> rowmap(32);
    for (int i = 0; i < 32; ++i) rowmap[i] = i;
    std::vector<float> scale, shift;
    q_tables(r, 32, 3, 3, 32, rowmap, scale, shift);
    QLayer L;
    L.q = r.q; L.cin = 3; L.cout = 32; L.coutPad = 32; L.ks = 3; L.stride = 1;
    L.wpk8 = c->wts.put(fr.data(), fr.size());
    L.scale = c->wts.put(scale.data(), scale.size() * 4);
    L.shift = c->wts.put(shift.data(), shift.size() * 4);
    c->q8[key + "#c3"] = L;
    return true;
}

// SFTLayer: three A fragments (hidden stack natural-k; scale/shift heads k-permuted) + 96 biases
bool pack_sft(hdrtv_ctx *c, const Pack &pk, const std::string &key, const std::string &name)
{
    std::vector<float> w0s, b0s, w1s, b1s, w0t, b0t, w1t, b1t;
    if (!pk.getw(name + ".SFT_scale_conv0", 256, w0s, c->err) || !pk.get(name + ".SFT_scale_conv0.bias", 16, b0s, c->err) ||
        !pk.getw(name + ".SFT_scale_conv1", 512, w1s, c->err) || !pk.get(name + ".SFT_scale_conv1.bias", 32, b1s, c->err) ||
        !pk.getw(name + ".SFT_shift_conv0", 256, w0t, c->err) || !pk.get(name + ".SFT_shift_conv0.bias", 16, b0t, c->err) ||
        !pk.getw(name + ".SFT_shift_conv1", 512, w1t, c->err) || !pk.get(name + ".SFT_shift_conv1.bias", 32, b1t, c->err))
        return false;
    std::vector<f16> fr(3 * 64 * 8);
    std::vector<float> bias(96);
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            const int m = lane & 31, p = 8 * (lane >> 5) + j;
            fr[(0 * 64 + lane) * 8 + j] = (f16)(m < 16 ? w0s[m * 16 + p] : w0t[(m - 16) * 16 + p]);
            fr[(1 * 64 + lane) * 8 + j] = (f16)w1s[m * 16 + acc_kperm16(p)];
            fr[(2 * 64 + lane) * 8 + j] = (f16)w1t[m * 16 + acc_kperm16(p)];
        }
    for (int i = 0; i < 16; ++i) { bias[i] = b0s[i]; bias[16 + i] = b0t[i]; }
    for (int i = 0; i < 32; ++i) { bias[32 + i] = b1s[i]; bias[64 + i] = b1t[i]; }
    SftLayer L;
    L.wfrag = c->wts.put(fr.data(), fr.size() * sizeof(f16));
    L.bias = c->wts.put(bias.data(), bias.size() * 4);
    const char *cv[4] = {".SFT_scale_conv0", ".SFT_shift_conv0", ".SFT_scale_conv1", ".SFT_shift_conv1"};
    int nq = 0;
    for (const char *n : cv) nq += pk.is_w8a8(name + n) ? 1 : 0;
    if (nq != 0 && nq != 4) { c->err = "SFT layer " + name + ": W8A8 on some of its four convs only is not supported"; return false; }
    if (nq == 4) {
        // conv32p's SQ path (conv32p.hip): fragment / constant layouts documented there and in common.h Conv32Params
        QRaw r[4];
        for (int i = 0; i < 4; ++i)
            if (!read_qraw(c, pk, name + cv[i], i < 2 ? 16 : 32, 16, r[i])) return false;
        std::vector<int8_t> qf((size_t)3 * 64 * 16, (int8_t)0);
        std::vector<float> K(192, 0.f);
        auto wsum = [](const QRaw &q, int row) { long t = 0; for (int k = 0; k < 16; ++k) t += q.w[row * 16 + k]; return (double)t; };
        for (int lane = 0; lane < 64; ++lane) {
            const int row = lane & 31, lh = lane >> 5;
            for (int j = 0; j < 16; ++j) {
                if (row < 16 && lh == 0) qf[((size_t)0 * 64 + lane) * 16 + j] = r[0].w[row * 16 + j];
                if (row >= 16 && lh == 1) qf[((size_t)0 * 64 + lane) * 16 + j] = r[1].w[(row - 16) * 16 + j];
                if (j < 8) {
                    const int hid = (j < 4 ? 4 * lh + j : 8 + 4 * lh + j - 4);
                    qf[((size_t)1 * 64 + lane) * 16 + j] = r[2].w[row * 16 + hid];
                    qf[((size_t)2 * 64 + lane) * 16 + j] = r[3].w[row * 16 + hid];
                }
            }
        }
        for (int lh = 0; lh < 2; ++lh)
            for (int j = 0; j < 16; ++j) {
                const int row = 8 * (j >> 2) + 4 * lh + (j & 3);
                const int br = row >> 4, idx = row & 15;          // hidden row: branch 0 scale / 1 shift
                const QRaw &h = r[br], &o = r[2 + br];
                const double inv1 = 1.0 / (double)o.q.scale;
                K[0 * 32 + lh * 16 + j] = (float)((double)h.q.scale * h.ws[idx] * inv1);
                K[1 * 32 + lh * 16 + j] = (float)(((double)h.ws[idx] * h.q.soff() * wsum(h, idx) + (double)h.b[idx]) * inv1);
                for (int b = 0; b < 2; ++b) {                     // second layers: output channel = row
                    const QRaw &q = r[2 + b];
                    K[(2 + 2 * b) * 32 + lh * 16 + j] = (float)((double)q.q.scale * q.ws[row]);
                    K[(3 + 2 * b) * 32 + lh * 16 + j] = (float)((double)q.ws[row] * q.q.soff() * wsum(q, row) + (double)q.b[row] + (b == 0 ? 1.0 : 0.0));
                }
            }
        L.q = true;
        for (int i = 0; i < 4; ++i) L.fq[i] = r[i].q;
        L.qfrag = c->wts.put(qf.data(), qf.size());
        L.qconst = c->wts.put(K.data(), K.size() * 4);
        for (int b = 0; b < 2; ++b) { L.inv[b] = r[b].q.inv(); L.zoff[b] = r[b].q.zoff(); L.hzoff[b] = r[2 + b].q.zoff(); }
    }
    c->sft[key] = L;
    return true;
}

// Fused LE condition trunk: cond_first.{0,2,4} + CondNet1.{0,2,4} as 40 A fragments + 352 biases
bool pack_cond_trunk(hdrtv_ctx *c, const Pack &pk)
{
    const char *names[6] = {"LE.cond_first.0", "LE.cond_first.2", "LE.cond_first.4", "LE.CondNet1.0", "LE.CondNet1.2", "LE.CondNet1.4"};
    std::vector<f16> fr((size_t)40 * 64 * 8, (f16)0.f);
    std::vector<float> bias(64 * 5 + 32, 0.f);
    std::vector<float> w, b;
    // layer 1: 3x3 from 3 channels, natural k = (ky*3+kx)*3 + c
    if (!pk.getw(std::string(names[0]), 64 * 27, w, c->err) || !pk.get(std::string(names[0]) + ".bias", 64, b, c->err))
        return false;
    for (int i = 0; i < 64; ++i) bias[i] = b[i];
    for (int mt = 0; mt < 2; ++mt)
        for (int ks = 0; ks < 2; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int m = mt * 32 + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
                    if (k < 27) fr[(((size_t)mt * 2 + ks) * 64 + lane) * 8 + j] = (f16)w[((size_t)m * 3 + k % 3) * 9 + k / 3];
                }
    // layers 2..5: 64x64 1x1, k permuted (operand is the previous accumulator)
    for (int l = 2; l <= 5; ++l) {
        if (!pk.getw(std::string(names[l - 1]), 64 * 64, w, c->err) || !pk.get(std::string(names[l - 1]) + ".bias", 64, b, c->err))
            return false;
        for (int i = 0; i < 64; ++i) bias[64 * (l - 1) + i] = b[i];
        for (int mt = 0; mt < 2; ++mt)
            for (int sidx = 0; sidx < 4; ++sidx)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int m = mt * 32 + (lane & 31), k = 16 * sidx + acc_kperm16(8 * (lane >> 5) + j);
                        fr[((size_t)(4 + (l - 2) * 8 + mt * 4 + sidx) * 64 + lane) * 8 + j] = (f16)w[(size_t)m * 64 + k];
                    }
    }
    // layer 6: 16x64, rows 16..31 zero
    if (!pk.getw(std::string(names[5]), 16 * 64, w, c->err) || !pk.get(std::string(names[5]) + ".bias", 16, b, c->err)) return false;
    for (int i = 0; i < 16; ++i) bias[320 + i] = b[i];
    for (int sidx = 0; sidx < 4; ++sidx)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int m = lane & 31, k = 16 * sidx + acc_kperm16(8 * (lane >> 5) + j);
                if (m < 16) fr[((size_t)(36 + sidx) * 64 + lane) * 8 + j] = (f16)w[(size_t)m * 64 + k];
            }
    c->trunk_wfrag = c->wts.put(fr.data(), fr.size() * sizeof(f16));
    c->trunk_bias = c->wts.put(bias.data(), bias.size() * 4);
    return true;
}

// CondNet2's tail (1x1 64->64, LeakyReLU, 1x1 64->16) for cond_tail_kernel: 12 A fragments + 96 biases
bool pack_cond_tail(hdrtv_ctx *c, const Pack &pk, const std::string &l1, const std::string &l2)
{
    std::vector<float> w1, b1, w2, b2;
    if (!pk.getw(l1, 64 * 64, w1, c->err) || !pk.get(l1 + ".bias", 64, b1, c->err) ||
        !pk.getw(l2, 16 * 64, w2, c->err) || !pk.get(l2 + ".bias", 16, b2, c->err))
        return false;
    std::vector<f16> fr((size_t)12 * 64 * 8, (f16)0.f);
    std::vector<float> bias(96, 0.f);
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
            const int m = lane & 31, p = 8 * (lane >> 5) + j;
            for (int sidx = 0; sidx < 4; ++sidx) {
                for (int mt = 0; mt < 2; ++mt)       // layer 1 reads its operand from memory: natural k
                    fr[((size_t)(mt * 4 + sidx) * 64 + lane) * 8 + j] = (f16)w1[(size_t)(mt * 32 + m) * 64 + 16 * sidx + p];
                if (m < 16)                          // layer 2 reads layer 1's accumulator tiles: K-permuted
                    fr[((size_t)(8 + sidx) * 64 + lane) * 8 + j] = (f16)w2[(size_t)m * 64 + 16 * sidx + acc_kperm16(p)];
            }
        }
    for (int i = 0; i < 64; ++i) bias[i] = b1[i];
    for (int i = 0; i < 16; ++i) bias[64 + i] = b2[i];
    c->tail_wfrag = c->wts.put(fr.data(), fr.size() * sizeof(f16));
    c->tail_bias = c->wts.put(bias.data(), bias.size() * 4);
    return true;
}

bool put_f32(hdrtv_ctx *c, const Pack &pk, const std::string &key, const std::string &name, size_t numel)
{
    std::vector<float> v;
    const std::string suffix = ".weight";
    const bool is_w = name.size() > suffix.size() && name.compare(name.size() - suffix.size(), suffix.size(), suffix) == 0;
    const std::string layer = is_w ? name.substr(0, name.size() - suffix.size()) : std::string();
    if (is_w ? !pk.getw(layer, numel, v, c->err, !pk.is_w8a8(layer)) : !pk.get(name, numel, v, c->err)) return false;
    c->f32v[key] = c->wts.put(v.data(), v.size() * 4);
    return true;
}

bool build_weights(hdrtv_ctx *c, const Pack &hr, const Pack *hg)
{
    {
        const std::vector<unsigned char> z(256, 0);
        c->zeros_off = c->wts.put(z.data(), z.size());
        const std::vector<unsigned char> d(8192, 0);
        c->dump_off = c->wts.put(d.data(), d.size());
    }
    // ---- AGCM (fp32 on device: tiny)
    const int cls_ci[5] = {3, 16, 32, 64, 128}, cls_co[5] = {16, 32, 64, 128, 128}, cls_idx[5] = {0, 4, 8, 12, 16};
    char nm[160], key[64];
    for (int i = 0; i < 5; ++i) {
        snprintf(nm, sizeof nm, "AGCM.classifier.model.%d", cls_idx[i]);
        snprintf(key, sizeof key, "cls%d.w", i);
        if (!put_f32(c, hr, key, std::string(nm) + ".weight", (size_t)cls_co[i] * cls_ci[i])) return false;
        snprintf(key, sizeof key, "cls%d.b", i);
        if (!put_f32(c, hr, key, std::string(nm) + ".bias", cls_co[i])) return false;
        if (i < 4) {
            snprintf(nm, sizeof nm, "AGCM.classifier.model.%d", cls_idx[i] + 3);
            snprintf(key, sizeof key, "cls%d.g", i);
            if (!put_f32(c, hr, key, std::string(nm) + ".weight", cls_co[i])) return false;
            snprintf(key, sizeof key, "cls%d.be", i);
            if (!put_f32(c, hr, key, std::string(nm) + ".bias", cls_co[i])) return false;
        }
    }
    if (!put_f32(c, hr, "cls20.w", "AGCM.classifier.model.20.weight", 6 * 128) ||
        !put_f32(c, hr, "cls20.b", "AGCM.classifier.model.20.bias", 6))
        return false;
    const char *stage[3] = {"first", "HR", "last"};
    const int stage_n[3] = {64, 64, 3};
    for (int s = 0; s < 3; ++s) {
        for (int kind = 0; kind < 2; ++kind) {
            snprintf(nm, sizeof nm, "AGCM.cond_%s_%s", kind ? "shift" : "scale", stage[s]);
            snprintf(key, sizeof key, "gfm.%c%d.w", kind ? 't' : 's', s);
            if (!put_f32(c, hr, key, std::string(nm) + ".weight", (size_t)stage_n[s] * 6)) return false;
            snprintf(key, sizeof key, "gfm.%c%d.b", kind ? 't' : 's', s);
            if (!put_f32(c, hr, key, std::string(nm) + ".bias", stage_n[s])) return false;
        }
    }
    if (!put_f32(c, hr, "agcm.w1", "AGCM.conv_first.weight", 192) || !put_f32(c, hr, "agcm.b1", "AGCM.conv_first.bias", 64) ||
        !put_f32(c, hr, "agcm.w2", "AGCM.HRconv.weight", 4096) || !put_f32(c, hr, "agcm.b2", "AGCM.HRconv.bias", 64) ||
        !put_f32(c, hr, "agcm.w3", "AGCM.conv_last.weight", 192) || !put_f32(c, hr, "agcm.b3", "AGCM.conv_last.bias", 3))
        return false;

    {   // W8A8 AGCM layers: classifier convs and Linear heads as fp32 fake-quant, the three GFM convs as an int8 chain
        const int idx6[6] = {0, 4, 8, 12, 16, 20};
        for (int i = 0; i < 6; ++i) {
            snprintf(nm, sizeof nm, "AGCM.classifier.model.%d", idx6[i]);
            if (!read_fakeq(c, hr, nm, c->cls_q[i])) return false;
        }
        const char *lin[6] = {"AGCM.cond_scale_first", "AGCM.cond_scale_HR", "AGCM.cond_scale_last",
                              "AGCM.cond_shift_first", "AGCM.cond_shift_HR", "AGCM.cond_shift_last"};
        bool any_lin = false;
        for (int i = 0; i < 6; ++i) {
            if (!read_fakeq(c, hr, lin[i], c->lin_q[i])) return false;
            any_lin = any_lin || c->lin_q[i].on;
        }
        const int nq = (int)hr.is_w8a8("AGCM.conv_first") + (int)hr.is_w8a8("AGCM.HRconv") + (int)hr.is_w8a8("AGCM.conv_last");
        if (nq == 3) { if (!pack_agcm_q8(c, hr)) return false; }
        else if (nq != 0 || any_lin) { c->err = "W8A8 AGCM: conv_first, HRconv and conv_last must be W8A8 together (Linear heads only with them)"; return false; }
    }

    // ---- LE
    // A W8A8 layer (weight_int8 + x_scale in the pack: the reference's `predequantize` off) runs on int8 MFMA when a
    // kernel exists for it; the pack is rejected otherwise -- there is no silent fake-quant or fp16 substitute.
    auto isq = [&](const std::string &L) { return hr.is_w8a8(L); };
    {
        const std::string suf = ".x_scale";
        for (const auto &kv : hr.e) {
            const std::string &k = kv.first;
            if (k.size() > suf.size() && k.compare(k.size() - suf.size(), suf.size(), suf) == 0) c->hr_i8 = true;
        }
        // the fused chains exist for the combinations the reference's recipes use (Appendix B of SURVEY.md)
        const char *tr[6] = {"LE.cond_first.0", "LE.cond_first.2", "LE.cond_first.4", "LE.CondNet1.0", "LE.CondNet1.2", "LE.CondNet1.4"};
        int ntr = 0;
        for (const char *n : tr) ntr += isq(n) ? 1 : 0;
        if (!(ntr == 0 || ntr == 6 || (ntr == 1 && isq("LE.CondNet1.4")))) {
            c->err = "W8A8 condition trunk: cond_first.{0,2,4} + CondNet1.{0,2,4} must be W8A8 together (or CondNet1.4 alone)";
            return false;
        }
        if (isq("LE.CondNet2.2") && !(isq("LE.CondNet2.4") && isq("LE.CondNet2.0"))) {
            c->err = "W8A8 CondNet2.2 needs W8A8 CondNet2.0 and CondNet2.4 (its input and output are int8 codes in the fused tail)";
            return false;
        }
        if (ntr == 6 && !pack_trunk_q8(c, hr)) return false;
        if (isq("LE.CondNet2.2") && !pack_tail_q8(c, hr)) return false;
    }
    if (!pack_cond_trunk(c, hr) || !pack_cond_tail(c, hr, "LE.CondNet2.2", "LE.CondNet2.4")) return false;
    if (isq("LE.conv_first") ? !(pack_conv_q8(c, hr, "LE.conv_first", 32, 32, 3, 1, 3) && pack_c3_q8(c, hr, "LE.conv_first") &&
                                 pack_c3(c, hr, "le.conv_first#fq", "LE.conv_first", 32, ""))
                             : !pack_c3(c, hr, "le.conv_first", "LE.conv_first", 32, ""))
        return false;
    if (!c->trunk_q8 && isq("LE.CondNet1.4") && !pack_q_last(c, hr, "LE.CondNet1.4", c->q_trunk6)) return false;
    if (!c->tail_q8 && isq("LE.CondNet2.4") && !pack_q_last(c, hr, "LE.CondNet2.4", c->q_tail2)) return false;
    struct Spec { const char *name; int co, ci, ks, stride, ps; };
    const Spec le_convs[] = {
        {"LE.CondNet3.4", 16, 64, 1, 1, 0}, {"LE.CondNet4.4", 16, 64, 3, 2, 0},
        {"LE.HR_conv1", 32, 32, 3, 1, 0}, {"LE.HR_conv2", 32, 32, 3, 1, 0}, {"LE.conv_last", 3, 32, 3, 1, 0},
        {"LE.down_conv1", 32, 32, 3, 2, 0}, {"LE.down_conv2", 32, 32, 3, 2, 0}, {"LE.down_conv3", 32, 32, 3, 2, 0},
        {"LE.up_conv1.0", 128, 32, 3, 1, 32}, {"LE.up_conv2.0", 128, 32, 3, 1, 32}, {"LE.up_conv3.0", 128, 32, 3, 1, 32},
    };
    for (const Spec &s : le_convs) {
        if (isq(s.name)) {
            if (s.ks == 3 && s.stride == 1 ? !pack_conv32_i8(c, hr, s.name, s.co, s.ps) : !pack_conv_q8(c, hr, s.name, s.co, s.ci, s.ks, s.stride))
                return false;
            // ... and, for the fused row kernels (le_rows.hip), its dequantised weights as an fp16 layer "<name>#fq": they apply the
            // layer's activation quantiser in registers and convolve in fp16 -- W8A8Conv2d.forward's own arithmetic
            if (s.ci == 32 && s.ks == 3 && !pack_conv(c, hr, std::string(s.name) + "#fq", s.name, s.co, s.ci, s.ks, s.stride, "", s.ps)) return false;
        } else if (!pack_conv(c, hr, s.name, s.name, s.co, s.ci, s.ks, s.stride, "", s.ps)) {
            return false;
        }
    }
    // stride-2 layers from the 64-channel condition map: CondNet{2,3,4}.0 merged (192 outputs) unless one of them is W8A8
    // (each W8A8 layer quantises the condition map with its own x_scale / x_zero); .2 layers alone
    if (!isq("LE.CondNet2.0") && !isq("LE.CondNet3.0") && !isq("LE.CondNet4.0")) {
        if (!pack_conv(c, hr, "LE.CondNet234.0", "LE.CondNet2.0+LE.CondNet3.0+LE.CondNet4.0", 192, 64, 3, 2, "", 0, 64)) return false;
    } else {
        for (const char *n : {"LE.CondNet2.0", "LE.CondNet3.0", "LE.CondNet4.0"})
            if (isq(n) ? !pack_conv_q8(c, hr, n, 64, 64, 3, 2) : !pack_conv(c, hr, n, n, 64, 64, 3, 2, "", 0, 64)) return false;
    }
    for (const char *n : {"LE.CondNet3.2", "LE.CondNet4.2"})
        if (isq(n) ? !pack_conv_q8(c, hr, n, 64, 64, 3, 2) : !pack_conv(c, hr, n, n, 64, 64, 3, 2, "", 0, 64)) return false;
    const char *trunks[5] = {"recon_trunk1", "recon_trunk2", "recon_trunk3", "recon_trunk4", "recon_trunk5"};
    const int trunk_n[5] = {1, 1, 4, 1, 1};
    for (int t = 0; t < 5; ++t)
        for (int b = 0; b < trunk_n[t]; ++b) {
            snprintf(nm, sizeof nm, "LE.%s.%d", trunks[t], b);
            const std::string base = nm;
            for (const char *cv : {".conv1", ".conv2"}) {
                if (isq(base + cv) ? !pack_conv32_i8(c, hr, base + cv, 32, 0) : !pack_conv(c, hr, base + cv, base + cv, 32, 32, 3, 1, "", 0))
                    return false;
                if (isq(base + cv) && !pack_conv(c, hr, base + cv + "#fq", base + cv, 32, 32, 3, 1, "", 0)) return false;
            }
            if (!pack_sft(c, hr, base + ".sft1", base + ".sft1") || !pack_sft(c, hr, base + ".sft2", base + ".sft2")) return false;
        }
    if (!pack_sft(c, hr, "LE.SFT_layer1", "LE.SFT_layer1") || !pack_sft(c, hr, "LE.SFT_layer2", "LE.SFT_layer2")) return false;

    // ---- HG
    if (hg) {
        if (!pack_c3(c, *hg, "hg.conv1", "conv1.0", 64, "conv1.1")) return false;
        c->hg_i8 = hg->has("conv3_1.0.weight_int8");
        if (!c->hg_i8) {
            const Spec blocks[] = {{"conv2", 128, 64, 3, 1, 0}, {"conv3_1", 256, 128, 3, 1, 0}, {"conv3_2", 256, 256, 3, 1, 0},
                                   {"conv4_1", 512, 256, 3, 1, 0}, {"conv4_2", 512, 512, 3, 1, 0}, {"conv5_1", 512, 512, 3, 1, 0},
                                   {"conv5_2", 512, 512, 3, 1, 0}, {"conv_code1", 512, 512, 3, 1, 0}, {"conv_code2", 512, 512, 3, 1, 0}};
            for (const Spec &s : blocks)
                if (!pack_conv(c, *hg, std::string("hg.") + s.name, std::string(s.name) + ".0", s.co, s.ci, 3, 1,
                               std::string(s.name) + ".1", 0))
                    return false;
            const Spec ups[] = {{"Up_conv1", 2048, 512, 3, 1, 512}, {"Up_conv2", 2048, 512, 3, 1, 512}, {"Up_conv3", 1024, 256, 3, 1, 256},
                                {"Up_conv4", 512, 128, 3, 1, 128}, {"Up_conv5", 256, 64, 3, 1, 64}};
            for (const Spec &s : ups)
                if (!pack_conv(c, *hg, std::string("hg.") + s.name, std::string(s.name) + ".0", s.co, s.ci, 3, 1, "", s.ps)) return false;
            const Spec fuses[] = {{"conv6", 512, 1024, 1, 1, 0}, {"conv7", 256, 1024, 1, 1, 0}, {"conv8", 128, 512, 1, 1, 0},
                                  {"conv9", 64, 256, 1, 1, 0}};
            for (const Spec &s : fuses)
                if (!pack_conv(c, *hg, std::string("hg.") + s.name, s.name, s.co, s.ci, 1, 1, "", 0)) return false;
        } else {
            // W8A8 checkpoint (weights.HG_W8A8_GROUPS): conv2 .. Up_conv5 and the fuse convs conv6..9 on int8 MFMA; conv1,
            // conv10, conv_last stay fp16 (conv1 writes int8 codes of its pooled output, Up_conv5 real-valued partial sums).  A layer's epilogue writes the codes of the layer that
            // reads its output; tensors read by two layers (encoder skip) or concatenated must share one quantiser.
            struct Q8 { const char *name; int co, ci, ks, ps; const char *bn; const char *consumer; const char *shares; };
            const Q8 q8[] = {
                {"conv2", 128, 64, 3, 0, "conv2.1", "conv3_1.0", "conv9"},
                {"conv3_1", 256, 128, 3, 0, "conv3_1.1", "conv3_2.0", nullptr}, {"conv3_2", 256, 256, 3, 0, "conv3_2.1", "conv4_1.0", "conv8"},
                {"conv4_1", 512, 256, 3, 0, "conv4_1.1", "conv4_2.0", nullptr}, {"conv4_2", 512, 512, 3, 0, "conv4_2.1", "conv5_1.0", "conv7"},
                {"conv5_1", 512, 512, 3, 0, "conv5_1.1", "conv5_2.0", nullptr}, {"conv5_2", 512, 512, 3, 0, "conv5_2.1", "conv_code1.0", "conv6"},
                {"conv_code1", 512, 512, 3, 0, "conv_code1.1", "conv_code2.0", nullptr},
                {"conv_code2", 512, 512, 3, 0, "conv_code2.1", "Up_conv1.0", nullptr},
                {"Up_conv1", 2048, 512, 3, 512, "", "conv6", nullptr}, {"conv6", 512, 1024, 1, 0, "", "Up_conv2.0", nullptr},
                {"Up_conv2", 2048, 512, 3, 512, "", "conv7", nullptr}, {"conv7", 256, 1024, 1, 0, "", "Up_conv3.0", nullptr},
                {"Up_conv3", 1024, 256, 3, 256, "", "conv8", nullptr}, {"conv8", 128, 512, 1, 0, "", "Up_conv4.0", nullptr},
                {"Up_conv4", 512, 128, 3, 128, "", "conv9", nullptr}, {"conv9", 64, 256, 1, 0, "", "Up_conv5.0", nullptr},
                {"Up_conv5", 256, 64, 3, 64, "", nullptr, nullptr}};
            for (const Q8 &L : q8) {
                ActQ out;
                if (L.consumer && !read_actq(c, *hg, L.consumer, out)) return false;
                if (L.shares) {
                    ActQ o2;
                    if (!read_actq(c, *hg, L.shares, o2)) return false;
                    if (o2.scale != out.scale || o2.kf != out.kf) {
                        c->err = std::string("W8A8 HG: ") + L.consumer + " and " + L.shares + " read one tensor and must share x_scale / x_zero";
                        return false;
                    }
                }
                const bool relu = L.ks == 3;       // conv blocks and Up blocks end in ReLU
                const std::string wname = L.ks == 3 ? std::string(L.name) + ".0" : std::string(L.name);
                if (!pack_conv_i8(c, *hg, std::string("hg.") + L.name, wname, L.co, L.ci, L.ks, L.bn, L.ps, out, relu)) return false;
            }
            ActQ q0;                       // the fp16 -> int8 boundary: conv1's pooled output, read by conv2
            if (!read_actq(c, *hg, "conv2.0", q0)) return false;
            c->hg_q0_inv = 1.f / q0.scale;
            c->hg_q0_zero = (float)(q0.kf - 128.0);
        }
        if (!put_f32(c, *hg, "hg.w10", "conv10.weight", 3 * 128) || !put_f32(c, *hg, "hg.b10", "conv10.bias", 3) ||
            !put_f32(c, *hg, "hg.wl", "conv_last.weight", 18) || !put_f32(c, *hg, "hg.bl", "conv_last.bias", 3))
            return false;
        {   // fused tail: conv10 = [first 64 inputs: Up_conv5 | last 64 inputs: conv1_out]
            std::vector<float> w10, w1;
            if (!hg->getw("conv10", 3 * 128, w10, c->err) || !hg->getw("conv1.0", 64 * 27, w1, c->err)) return false;
            std::vector<float> w10a(3 * 64);
            for (int o = 0; o < 3; ++o)
                for (int k = 0; k < 64; ++k) w10a[o * 64 + k] = w10[o * 128 + k];
            c->hg_w10a = c->wts.put(w10a.data(), w10a.size() * 4);
            std::vector<f16> fr((size_t)10 * 64 * 8, (f16)0.f);      // 6 conv1 fragments (as pack_c3), 4 of conv10's second half
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int r = lane & 31, pslot = 8 * (lane >> 5) + j;
                    for (int i = 0; i < 2; ++i)
                        for (int ky = 0; ky < 3; ++ky)
                            fr[(((size_t)i * 3 + ky) * 64 + lane) * 8 + j] = (f16)c3_welem(w1, i * 32 + r, ky, lane >> 5, j);
                    for (int sidx = 0; sidx < 4; ++sidx)
                        if (r < 3) fr[((size_t)(6 + sidx) * 64 + lane) * 8 + j] = (f16)w10[r * 128 + 64 + 16 * sidx + acc_kperm16(pslot)];
                }
            c->hgf_wfrag = c->wts.put(fr.data(), fr.size() * sizeof(f16));
        }
    }
    return true;
}

// ------------------------------------------------------------------------------- workspace
inline int half_up(int n) { return (n - 1) / 2 + 1; }

Tensor &ws_add(hdrtv_ctx *c, const std::string &name, int C, int H, int W, int layout)
{
    Tensor t;
    t.C = C; t.H = H; t.W = W; t.layout = layout;
    t.off = c->ws.reserve(t.bytes() + 256);
    c->t[name] = t;
    return c->t[name];
}

template <typename T>
T *wsp(hdrtv_ctx *c, const std::string &name)
{
    auto it = c->t.find(name);
    if (it == c->t.end()) { fprintf(stderr, "hdrtv: internal error, no workspace tensor %s\n", name.c_str()); abort(); }
    return reinterpret_cast<T *>(c->ws.dev + it->second.off);
}
template <typename T>
const T *wtp(hdrtv_ctx *c, size_t off) { return reinterpret_cast<const T *>(c->wts.dev + off); }

// ATen _upsample_bicubic2d_aa tap table for scale 4 (see oracle/hdrtv_oracle.c aa_weights)
float cubic_aa(float x)
{
    const float a = -0.5f;
    x = std::fabs(x);
    if (x < 1.0f) return ((a + 2.0f) * x - (a + 3.0f)) * x * x + 1.0f;
    if (x < 2.0f) return (((x - 5.0f) * x + 8.0f) * x - 4.0f) * a;
    return 0.0f;
}
void aa_table(int in, int out, std::vector<float> &w, std::vector<int> &mn, std::vector<int> &ns)
{
    const float scale = 4.0f, support = 8.0f;
    w.assign((size_t)out * 17, 0.f);
    mn.resize(out);
    ns.resize(out);
    for (int i = 0; i < out; ++i) {
        const float center = scale * ((float)i + 0.5f);
        int xmin = (int)(center - support + 0.5f);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5f);
        if (xmax > in) xmax = in;
        const int xs = xmax - xmin;
        float total = 0.f;
        for (int j = 0; j < xs; ++j) {
            w[(size_t)i * 17 + j] = cubic_aa(((float)(j + xmin) - center + 0.5f) / scale);
            total += w[(size_t)i * 17 + j];
        }
        for (int j = 0; j < xs; ++j) w[(size_t)i * 17 + j] /= total;
        mn[i] = xmin;
        ns[i] = xs;
    }
}

struct Shapes {
    int H, W, h4, w4;
    int ch[6], cw[6];        // classifier spatial sizes: [0]=cond, [i]=after block i
    int H1, W1, H2, W2, H3, W3;
    int Hp, Wp;
};
Shapes shapes_for(int H, int W)
{
    Shapes s;
    s.H = H; s.W = W;
    s.h4 = H / 4 > 0 ? H / 4 : 1; s.w4 = W / 4 > 0 ? W / 4 : 1;
    s.ch[0] = s.h4; s.cw[0] = s.w4;
    for (int i = 1; i <= 5; ++i) { s.ch[i] = half_up(s.ch[i - 1]); s.cw[i] = half_up(s.cw[i - 1]); }
    s.H1 = half_up(H); s.W1 = half_up(W);
    s.H2 = half_up(s.H1); s.W2 = half_up(s.W1);
    s.H3 = half_up(s.H2); s.W3 = half_up(s.W2);
    s.Hp = (H + 31) / 32 * 32; s.Wp = (W + 31) / 32 * 32;
    return s;
}

int f32_plan(hdrtv_ctx *c, int H, int W);   // fp32_graph.inc: registers the fp32 graph's tensors

int do_reserve(hdrtv_ctx *c, int H, int W)
{
    if (c->H == H && c->W == W && c->ws.dev) return HDRTV_OK;
    if (H < 8 || W < 8 || H > 16384 || W > 16384) return fail(c, HDRTV_EINVAL, "unsupported frame size %dx%d", W, H);
    const Shapes s = shapes_for(H, W);
    // InstanceNorm2d needs more than one spatial element at the 4th classifier block (the reference raises
    // ValueError there too: torch/nn/functional.py _verify_spatial_size)
    // LDS-DMA (buffer loads) addresses a tensor with 32-bit byte offsets below 2 GiB: the widest tensors are 128 bytes per
    // (padded) pixel -> 16.7 Mpixel (5120 x 2880 fits; 7680 x 4320 does not)
    if ((size_t)((H + 31) / 32 * 32) * (size_t)((W + 31) / 32 * 32) * 128 >= ((size_t)1 << 31))
        return fail(c, HDRTV_EINVAL, "unsupported frame size %dx%d: more than 16.7 Mpixel (32-bit LDS-DMA offsets)", W, H);
    if (s.ch[4] * s.cw[4] < 2) return fail(c, HDRTV_EINVAL, "frame %dx%d too small for the AGCM classifier", W, H);
    // F.pad(mode="reflect") (HG_Composite_arch.py:97-103) needs the padding to be smaller than the dimension; torch raises
    if (c->has_hg && (s.Hp - H >= H || s.Wp - W >= W))
        return fail(c, HDRTV_EINVAL, "frame %dx%d too small for the HG head's reflect padding to a multiple of 32", W, H);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipDeviceSynchronize());
    c->H = c->W = 0;                       // no valid workspace until every step below has succeeded
    if (c->ws.dev) { (void)hipFree(c->ws.dev); c->ws.dev = nullptr; }
    c->ws = Arena();
    c->t.clear();
    // resize tables
    std::vector<float> wx, wy;
    std::vector<int> xmn, xns, ymn, yns;
    aa_table(W, s.w4, wx, xmn, xns);
    aa_table(H, s.h4, wy, ymn, yns);
    ws_add(c, "aa.wx", (int)wx.size(), 1, 1, 3); ws_add(c, "aa.wy", (int)wy.size(), 1, 1, 3);
    ws_add(c, "aa.xmn", s.w4, 1, 1, 3); ws_add(c, "aa.xns", s.w4, 1, 1, 3);
    ws_add(c, "aa.ymn", s.h4, 1, 1, 3); ws_add(c, "aa.yns", s.h4, 1, 1, 3);
    if (c->fp32) {
        f32_plan(c, H, W);
    } else {
    // AGCM
    const int cls_co[5] = {16, 32, 64, 128, 128};
    char nm[64];
    for (int i = 0; i < 5; ++i) {
        snprintf(nm, sizeof nm, "agcm.u%d", i + 1);
        ws_add(c, nm, cls_co[i], s.ch[i + 1], s.cw[i + 1], 2);
        snprintf(nm, sizeof nm, "agcm.mean%d", i + 1);
        ws_add(c, nm, cls_co[i], 1, 1, 3);
        snprintf(nm, sizeof nm, "agcm.rstd%d", i + 1);
        ws_add(c, nm, cls_co[i], 1, 1, 3);
    }
    ws_add(c, "agcm.part", 2 * 128 * ((s.ch[1] * s.cw[1] + 15) / 16) + 2 * 128 * 1024, 1, 1, 3);   // per-workgroup (sum, sumsq) partials
    ws_add(c, "agcm.frags", 14 * 64 * 8 / 2, 1, 1, 3);   // f16 elements stored in an f32-sized slot
    ws_add(c, "agcm.bias", 168, 1, 1, 3);
    ws_add(c, "agcm.out", 3, H, W, 1);
    // LE
    ws_add(c, "dbg.stamps", 8 * 8 * 512 * 2, 1, 1, 3);     // diagnostic builds only: [workgroup*NW + wave][8] u64 cycle sums
    ws_add(c, "le.cond", 64, H, W, 0);
    ws_add(c, "le.cond1", 16, H, W, 0);
    ws_add(c, "le.x192", 192, s.H1, s.W1, 0);
    ws_add(c, "le.h2a", 64, s.H2, s.W2, 0); ws_add(c, "le.h2b", 64, s.H2, s.W2, 0);
    if (c->hr_i8) {           // W8A8 condition nets: un-merged first layers, int8 codes between W8A8 layers
        ws_add(c, "le.c2a", 64, s.H1, s.W1, 0); ws_add(c, "le.c3a", 64, s.H1, s.W1, 0); ws_add(c, "le.c4a", 64, s.H1, s.W1, 0);
        ws_add(c, "le8.c3a", 64, s.H1, s.W1, 5); ws_add(c, "le8.c4a", 64, s.H1, s.W1, 5);
        ws_add(c, "le8.h2a", 64, s.H2, s.W2, 5); ws_add(c, "le8.h2b", 64, s.H2, s.W2, 5);
        ws_add(c, "le8.c2a", 64, s.H1, s.W1, 5);
        ws_add(c, "le8.img32", 32, H, W, 5);            // conv_first's input as NHWC codes (3 real channels)
        ws_add(c, "agcm.qconst", 320, 1, 1, 3);
    }
    ws_add(c, "le.cond2", 16, s.H1, s.W1, 0); ws_add(c, "le.cond3", 16, s.H2, s.W2, 0); ws_add(c, "le.cond4", 16, s.H3, s.W3, 0);
    ws_add(c, "le.f0a", 32, H, W, 0); ws_add(c, "le.f0b", 32, H, W, 0); ws_add(c, "le.fea0", 32, H, W, 0);
    ws_add(c, "le.up3", 32, H, W, 0);
    ws_add(c, "le.fea1a", 32, s.H1, s.W1, 0); ws_add(c, "le.fea1", 32, s.H1, s.W1, 0); ws_add(c, "le.l1b", 32, s.H1, s.W1, 0);
    ws_add(c, "le.up2", 32, s.H1, s.W1, 0); ws_add(c, "le.t5", 32, s.H1, s.W1, 0);
    ws_add(c, "le.fea2a", 32, s.H2, s.W2, 0); ws_add(c, "le.fea2", 32, s.H2, s.W2, 0); ws_add(c, "le.l2b", 32, s.H2, s.W2, 0);
    ws_add(c, "le.up1", 32, s.H2, s.W2, 0); ws_add(c, "le.t4", 32, s.H2, s.W2, 0);
    ws_add(c, "le.fea3", 32, s.H3, s.W3, 0); ws_add(c, "le.l3b", 32, s.H3, s.W3, 0);
    ws_add(c, "le.t3x", 32, s.H3, s.W3, 0); ws_add(c, "le.t3y", 32, s.H3, s.W3, 0);
    ws_add(c, "le.out", 3, H, W, 1);
    if (c->has_hg) {
        const int Hp = s.Hp, Wp = s.Wp;
        ws_add(c, "hg.img", 3, Hp, Wp, 1); ws_add(c, "hg.mask", 1, Hp, Wp, 4);
        ws_add(c, "hg.part", 4, Hp, Wp, 3);
        ws_add(c, "hg.part2", 4, Hp, Wp, 3);        // conv10's second half (over conv1), left by conv1's kernel
        if (!c->hg_i8) {
            ws_add(c, "hg.p1", 64, Hp / 2, Wp / 2, 0);
            ws_add(c, "hg.conv2", 128, Hp / 2, Wp / 2, 0); ws_add(c, "hg.up4", 128, Hp / 2, Wp / 2, 0);
            ws_add(c, "hg.p3", 256, Hp / 4, Wp / 4, 0); ws_add(c, "hg.conv3_2", 256, Hp / 4, Wp / 4, 0);
            ws_add(c, "hg.p4", 512, Hp / 8, Wp / 8, 0); ws_add(c, "hg.conv4_2", 512, Hp / 8, Wp / 8, 0);
            ws_add(c, "hg.p5", 512, Hp / 16, Wp / 16, 0); ws_add(c, "hg.conv5_2", 512, Hp / 16, Wp / 16, 0);
            ws_add(c, "hg.pc", 512, Hp / 32, Wp / 32, 0); ws_add(c, "hg.conv_code2", 512, Hp / 32, Wp / 32, 0);
            ws_add(c, "hg.up1", 512, Hp / 16, Wp / 16, 0); ws_add(c, "hg.conv6", 512, Hp / 16, Wp / 16, 0);
            ws_add(c, "hg.up2", 512, Hp / 8, Wp / 8, 0); ws_add(c, "hg.conv7", 256, Hp / 8, Wp / 8, 0);
            ws_add(c, "hg.up3", 256, Hp / 4, Wp / 4, 0); ws_add(c, "hg.conv8", 128, Hp / 4, Wp / 4, 0);
        } else {            // W8A8: the same tensors as int8 codes (q - 128), each once
            ws_add(c, "hg8.p1", 64, Hp / 2, Wp / 2, 5);
            ws_add(c, "hg8.conv2", 128, Hp / 2, Wp / 2, 5); ws_add(c, "hg8.up4", 128, Hp / 2, Wp / 2, 5);
            ws_add(c, "hg8.conv9", 64, Hp / 2, Wp / 2, 5);
            ws_add(c, "hg8.p3", 256, Hp / 4, Wp / 4, 5); ws_add(c, "hg8.conv3_2", 256, Hp / 4, Wp / 4, 5);
            ws_add(c, "hg8.p4", 512, Hp / 8, Wp / 8, 5); ws_add(c, "hg8.conv4_2", 512, Hp / 8, Wp / 8, 5);
            ws_add(c, "hg8.p5", 512, Hp / 16, Wp / 16, 5); ws_add(c, "hg8.conv5_2", 512, Hp / 16, Wp / 16, 5);
            ws_add(c, "hg8.pc", 512, Hp / 32, Wp / 32, 5); ws_add(c, "hg8.conv_code2", 512, Hp / 32, Wp / 32, 5);
            ws_add(c, "hg8.up1", 512, Hp / 16, Wp / 16, 5); ws_add(c, "hg8.conv6", 512, Hp / 16, Wp / 16, 5);
            ws_add(c, "hg8.up2", 512, Hp / 8, Wp / 8, 5); ws_add(c, "hg8.conv7", 256, Hp / 8, Wp / 8, 5);
            ws_add(c, "hg8.up3", 256, Hp / 4, Wp / 4, 5); ws_add(c, "hg8.conv8", 128, Hp / 4, Wp / 4, 5);
        }
        if (!c->hg_i8) ws_add(c, "hg.conv9", 64, Hp / 2, Wp / 2, 0);
    }
    }
    if (hipMalloc((void **)&c->ws.dev, c->ws.size + 4096) != hipSuccess) {
        c->ws.dev = nullptr;
        c->H = c->W = 0;
        return fail(c, HDRTV_ENOMEM, "workspace allocation of %zu bytes failed", c->ws.size);
    }
    hipError_t e = hipMemset(c->ws.dev, 0, c->ws.size + 4096);
    auto up = [&](const char *name, const void *src, size_t bytes) {
        if (e == hipSuccess) e = hipMemcpy(wsp<char>(c, name), src, bytes, hipMemcpyHostToDevice);
    };
    up("aa.wx", wx.data(), wx.size() * 4); up("aa.wy", wy.data(), wy.size() * 4);
    up("aa.xmn", xmn.data(), xmn.size() * 4); up("aa.xns", xns.data(), xns.size() * 4);
    up("aa.ymn", ymn.data(), ymn.size() * 4); up("aa.yns", yns.data(), yns.size() * 4);
    if (e != hipSuccess) {                 // leave no half-initialised workspace behind a size that looks reserved
        (void)hipFree(c->ws.dev);
        c->ws.dev = nullptr;
        return fail(c, HDRTV_EHIP, "workspace initialisation failed: %s", hipGetErrorString(e));
    }
    c->H = H; c->W = W;
    return HDRTV_OK;
}

// ----------------------------------------------------------------------- launch sequencing
struct Seq {
    hdrtv_ctx *c;
    hipStream_t s;
    int rc = HDRTV_OK;
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
    void conv(const std::string &key, const f16 *src0, int c0, const f16 *src1, int c1, int Hi, int Wi, int act, int mode,
              f16 *dst, int dstC, int Hd, int Wd, const f16 *res1 = nullptr, const f16 *res2 = nullptr, f16 *dst_full = nullptr,
              f16 *dst_planar = nullptr, const f16 *res_planar = nullptr, const float *dotw = nullptr, float *dst_dot = nullptr,
              int s0_stride = 0)
    {
        if (!ok()) return;
        auto it = c->conv.find(key);
        if (it == c->conv.end()) { rc = fail(c, HDRTV_ESTATE, "no packed conv %s", key.c_str()); return; }
        const ConvLayer &L = it->second;
        ConvParams p;
        memset(&p, 0, sizeof p);
        p.src0 = src0; p.src1 = src1; p.c0 = c0; p.c1 = c1;
        p.s0_stride = s0_stride ? s0_stride : c0; p.s1_stride = c1;
        p.Hi = Hi; p.Wi = Wi;
        const int pad = L.ks / 2;
        p.Ho = (Hi + 2 * pad - L.ks) / L.stride + 1;
        p.Wo = (Wi + 2 * pad - L.ks) / L.stride + 1;
        p.wpk = wtp<f16>(c, L.wpk); p.scale = wtp<float>(c, L.scale); p.shift = wtp<float>(c, L.shift);
        p.CoutPad = L.coutPad; p.Cout = L.cout; p.act = act; p.mode = mode;
        p.dst = dst; p.dst_full = dst_full; p.dstC = dstC; p.Hd = Hd; p.Wd = Wd;
        p.res1 = res1; p.res2 = res2; p.dst_planar = dst_planar; p.res_planar = res_planar;
        p.dotw = dotw; p.dst_dot = dst_dot;
        if (c0 + c1 != L.cin) { rc = fail(c, HDRTV_ESTATE, "conv %s: channel mismatch", key.c_str()); return; }
        p.zeros = wtp<f16>(c, c->zeros_off);
        const bool g64 = L.stride == 1 && L.cin_t == 64 && L.bn == 128 && !res1 && !res2 && !dst_full && mode != ST_PLANAR3;
        const bool pglds = g64 && L.ks == 3 && L.cout == L.coutPad;      // HG 3x3 convs: persistent LDS-DMA kernel
        const bool glds1 = g64 && L.ks == 1 && mode == ST_NHWC && (c0 + c1) >= 128 && L.coutPad <= 512 && (act == ACT_RELU || act == ACT_NONE);   // HG 1x1 fuse convs
        p.trash = const_cast<char *>(wtp<char>(c, c->dump_off));
        const int nt_slow = c->var.at("pglds_nt_slow");
        // default: the Up convs (Cout = 4 Cin: 4 .. 16 Cout-tiles per pixel tile) walk Cout-tile slowest -- an XCD then shares one
        // weight slab instead of re-fetching up to 16 (-17 % L2 misses, profiles/r02_pmc_traffic_tile_order.json); the other
        // layers walk it fastest so that the blocks of an XCD share halo tiles (+45 .. +75 % misses the other way round)
        p.nt_slow = nt_slow == 3 ? (mode == ST_PS) : (nt_slow == 2 ? (L.coutPad >= 512) : nt_slow);
        const bool s2g = L.ks == 3 && L.stride == 2 && L.cin_t == 64 && L.bn == L.coutPad && (L.coutPad == 64 || L.coutPad == 192);
        const bool no_t16 = c->var.at("no_t16") != 0;                         // developer A/B: the generic implicit-GEMM kernel
        const bool t16 = !no_t16 && L.ks == 3 && L.stride == 2 && L.cin == 32 && L.coutPad == 32 && !src1 && mode == ST_NHWC && !res1 && !res2;
        // HG 3x3 convs with Cout a multiple of 256: the private-weight schedule (conv3x3_prw.hip); variant prw = 0: conv_pglds
        // variant "prw": 0 = never, 1 (default) = the cheapest shape per layer, 2 / 3 = 16-row / 8-row tiles wherever it applies
        const int use_prw_mode = c->var.at("prw");
        const bool use_prw = use_prw_mode != 0;
        const bool prw_dot3 = mode == ST_PS_DOT3 && L.coutPad == 256;                // Up_conv5: always the 16-row shape
        bool prw = pglds && use_prw && (L.coutPad % 256) == 0 && (mode != ST_PS_DOT3 || prw_dot3);
        int prw_th = 16;
        if (prw && prw_dot3) {
        } else if (prw && use_prw_mode == 1) {
            // Its tiles cover 256 output channels (conv_pglds: 128).  Pick the shape whose tile count wastes least of the last
            // round on n_cu workgroups: relative cost per unit of work 1.0 (16-row tiles), 1.09 (8-row tiles: twice the weight
            // bytes per MAC, 1.11x the halo), 1.15 - 1.22 (conv_pglds) -- measured on full rounds, profiles/r03_prw_ab.txt
            const long tx = (p.Wo + 15) / 16, n = c->n_cu;
            auto cost = [&](long tiles, double rel) { return (double)(((tiles + n - 1) / n) * n) / (double)tiles * rel; };
            const double c16 = cost(tx * ((p.Ho + 15) / 16) * (L.coutPad / 256), 1.0);
            const double c8 = cost(tx * ((p.Ho + 7) / 8) * (L.coutPad / 256), 1.09);
            const double c0 = cost(tx * ((p.Ho + 15) / 16) * (L.coutPad / 128), 1.22);
            if (c0 <= c16 && c0 <= c8) prw = false;
            else prw_th = c8 < c16 ? 8 : 16;
        } else if (prw && use_prw_mode == 3) {
            prw_th = 8;
        }
        char tag[64];
        if (t16) snprintf(tag, sizeof tag, "conv_t16<32,3,2>");
        else if (s2g) snprintf(tag, sizeof tag, "conv3x3s2_preg<%d>", L.coutPad);
        else if (pglds) snprintf(tag, sizeof tag, "%s<%s>", prw ? (prw_th == 8 ? "conv_prw8" : "conv_prw") : "conv_pglds", mode == ST_POOL ? "pool" : (mode == ST_PS ? "ps" : (mode == ST_PS_DOT3 ? "ps_dot3" : "nhwc")));
        else if (glds1) snprintf(tag, sizeof tag, "conv_glds1");
        else snprintf(tag, sizeof tag, "conv_igemm<%d,%d,%d,%d>", L.cin_t, L.bn, L.ks, L.stride);
        const double macs = (double)p.Ho * p.Wo * L.cin * L.ks * L.ks * L.cout;
        double bytes = 2.0 * Hi * Wi * L.cin + 2.0 * L.ks * L.ks * L.cin * L.coutPad;
        const double outel = mode == ST_POOL ? (double)Hd * Wd * L.cout + (dst_full ? (double)p.Ho * p.Wo * L.cout : 0.0)
                                              : (mode == ST_PLANAR3 ? 3.0 * Hd * Wd
                                                                    : (mode == ST_PS_DOT3 ? 8.0 * Hd * Wd : (double)p.Ho * p.Wo * L.cout));
        bytes += 2.0 * outel * (1 + (res1 ? 1 : 0) + (res2 ? 1 : 0) + (res_planar ? 1 : 0));
        chk(t16 ? conv_t16_launch(p, s, c->n_cu) : s2g ? conv3x3s2_preg_launch(p, c->n_cu, s)
                : (pglds ? (prw ? conv_prw_launch(p, prw_th, c->n_cu, s) : conv_pglds_launch(p, c->n_cu, s))
                         : (glds1 ? conv_glds1_launch(p, s, c->n_cu, c->var.at("glds1_old") != 0) : conv_igemm_launch(p, L.cin_t, L.bn, L.ks, L.stride, s))),
            key.c_str(), tag, macs, bytes);
    }
    // W8A8 HG layer on int8 MFMA: 3x3 (conv3x3_pglds_i8.hip) or 1x1 (conv_i8_misc.hip)
    void conv8(const std::string &key, const int8_t *src0, int c0, const int8_t *src1, int c1, int Hi, int Wi, int mode, void *dst,
               int dstC, int Hd, int Wd, const float *dotw = nullptr, float *dst_dot = nullptr)
    {
        if (!ok()) return;
        auto it = c->conv8.find(key);
        if (it == c->conv8.end()) { rc = fail(c, HDRTV_ESTATE, "no packed int8 conv %s", key.c_str()); return; }
        const ConvI8Layer &L = it->second;
        if (c0 + c1 != L.cin) { rc = fail(c, HDRTV_ESTATE, "conv %s: channel mismatch", key.c_str()); return; }
        ConvI8Params p;
        memset(&p, 0, sizeof p);
        p.src0 = src0; p.src1 = src1; p.c0 = c0; p.c1 = c1; p.Hi = Hi; p.Wi = Wi; p.Ho = Hi; p.Wo = Wi;
        p.wpk = wtp<int8_t>(c, L.wpk); p.scale = wtp<float>(c, L.scale); p.shift = wtp<float>(c, L.shift);
        p.Cout = L.cout; p.mode = mode; p.out_f16 = L.out_f16; p.dst = dst; p.dstC = dstC; p.Hd = Hd; p.Wd = Wd;
        p.padline = wtp<int8_t>(c, L.padline);
        p.delta = L.has_delta ? wtp<float>(c, L.delta) : nullptr;
        p.delta_acc = L.has_delta ? wtp<int>(c, L.delta_acc) : nullptr;
        p.lo_clamp = L.lo_clamp;
        p.trash = const_cast<char *>(wtp<char>(c, c->dump_off));
        p.dotw = dotw; p.dst_dot = dst_dot;
        char tag[64];
        if (L.ks == 3) snprintf(tag, sizeof tag, "conv_pglds_i8<%s%s>", mode == ST_POOL ? "pool" : (mode == ST_PS ? "ps" : (mode == ST_PS_DOT3 ? "ps_dot3" : "nhwc")), c0 == 64 ? ",c64" : "");
        else snprintf(tag, sizeof tag, "conv1x1_i8%s", L.out_f16 ? "<f16>" : "");
        const double macs = (double)Hi * Wi * L.cin * L.ks * L.ks * L.cout_real;
        const double outel = mode == ST_POOL ? (double)Hd * Wd * L.cout_real : (double)Hi * Wi * L.cout_real;
        const double bytes = (double)Hi * Wi * L.cin + (double)L.ks * L.ks * L.cin * L.cout +
                             (mode == ST_PS_DOT3 ? 16.0 * Hd * Wd : outel * (L.out_f16 ? 2.0 : 1.0));
        // the private-weight schedule (conv3x3_prw_i8.hip) and its tile shape, picked as for the fp16 layers (Seq::conv)
        const int prw_mode = c->var.at("prw");
        bool prw = prw_mode != 0 && L.ks == 3 && c0 != 64 && (L.cout % 256) == 0 && mode != ST_PS_DOT3 && !L.out_f16;
        int prw_th = 16;
        if (prw && prw_mode == 1) {
            const long tx = (Wi + 15) / 16, n = c->n_cu;
            auto cost = [&](long tiles, double rel) { return (double)(((tiles + n - 1) / n) * n) / (double)tiles * rel; };
            const double c16 = cost(tx * ((Hi + 15) / 16) * (L.cout / 256), 1.0), c8 = cost(tx * ((Hi + 7) / 8) * (L.cout / 256), 1.09);
            const double c0c = cost(tx * ((Hi + 15) / 16) * (L.cout / 128), 1.22);
            if (c0c <= c16 && c0c <= c8) prw = false;
            else prw_th = c8 < c16 ? 8 : 16;
        } else if (prw && prw_mode == 3) {
            prw_th = 8;
        }
        // variant "prw_i8": 0 = never, 1 = only where the 8-row tiles win (the low-resolution layers), 2 = wherever "prw" selects it
        const int i8_mode = c->var.at("prw_i8");
        if (i8_mode == 0 || (i8_mode == 1 && prw_th != 8)) prw = false;
        if (prw) snprintf(tag, sizeof tag, "conv_prw%s_i8<%s>", prw_th == 8 ? "8" : "", mode == ST_POOL ? "pool" : (mode == ST_PS ? "ps" : "nhwc"));
        chk(L.ks == 3 ? (prw ? conv_prw_i8_launch(p, prw_th, c->n_cu, s) : conv_pglds_i8_launch(p, c->n_cu, s)) : conv1x1_i8_launch(p, s),
            key.c_str(), tag, macs, bytes);
    }
    // W8A8 LE layer on int8 MFMA (conv_q8.hip).  src: f16 NHWC (quantised on load) or this layer's int8 codes; dst: f16, or
    // (oq != nullptr) the int8 codes of the reading layer's quantiser *oq
    void convq8(const std::string &key, const void *src, bool src_i8, int src_stride, int Hi, int Wi, int act, void *dst, int dstC,
                const ActQf *oq)
    {
        if (!ok()) return;
        auto it = c->q8.find(key);
        if (it == c->q8.end()) { rc = fail(c, HDRTV_ESTATE, "no packed W8A8 conv %s", key.c_str()); return; }
        const QLayer &L = it->second;
        ConvQ8Params p;
        memset(&p, 0, sizeof p);
        p.src = src; p.src_i8 = src_i8 ? 1 : 0; p.Cin = L.cin; p.src_stride = src_stride; p.Hi = Hi; p.Wi = Wi;
        p.ks = L.ks; p.stride = L.stride;
        const int pad = L.ks / 2;
        p.Ho = (Hi + 2 * pad - L.ks) / L.stride + 1; p.Wo = (Wi + 2 * pad - L.ks) / L.stride + 1;
        p.wpk8 = wtp<int8_t>(c, L.wpk8); p.scale = wtp<float>(c, L.scale); p.shift = wtp<float>(c, L.shift);
        p.CoutPad = L.coutPad; p.Cout = L.cout; p.act = act; p.q_inv = L.q.inv(); p.q_zoff = L.q.zoff();
        p.dst = dst; p.dst_i8 = oq ? 1 : 0; p.dstC = dstC;
        if (oq) { p.oq_inv = oq->inv(); p.oq_zoff = oq->zoff(); }
        char tag[64];
        snprintf(tag, sizeof tag, "conv_q8<%d,%d,%d>", L.cin, L.ks, L.stride);
        const double macs = (double)p.Ho * p.Wo * L.cin * L.ks * L.ks * L.cout;
        const double bytes = (double)Hi * Wi * L.cin * (src_i8 ? 1.0 : 2.0) + (double)L.ks * L.ks * L.cin * L.coutPad +
                             (double)p.Ho * p.Wo * L.cout * (oq ? 1.0 : 2.0);
        chk(conv_q8_launch(p, s, c->n_cu), key.c_str(), tag, macs, bytes);
    }
    void c3(const std::string &key, const f16 *in, int H, int W, int act, f16 *out, f16 *out_pool, float pool_q_inv = 0.f,
            float pool_q_zero = 0.f, const f16 *w2frag = nullptr, float *part2 = nullptr)
    {
        if (!ok()) return;
        const C3Layer &L = c->c3.at(key);
        chk(conv_c3_launch(in, H, W, wtp<f16>(c, L.wfrag), wtp<float>(c, L.scale), wtp<float>(c, L.shift), L.cout, act, out,
                           out_pool, c->n_cu, s, pool_q_inv, pool_q_zero, w2frag, part2), key.c_str(),
            part2 ? "conv_c3<64,dot3>" : (L.cout == 64 ? "conv_c3<64>" : "conv_c3<32>"), (double)H * W * (27 * L.cout + (part2 ? 192 : 0)),
            (double)H * W * (6.0 + (out ? 2.0 * L.cout : 0.0) + (out_pool ? (pool_q_inv > 0.f ? 0.25 : 0.5) * L.cout : 0.0) + (part2 ? 16.0 : 0.0)));
    }
    // persistent 32-channel 3x3 conv, optionally with the SFT layer `sft_key` fused in front (conv32p.hip)
    void conv32(const std::string &key, const f16 *src, const f16 *cond, const std::string &sft_key, int H, int W, int act,
                int mode, f16 *dst, int dstC, int Hd, int Wd, const f16 *res1 = nullptr, const f16 *res2 = nullptr,
                f16 *dst_planar = nullptr, const f16 *res_planar = nullptr, const f16 *c3_img = nullptr, const std::string &c3_key = "")
    {
        if (!ok()) return;
        auto it = c->conv.find(key);
        auto iq = c->q32.find(key);
        if (it == c->conv.end() && iq == c->q32.end()) { rc = fail(c, HDRTV_ESTATE, "no packed conv %s", key.c_str()); return; }
        const bool i8 = iq != c->q32.end();
        ConvLayer L;
        Conv32Params p;
        memset(&p, 0, sizeof p);
        if (i8) {             // W8A8 layer: int8 MFMA on the quantised tile
            const QLayer &Q = iq->second;
            L.cout = Q.cout; L.coutPad = Q.coutPad;
            p.wpk8 = wtp<int8_t>(c, Q.wpk8); p.scale = wtp<float>(c, Q.scale); p.shift = wtp<float>(c, Q.shift);
            p.q_inv = Q.q.inv(); p.q_zoff = Q.q.zoff();
        } else {
            L = it->second;
            p.wpk = wtp<f16>(c, L.wpk); p.scale = wtp<float>(c, L.scale); p.shift = wtp<float>(c, L.shift);
        }
        p.src = src; p.cond = cond; p.H = H; p.W = W;
        bool sq = false;
        if (cond) {
            const SftLayer &S = c->sft.at(sft_key);
            p.sft_wfrag = wtp<f16>(c, S.wfrag); p.sft_bias = wtp<float>(c, S.bias);
            if (S.q) {        // W8A8 SFT convs: int8 MFMA on the quantised condition pixel
                sq = true;
                p.sq_wfrag = wtp<int8_t>(c, S.qfrag); p.sq_const = wtp<float>(c, S.qconst);
                for (int b = 0; b < 2; ++b) { p.sq_inv[b] = S.inv[b]; p.sq_zoff[b] = S.zoff[b]; p.sq_hzoff[b] = S.hzoff[b]; }
            }
        }
        p.CoutPad = L.coutPad; p.Cout = L.cout; p.act = act; p.mode = mode;
        p.dst = dst; p.dstC = dstC; p.Hd = Hd; p.Wd = Wd; p.res1 = res1; p.res2 = res2;
        p.dst_planar = dst_planar; p.res_planar = res_planar; p.zeros = wtp<f16>(c, c->zeros_off);
        p.dump = reinterpret_cast<f16 *>(stamp_buf());
        const double npx = (double)H * W;
        const double macs = npx * 32 * 9 * L.cout + (cond ? npx * 2 * (16 * 16 + 16 * 32) : 0.0) + (c3_img ? npx * 27 * 32 : 0.0);
        const double outb = mode == ST_PLANAR3 ? 6.0 * npx : 2.0 * npx * L.cout;
        const double bytes = npx * ((c3_img ? 6 : 64) + (cond ? 32 : 0)) + outb * (1 + (res1 ? 1 : 0) + (res2 ? 1 : 0) + (res_planar ? 1 : 0)) +
                             (i8 ? 1.0 : 2.0) * 9 * 32 * L.coutPad;
        p.trash = const_cast<char *>(wtp<char>(c, c->dump_off));
        if (c3_img) {         // conv_first fused in front of SFT_layer1 + HR_conv1: src is the planar image
            const C3Layer &L3 = c->c3.at(c3_key);
            p.c3_img = c3_img; p.c3_wfrag = wtp<f16>(c, L3.wfrag);      // bias inside the fragments (pack_c3), no BatchNorm
        }
        // single-pass layers run the one-barrier schedule (conv32s.hip); variant "conv32_old" is the developer A/B switch
        const bool old_sched = c->var.at("conv32_old") != 0;
        const bool one_barrier = L.coutPad == 32 && !old_sched;
        char tag[48];
        snprintf(tag, sizeof tag, "conv32%c<%d,%s%s%s>", one_barrier ? 's' : 'p', L.coutPad / 32, c3_img ? "c3+" : "", cond ? (sq ? "sft-i8" : "sft") : "plain", i8 ? ",i8" : "");
        if (c3_img && !one_barrier) { rc = fail(c, HDRTV_ESTATE, "conv_first fusion needs the one-barrier schedule"); return; }
        chk(one_barrier ? conv32s_launch(p, c->n_cu, s, c->var.at("conv32_nosplit") != 0) : conv32p_launch(p, c->n_cu, s, c->var.at("conv32_nw")), key.c_str(), tag, macs, bytes);
    }
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
    void resblock(const std::string &base, const f16 *x, const f16 *cond, int H, int W, f16 *tb, f16 *y,
                  const f16 *extra = nullptr)
    {
        // fp16 block without a second residual, enough rows per segment to amortise the 4-row warm-up: ONE row-streaming
        // launch (le_rows.hip), the intermediate never leaves LDS; bit-identical to the two launches below
        if (ok() && c->var.at("le_rows") && !extra && rows_fit(H, W)) {
            RowsRbParams p;
            memset(&p, 0, sizeof p);
            bool q1, q2, qs1, qs2;
            const ConvLayer *L1 = rows_conv(base + ".conv1", p.fq_c1, q1), *L2 = rows_conv(base + ".conv2", p.fq_c2, q2);
            const SftLayer &S1 = c->sft.at(base + ".sft1"), &S2 = c->sft.at(base + ".sft2");
            if (L1 && L2 && rows_sft(S1, p.fq_s1, qs1) && rows_sft(S2, p.fq_s2, qs2)) {
                p.fq = (q1 ? 1 : 0) | (q2 ? 2 : 0) | (qs1 ? 4 : 0) | (qs2 ? 8 : 0);
                p.x = x; p.cond = cond; p.H = H; p.W = W; p.dst = y;
                p.w1 = wtp<f16>(c, L1->wpk); p.w2 = wtp<f16>(c, L2->wpk); p.b1 = wtp<float>(c, L1->shift); p.b2 = wtp<float>(c, L2->shift);
                p.sft1_wfrag = wtp<f16>(c, S1.wfrag); p.sft1_bias = wtp<float>(c, S1.bias);
                p.sft2_wfrag = wtp<f16>(c, S2.wfrag); p.sft2_bias = wtp<float>(c, S2.bias);
                p.trash = const_cast<char *>(wtp<char>(c, c->dump_off));
                p.dump = stamp_buf();
                const double npx = (double)H * W;
                chk(le_rb_rows_launch(p, c->n_cu, s), base.c_str(), p.fq ? "le_rb_rows<fq>" : "le_rb_rows",
                    npx * (2.0 * 32 * 9 * 32 + 4.0 * (16 * 16 + 16 * 32)), npx * (64 + 32 + 64) + 2.0 * 2 * 9 * 32 * 32);
                return;
            }
        }
        conv32(base + ".conv1", x, cond, base + ".sft1", H, W, ACT_RELU, ST_NHWC, tb, 32, H, W);
        conv32(base + ".conv2", tb, cond, base + ".sft2", H, W, ACT_NONE, ST_NHWC, y, 32, H, W, x, extra);
    }
};

#include "fp32_graph.inc"

int f32_plan(hdrtv_ctx *c, int H, int W)
{
    Seq q{c, nullptr};
    return run_f32(c, q, true, H, W, nullptr, nullptr, nullptr, nullptr);
}

int run_agcm(hdrtv_ctx *c, Seq &q, const f16 *rgb, const f16 *cond, f16 *agcm_out)
{
    const Shapes s = shapes_for(c->H, c->W);
    const int cls_ci[5] = {3, 16, 32, 64, 128}, cls_co[5] = {16, 32, 64, 128, 128};
    char a[64], b[64];
    for (int i = 0; i < 5; ++i) {
        snprintf(a, sizeof a, "agcm.u%d", i + 1);
        float *out = wsp<float>(c, a);
        const void *in = cond;
        const float *nm = nullptr, *nr = nullptr, *ng = nullptr, *nb = nullptr;
        if (i > 0) {
            snprintf(b, sizeof b, "agcm.u%d", i);
            in = wsp<float>(c, b);
            snprintf(b, sizeof b, "agcm.mean%d", i); nm = wsp<float>(c, b);
            snprintf(b, sizeof b, "agcm.rstd%d", i); nr = wsp<float>(c, b);
            snprintf(b, sizeof b, "cls%d.g", i - 1); ng = wtp<float>(c, c->f32v.at(b));
            snprintf(b, sizeof b, "cls%d.be", i - 1); nb = wtp<float>(c, c->f32v.at(b));
        }
        snprintf(b, sizeof b, "cls%d.w", i);
        const float *w = wtp<float>(c, c->f32v.at(b));
        snprintf(b, sizeof b, "cls%d.b", i);
        const float *bias = wtp<float>(c, c->f32v.at(b));
        int nblk = 0;
        q.chk(cls_block_launch(in, i == 0, cls_ci[i], s.ch[i], s.cw[i], nm, nr, ng, nb, w, bias, cls_co[i], out, s.ch[i + 1],
                               s.cw[i + 1], wsp<float>(c, "agcm.part"), q.s, c->cls_q[i].on ? &c->cls_q[i] : nullptr,
                               (i == 4 && c->cls_q[5].on) ? &c->cls_q[5] : nullptr, &nblk),
              "cls_block", c->cls_q[i].on ? "cls_block<fq>" : "cls_block", (double)s.ch[i] * s.cw[i] * cls_ci[i] * cls_co[i]);
        snprintf(a, sizeof a, "agcm.mean%d", i + 1);
        snprintf(b, sizeof b, "agcm.rstd%d", i + 1);
        q.chk(cls_stats_launch(wsp<float>(c, "agcm.part"), cls_co[i], nblk, s.ch[i + 1] * s.cw[i + 1], 1e-5f, wsp<float>(c, a),
                               wsp<float>(c, b), q.s),
              "cls_stats", "cls_stats");
    }
    AgcmFoldArgs fa;
    fa.mean5 = wsp<float>(c, "agcm.mean5");
    fa.w20 = wtp<float>(c, c->f32v.at("cls20.w")); fa.b20 = wtp<float>(c, c->f32v.at("cls20.b"));
    for (int st = 0; st < 3; ++st) {
        snprintf(a, sizeof a, "gfm.s%d.w", st); fa.ws[st] = wtp<float>(c, c->f32v.at(a));
        snprintf(a, sizeof a, "gfm.s%d.b", st); fa.bs[st] = wtp<float>(c, c->f32v.at(a));
        snprintf(a, sizeof a, "gfm.t%d.w", st); fa.wt[st] = wtp<float>(c, c->f32v.at(a));
        snprintf(a, sizeof a, "gfm.t%d.b", st); fa.bt[st] = wtp<float>(c, c->f32v.at(a));
    }
    fa.w1 = wtp<float>(c, c->f32v.at("agcm.w1")); fa.b1 = wtp<float>(c, c->f32v.at("agcm.b1"));
    fa.w2 = wtp<float>(c, c->f32v.at("agcm.w2")); fa.b2 = wtp<float>(c, c->f32v.at("agcm.b2"));
    fa.w3 = wtp<float>(c, c->f32v.at("agcm.w3")); fa.b3 = wtp<float>(c, c->f32v.at("agcm.b3"));
    if (c->agcm_q8) {         // W8A8 GFM convs: per-frame dequantisation constants, then the int8 chain
        AgcmFoldQ8Args qa;
        qa.q20 = c->cls_q[5];
        for (int i = 0; i < 6; ++i) qa.qlin[i] = c->lin_q[i];
        qa.P = wtp<float>(c, c->ag_P); qa.Q = wtp<float>(c, c->ag_Q);
        qa.inv2 = c->ag_q[1].inv(); qa.inv3 = c->ag_q[2].inv();
        qa.consts = wsp<float>(c, "agcm.qconst");
        q.chk(agcm_fold_q8_launch(fa, qa, wsp<float>(c, "agcm.bias"), q.s), "agcm_fold", "agcm_fold<q8>", 128.0 * 6 + 6.0 * (64 + 64 + 3) * 2);
        q.chk(agcm_mlp_q8_launch(rgb, agcm_out, (size_t)c->H * c->W, wtp<int8_t>(c, c->ag_frag), qa.consts, c->ag_q[0].inv(), c->ag_q[0].zoff(),
                                 c->ag_q[1].zoff(), c->ag_q[2].zoff(), q.s),
              "agcm_mlp", "agcm_mlp<q8>", (double)c->H * c->W * (3 * 64 + 64 * 64 + 64 * 3), 12.0 * c->H * c->W);
        return q.rc;
    }
    q.chk(agcm_fold_launch(fa, wsp<f16>(c, "agcm.frags"), wsp<float>(c, "agcm.bias"), q.s), "agcm_fold", "agcm_fold",
          128.0 * 6 + 6.0 * (64 + 64 + 3) * 2);
    q.chk(agcm_mlp_launch(rgb, agcm_out, (size_t)c->H * c->W, wsp<f16>(c, "agcm.frags"), wsp<float>(c, "agcm.bias"), q.s),
          "agcm_mlp", "agcm_mlp", (double)c->H * c->W * (3 * 64 + 64 * 64 + 64 * 3), 12.0 * c->H * c->W);
    return q.rc;
}

// HDRUNet3T1._forward_safe_aligned (HDRUNet3T1_arch.py:152-206) with x = [agcm_out, agcm_out]
int run_le(hdrtv_ctx *c, Seq &q, const f16 *img, f16 *out_planar)
{
    const Shapes s = shapes_for(c->H, c->W);
    const int H = s.H, W = s.W;
    f16 *cond = wsp<f16>(c, "le.cond");
    f16 *cond1 = wsp<f16>(c, "le.cond1"), *cond2 = wsp<f16>(c, "le.cond2"), *cond3 = wsp<f16>(c, "le.cond3"),
        *cond4 = wsp<f16>(c, "le.cond4");
    f16 *h2a = wsp<f16>(c, "le.h2a");
    // condition trunk
    // cond_first (3 layers) + CondNet1 (3 layers) in one launch: img -> cond (64 ch) and cond1 (16 ch)
    auto qlast = [&](const QLastLayer &Q, QLastArgs &a) -> const QLastArgs * {
        if (!Q.on) return nullptr;
        a.wq = wtp<int8_t>(c, Q.wq); a.ss = wtp<float>(c, Q.ss); a.q_inv = Q.q.inv(); a.q_zoff = Q.q.zoff();
        return &a;
    };
    QLastArgs qa6, qa2;
    if (q.ok() && c->trunk_q8) {
        TrunkQ8Args ta;
        ta.wfrag = wtp<int8_t>(c, c->tq_frag); ta.consts = wtp<float>(c, c->tq_const);
        ta.q1_inv = c->tq_q[0].inv(); ta.q1_zoff = c->tq_q[0].zoff(); ta.q4_inv = c->tq_q[3].inv();
        for (int i = 0; i < 5; ++i) ta.zoff[i] = c->tq_q[i + 1].zoff();
        q.chk(le_cond_trunk_q8_launch(img, H, W, ta, cond, cond1, c->n_cu, q.s), "LE.cond_trunk", "le_cond_trunk_q8",
              (double)H * W * (27 * 64 + 4 * 64 * 64 + 64 * 16), (double)H * W * (6 + 128 + 32));
    } else if (q.ok())
        q.chk(le_cond_trunk_launch(img, H, W, wtp<f16>(c, c->trunk_wfrag), wtp<float>(c, c->trunk_bias), cond, cond1, c->n_cu, q.s,
                                   qlast(c->q_trunk6, qa6)),
              "LE.cond_trunk", c->q_trunk6.on ? "le_cond_trunk<q6>" : "le_cond_trunk", (double)H * W * (27 * 64 + 4 * 64 * 64 + 64 * 16), (double)H * W * (6 + 128 + 32));
    auto isq8 = [&](const char *L) { return c->q8.find(L) != c->q8.end(); };
    auto qof = [&](const char *L) -> const ActQf * { auto it = c->q8.find(L); return it == c->q8.end() ? nullptr : &it->second.q; };
    f16 *h2b = wsp<f16>(c, "le.h2b");
    // CondNet{2,3,4}.0 (3x3 / stride 2 from the 64-channel condition map).  All-fp16 recipes read `cond` once (one launch,
    // 192 channels); a W8A8 layer among them quantises `cond` with its own x_scale / x_zero and runs alone, writing the int8
    // codes of the layer that reads it when that one is W8A8 too.
    const f16 *a2 = nullptr, *a3 = nullptr, *a4 = nullptr;
    const int8_t *a2q = nullptr, *a3q = nullptr, *a4q = nullptr;
    int astride = 64;
    if (c->conv.find("LE.CondNet234.0") != c->conv.end()) {
        f16 *x192 = wsp<f16>(c, "le.x192");
        q.conv("LE.CondNet234.0", cond, 64, nullptr, 0, H, W, ACT_LRELU01, ST_NHWC, x192, 192, s.H1, s.W1);
        a2 = x192; a3 = x192 + 64; a4 = x192 + 128; astride = 192;
    } else {
        f16 *ca[3] = {wsp<f16>(c, "le.c2a"), wsp<f16>(c, "le.c3a"), wsp<f16>(c, "le.c4a")};
        int8_t *ca8[3] = {wsp<int8_t>(c, "le8.c2a"), wsp<int8_t>(c, "le8.c3a"), wsp<int8_t>(c, "le8.c4a")};
        const char *l0[3] = {"LE.CondNet2.0", "LE.CondNet3.0", "LE.CondNet4.0"}, *l2[3] = {nullptr, "LE.CondNet3.2", "LE.CondNet4.2"};
        const f16 **af[3] = {&a2, &a3, &a4};
        const int8_t **aq[3] = {&a2q, &a3q, &a4q};
        // the W8A8 ones together: `cond` is read once and quantised per layer in registers (conv_q8_multi)
        ConvQ8MultiParams mp;
        memset(&mp, 0, sizeof mp);
        mp.src = cond; mp.src_stride = 64; mp.Hi = H; mp.Wi = W; mp.Ho = s.H1; mp.Wo = s.W1;
        double m_macs = 0.0, m_bytes = 2.0 * 64 * H * W;
        for (int i = 0; i < 3; ++i) {
            if (!isq8(l0[i])) continue;
            const QLayer &L = c->q8.at(l0[i]);
            const ActQf *oq = l2[i] ? qof(l2[i]) : (c->tail_q8 ? &c->tl_q[0] : nullptr);    // CondNet2.0 feeds the fused tail
            ConvQ8Group &G = mp.g[mp.ngroups++];
            G.wpk8 = wtp<int8_t>(c, L.wpk8); G.scale = wtp<float>(c, L.scale); G.shift = wtp<float>(c, L.shift);
            G.q_inv = L.q.inv(); G.q_zoff = L.q.zoff(); G.act = ACT_LRELU01;
            G.dst = oq ? (void *)ca8[i] : (void *)ca[i]; G.dst_i8 = oq ? 1 : 0;
            if (oq) { G.oq_inv = oq->inv(); G.oq_zoff = oq->zoff(); *aq[i] = ca8[i]; } else { *af[i] = ca[i]; }
            m_macs += (double)s.H1 * s.W1 * 64 * 9 * 64;
            m_bytes += (double)s.H1 * s.W1 * 64 * (oq ? 1.0 : 2.0) + 9.0 * 64 * 64;
        }
        if (mp.ngroups && q.ok()) {
            char tag[48];
            snprintf(tag, sizeof tag, "conv_q8_multi<%d>", mp.ngroups);
            q.chk(conv_q8_multi_launch(mp, c->n_cu, q.s), "LE.CondNet234.0", tag, m_macs, m_bytes);
        }
        for (int i = 0; i < 3; ++i) {
            if (isq8(l0[i])) {
                continue;
            } else {
                q.conv(l0[i], cond, 64, nullptr, 0, H, W, ACT_LRELU01, ST_NHWC, ca[i], 64, s.H1, s.W1);
                *af[i] = ca[i];
            }
        }
    }
    // CondNet2.2 + .4 (1x1 64->64, LeakyReLU, 1x1 64->16) in one pass over CondNet2.0's 64 channels
    if (q.ok() && c->tail_q8) {
        if (!a2q) return q.rc = fail(c, HDRTV_ESTATE, "internal: W8A8 CondNet2 tail without int8 input");
        q.chk(cond_tail_q8_launch(a2q, (size_t)s.H1 * s.W1, wtp<int8_t>(c, c->tl_frag), wtp<float>(c, c->tl_const), c->tl_q[1].zoff(), cond2,
                                  c->n_cu, q.s),
              "LE.CondNet2.2+4", "cond_tail_q8", (double)s.H1 * s.W1 * (64 * 64 + 64 * 16), (double)s.H1 * s.W1 * (64 + 32));
    } else if (q.ok())
        q.chk(cond_tail_launch(a2, astride, (size_t)s.H1 * s.W1, wtp<f16>(c, c->tail_wfrag), wtp<float>(c, c->tail_bias), cond2, c->n_cu, q.s,
                               qlast(c->q_tail2, qa2)),
              "LE.CondNet2.2+4", c->q_tail2.on ? "cond_tail<q2>" : "cond_tail", (double)s.H1 * s.W1 * (64 * 64 + 64 * 16), (double)s.H1 * s.W1 * (128 + 32));
    // CondNet3 / CondNet4: .2 (3x3 / stride 2, 64 -> 64, LeakyReLU) then .4 (1x1 resp. 3x3 / stride 2, 64 -> 16)
    {
        const f16 *af[2] = {a3, a4};
        const int8_t *aq[2] = {a3q, a4q};
        f16 *h2[2] = {h2a, h2b}, *cout[2] = {cond3, cond4};
        const char *l2[2] = {"LE.CondNet3.2", "LE.CondNet4.2"}, *l4[2] = {"LE.CondNet3.4", "LE.CondNet4.4"};
        for (int i = 0; i < 2; ++i) {
            const void *h = h2[i];
            bool h_i8 = false;
            if (isq8(l2[i])) {
                const ActQf *oq = qof(l4[i]);
                int8_t *h8 = oq ? wsp<int8_t>(c, i ? "le8.h2b" : "le8.h2a") : nullptr;
                q.convq8(l2[i], aq[i] ? (const void *)aq[i] : (const void *)af[i], aq[i] != nullptr, aq[i] ? 64 : astride, s.H1, s.W1,
                         ACT_LRELU01, oq ? (void *)h8 : (void *)h2[i], 64, oq);
                if (oq) { h = h8; h_i8 = true; }
            } else {
                q.conv(l2[i], af[i], 64, nullptr, 0, s.H1, s.W1, ACT_LRELU01, ST_NHWC, h2[i], 64, s.H2, s.W2, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, astride);
            }
            if (isq8(l4[i])) q.convq8(l4[i], h, h_i8, 64, s.H2, s.W2, ACT_NONE, cout[i], 16, nullptr);
            else q.conv(l4[i], h2[i], 64, nullptr, 0, s.H2, s.W2, ACT_NONE, ST_NHWC, cout[i], 16, i ? s.H3 : s.H2, i ? s.W3 : s.W2);
        }
    }
    // main branch: every SFT is fused into the 3x3 conv that follows it
    f16 *f0a = wsp<f16>(c, "le.f0a"), *f0b = wsp<f16>(c, "le.f0b"), *fea0 = wsp<f16>(c, "le.fea0"), *up3 = wsp<f16>(c, "le.up3");
    bool head_fused = false;
    // conv_first .. down_conv1 in one row-streaming launch (le_rows.hip) when the shapes are even and every layer is fp16 or a W8A8
    // layer the kernel runs as fake-quant (variant le_rows_fq)
    if (q.ok() && c->var.at("le_rows") && !c->var.at("no_c3fuse") && !c->var.at("conv32_old") && !(H & 1) && !(W & 1) && q.rows_fit(H, W)) {
        RowsHeadParams p;
        memset(&p, 0, sizeof p);
        bool qh, qd, qs, qi = isq8("LE.conv_first");
        const ConvLayer *Lh = q.rows_conv("LE.HR_conv1", p.fq_y, qh), *Ld = q.rows_conv("LE.down_conv1", p.fq_f, qd);
        const SftLayer &S1 = c->sft.at("LE.SFT_layer1");
        auto i3 = c->c3.find(qi ? "le.conv_first#fq" : "le.conv_first");
        if (Lh && Ld && q.rows_sft(S1, p.fq_s, qs) && i3 != c->c3.end() && (!qi || c->var.at("le_rows_fq"))) {
            if (qi) p.fq_img = Seq::fqp(c->q8.at("LE.conv_first").q);
            p.fq = (qi ? 1 : 0) | (qh ? 2 : 0) | (qd ? 4 : 0) | (qs ? 8 : 0);
            p.img = img; p.cond = cond1; p.H = H; p.W = W; p.fea0 = fea0; p.fea1 = wsp<f16>(c, "le.fea1a");
            p.c3_wfrag = wtp<f16>(c, i3->second.wfrag);
            p.sft_wfrag = wtp<f16>(c, S1.wfrag); p.sft_bias = wtp<float>(c, S1.bias);
            p.w_hr = wtp<f16>(c, Lh->wpk); p.b_hr = wtp<float>(c, Lh->shift); p.w_down = wtp<f16>(c, Ld->wpk); p.b_down = wtp<float>(c, Ld->shift);
            p.trash = const_cast<char *>(wtp<char>(c, c->dump_off));
            p.dump = q.stamp_buf();
            const double npx = (double)H * W;
            q.chk(le_head_rows_launch(p, c->n_cu, q.s), "LE.head", p.fq ? "le_head_rows<fq>" : "le_head_rows",
                  npx * (27.0 * 32 + 2.0 * (16 * 16 + 16 * 32) + 32.0 * 9 * 32 + 32.0 * 9 * 32 / 4), npx * (6 + 32 + 64 + 16) + 2.0 * 2 * 9 * 32 * 32);
            head_fused = true;
        }
    }
    if (!head_fused) {
        if (isq8("LE.conv_first")) {
            if (c->var.at("no_c3q8")) {                  // developer A/B switch: the generic two-launch form     // the 3 planes as NHWC int8 codes (3 of 32 bytes real), then the generic int8 conv
                int8_t *img32 = wsp<int8_t>(c, "le8.img32");
                const QLayer &Lq = c->q8.at("LE.conv_first");
                if (q.ok()) q.chk(planar3_to_q8_launch(img, (size_t)H * W, Lq.q.inv(), Lq.q.zoff(), img32, q.s), "le.conv_first.pack", "planar3_to_q8", 0.0, 38.0 * H * W);
                q.convq8("LE.conv_first", img32, true, 32, H, W, ACT_RELU, f0a, 32, nullptr);
            } else if (q.ok()) {      // quantised while the patch is staged, K = (ky | kx4, c4): two int8 MFMAs per 32 pixels (conv_c3_q8)
                const QLayer &Lq = c->q8.at("LE.conv_first#c3");
                q.chk(conv_c3_q8_launch(img, H, W, wtp<int8_t>(c, Lq.wpk8), wtp<float>(c, Lq.scale), wtp<float>(c, Lq.shift), Lq.q.inv(),
                                        Lq.q.zoff(), ACT_RELU, f0a, c->n_cu, q.s),
                      "LE.conv_first", "conv_c3_q8", (double)H * W * 27 * 32, (double)H * W * (6.0 + 64.0));
            }
            q.conv32("LE.HR_conv1", f0a, cond1, "LE.SFT_layer1", H, W, ACT_RELU, ST_NHWC, fea0, 32, H, W);
        } else {
            // fp16 conv_first is computed inside HR_conv1's kernel from the three planes (conv32s.hip, C3): its 32-channel output
            // (0.53 GB at 4K, written and read back) never exists.  Variants no_c3fuse / conv32_old: the two-launch form
            // (developer A/B switches); a W8A8 HR_conv1 behind an fp16 conv_first has no fused kernel.
            const bool fuse = !c->var.at("no_c3fuse") && !c->var.at("conv32_old") && c->q32.find("LE.HR_conv1") == c->q32.end();
            if (fuse) {
                q.conv32("LE.HR_conv1", f0a, cond1, "LE.SFT_layer1", H, W, ACT_RELU, ST_NHWC, fea0, 32, H, W, nullptr, nullptr, nullptr, nullptr,
                         img, "le.conv_first");
            } else {
                q.c3("le.conv_first", img, H, W, ACT_RELU, f0a, nullptr);
                q.conv32("LE.HR_conv1", f0a, cond1, "LE.SFT_layer1", H, W, ACT_RELU, ST_NHWC, fea0, 32, H, W);
            }
        }
    }
    f16 *fea1a = wsp<f16>(c, "le.fea1a"), *fea1 = wsp<f16>(c, "le.fea1"), *l1b = wsp<f16>(c, "le.l1b");
    auto down = [&](const char *key, const f16 *src, int Hi, int Wi, f16 *dst, int Ho, int Wo) {
        if (isq8(key)) q.convq8(key, src, false, 32, Hi, Wi, ACT_RELU, dst, 32, nullptr);
        else q.conv(key, src, 32, nullptr, 0, Hi, Wi, ACT_RELU, ST_NHWC, dst, 32, Ho, Wo);
    };
    if (!head_fused) down("LE.down_conv1", fea0, H, W, fea1a, s.H1, s.W1);
    q.resblock("LE.recon_trunk1.0", fea1a, cond2, s.H1, s.W1, l1b, fea1);
    f16 *fea2a = wsp<f16>(c, "le.fea2a"), *fea2 = wsp<f16>(c, "le.fea2"), *l2b = wsp<f16>(c, "le.l2b");
    down("LE.down_conv2", fea1, s.H1, s.W1, fea2a, s.H2, s.W2);
    q.resblock("LE.recon_trunk2.0", fea2a, cond3, s.H2, s.W2, l2b, fea2);
    f16 *fea3 = wsp<f16>(c, "le.fea3"), *l3b = wsp<f16>(c, "le.l3b"), *t3x = wsp<f16>(c, "le.t3x"), *t3y = wsp<f16>(c, "le.t3y");
    down("LE.down_conv3", fea2, s.H2, s.W2, fea3, s.H3, s.W3);
    q.resblock("LE.recon_trunk3.0", fea3, cond4, s.H3, s.W3, l3b, t3x);
    q.resblock("LE.recon_trunk3.1", t3x, cond4, s.H3, s.W3, l3b, t3y);
    q.resblock("LE.recon_trunk3.2", t3y, cond4, s.H3, s.W3, l3b, t3x);
    q.resblock("LE.recon_trunk3.3", t3x, cond4, s.H3, s.W3, l3b, t3y, fea3);   // "+ fea3" (line 180) fused as 2nd residual
    // up path: relu(shuffle(conv)) + skip, cropped to the skip's size (_align_to)
    f16 *up1 = wsp<f16>(c, "le.up1"), *t4 = wsp<f16>(c, "le.t4");
    q.conv32("LE.up_conv1.0", t3y, nullptr, "", s.H3, s.W3, ACT_RELU, ST_PS, up1, 32, s.H2, s.W2, fea2);
    q.resblock("LE.recon_trunk4.0", up1, cond3, s.H2, s.W2, l2b, t4);
    f16 *up2 = wsp<f16>(c, "le.up2"), *t5 = wsp<f16>(c, "le.t5");
    q.conv32("LE.up_conv2.0", t4, nullptr, "", s.H2, s.W2, ACT_RELU, ST_PS, up2, 32, s.H1, s.W1, fea1);
    q.resblock("LE.recon_trunk5.0", up2, cond2, s.H1, s.W1, l1b, t5);
    // the full-resolution tail: one row-streaming launch (le_rows.hip) when the shapes are even and every layer is fp16 or a W8A8
    // layer the kernel runs as fake-quant, else per layer
    if (q.ok() && c->var.at("le_rows") && !(H & 1) && !(W & 1) && s.H1 * 2 == H && s.W1 * 2 == W && q.rows_fit(H, W)) {
        RowsTailParams p;
        memset(&p, 0, sizeof p);
        bool qu, qh, ql, qs;
        const ConvLayer *Lu = q.rows_conv("LE.up_conv3.0", p.fq_u, qu), *Lh = q.rows_conv("LE.HR_conv2", p.fq_y, qh),
                        *Ll = q.rows_conv("LE.conv_last", p.fq_z, ql);
        const SftLayer &S2 = c->sft.at("LE.SFT_layer2");
        if (Lu && Lh && Ll && q.rows_sft(S2, p.fq_s, qs)) {
            p.fq = (qu ? 1 : 0) | (qh ? 2 : 0) | (ql ? 4 : 0) | (qs ? 8 : 0);
            p.u = t5; p.fea0 = fea0; p.cond = cond1; p.res_planar = img; p.dst_planar = out_planar; p.H = H; p.W = W;
            p.w_up = wtp<f16>(c, Lu->wpk); p.b_up = wtp<float>(c, Lu->shift);
            p.sft_wfrag = wtp<f16>(c, S2.wfrag); p.sft_bias = wtp<float>(c, S2.bias);
            p.w_hr = wtp<f16>(c, Lh->wpk); p.b_hr = wtp<float>(c, Lh->shift); p.w_last = wtp<f16>(c, Ll->wpk); p.b_last = wtp<float>(c, Ll->shift);
            p.trash = const_cast<char *>(wtp<char>(c, c->dump_off));
            p.dump = q.stamp_buf();
            const double npx = (double)H * W;
            q.chk(le_tail_rows_launch(p, c->n_cu, q.s), "LE.tail", p.fq ? "le_tail_rows<fq>" : "le_tail_rows",
                  npx * (32.0 * 9 * 128 / 4 + 2.0 * (16 * 16 + 16 * 32) + 32.0 * 9 * 32 + 32.0 * 9 * 3), npx * (16 + 64 + 32 + 6 + 6) + 2.0 * 9 * 32 * (128 + 32 + 32));
            return q.rc;
        }
    }
    q.conv32("LE.up_conv3.0", t5, nullptr, "", s.H1, s.W1, ACT_RELU, ST_PS, up3, 32, H, W, fea0);
    q.conv32("LE.HR_conv2", up3, cond1, "LE.SFT_layer2", H, W, ACT_RELU, ST_NHWC, f0b, 32, H, W);
    q.conv32("LE.conv_last", f0b, nullptr, "", H, W, ACT_NONE, ST_PLANAR3, nullptr, 0, H, W, nullptr, nullptr, out_planar, img);
    return q.rc;
}

// HG_Composite.forward tail + Hallucination_Generator.forward
int run_hg(hdrtv_ctx *c, Seq &q, const f16 *base, void *out, int out_f32)
{
    const Shapes s = shapes_for(c->H, c->W);
    const int Hp = s.Hp, Wp = s.Wp;
    f16 *img = wsp<f16>(c, "hg.img");
    uint8_t *mask = wsp<uint8_t>(c, "hg.mask");
    q.chk(hg_prep_launch(base, s.H, s.W, Hp, Wp, img, mask, c->mask_r, 0.1f, q.s), "hg_prep", "hg_prep", 0.0, 13.0 * Hp * Wp);
    float *part = wsp<float>(c, "hg.part");
    // conv1: only the pooled map is kept; its kernel also leaves conv10's second half (the 64 -> 3 sums over conv1's channels) per
    // pixel, so the tail is a per-pixel kernel.  Variant final_recompute (developer A/B switch): the tail recomputes conv1
    // instead (hg_final_fused) -- same arithmetic, same results.
    const bool light = !c->var.at("final_recompute");
    float *part2 = light ? wsp<float>(c, "hg.part2") : nullptr;
    const f16 *w2frag = light ? wtp<f16>(c, c->hgf_wfrag) + 6 * 64 * 8 : nullptr;      // fragments 6..9 of the tail's set
    if (c->hg_i8) {
        int8_t *p1q = wsp<int8_t>(c, "hg8.p1");
        // the fp16 -> int8 boundary costs no pass of its own: conv1 stores the codes its reader conv2 wants
        q.c3("hg.conv1", img, Hp, Wp, ACT_RELU, nullptr, reinterpret_cast<f16 *>(p1q), c->hg_q0_inv, c->hg_q0_zero, w2frag, part2);
        // W8A8 checkpoint: conv2 .. conv9 on int8 MFMA, every activation between conv1 and conv9 one int8 tensor
        int8_t *c2q = wsp<int8_t>(c, "hg8.conv2"), *p3 = wsp<int8_t>(c, "hg8.p3"), *c3 = wsp<int8_t>(c, "hg8.conv3_2"),
               *p4 = wsp<int8_t>(c, "hg8.p4"), *c4 = wsp<int8_t>(c, "hg8.conv4_2"), *p5 = wsp<int8_t>(c, "hg8.p5"),
               *c5 = wsp<int8_t>(c, "hg8.conv5_2"), *pc = wsp<int8_t>(c, "hg8.pc"), *code = wsp<int8_t>(c, "hg8.conv_code2"),
               *u1 = wsp<int8_t>(c, "hg8.up1"), *c6 = wsp<int8_t>(c, "hg8.conv6"), *u2 = wsp<int8_t>(c, "hg8.up2"),
               *c7 = wsp<int8_t>(c, "hg8.conv7"), *u3 = wsp<int8_t>(c, "hg8.up3"), *c8 = wsp<int8_t>(c, "hg8.conv8"),
               *u4q = wsp<int8_t>(c, "hg8.up4"), *c9q = wsp<int8_t>(c, "hg8.conv9");
        q.conv8("hg.conv2", p1q, 64, nullptr, 0, Hp / 2, Wp / 2, ST_NHWC, c2q, 128, Hp / 2, Wp / 2);
        q.conv8("hg.conv3_1", c2q, 128, nullptr, 0, Hp / 2, Wp / 2, ST_POOL, p3, 256, Hp / 4, Wp / 4);
        q.conv8("hg.conv3_2", p3, 256, nullptr, 0, Hp / 4, Wp / 4, ST_NHWC, c3, 256, Hp / 4, Wp / 4);
        q.conv8("hg.conv4_1", c3, 256, nullptr, 0, Hp / 4, Wp / 4, ST_POOL, p4, 512, Hp / 8, Wp / 8);
        q.conv8("hg.conv4_2", p4, 512, nullptr, 0, Hp / 8, Wp / 8, ST_NHWC, c4, 512, Hp / 8, Wp / 8);
        q.conv8("hg.conv5_1", c4, 512, nullptr, 0, Hp / 8, Wp / 8, ST_POOL, p5, 512, Hp / 16, Wp / 16);
        q.conv8("hg.conv5_2", p5, 512, nullptr, 0, Hp / 16, Wp / 16, ST_NHWC, c5, 512, Hp / 16, Wp / 16);
        q.conv8("hg.conv_code1", c5, 512, nullptr, 0, Hp / 16, Wp / 16, ST_POOL, pc, 512, Hp / 32, Wp / 32);
        q.conv8("hg.conv_code2", pc, 512, nullptr, 0, Hp / 32, Wp / 32, ST_NHWC, code, 512, Hp / 32, Wp / 32);
        q.conv8("hg.Up_conv1", code, 512, nullptr, 0, Hp / 32, Wp / 32, ST_PS, u1, 512, Hp / 16, Wp / 16);
        q.conv8("hg.conv6", u1, 512, c5, 512, Hp / 16, Wp / 16, ST_NHWC, c6, 512, Hp / 16, Wp / 16);
        q.conv8("hg.Up_conv2", c6, 512, nullptr, 0, Hp / 16, Wp / 16, ST_PS, u2, 512, Hp / 8, Wp / 8);
        q.conv8("hg.conv7", u2, 512, c4, 512, Hp / 8, Wp / 8, ST_NHWC, c7, 256, Hp / 8, Wp / 8);
        q.conv8("hg.Up_conv3", c7, 256, nullptr, 0, Hp / 8, Wp / 8, ST_PS, u3, 256, Hp / 4, Wp / 4);
        q.conv8("hg.conv8", u3, 256, c3, 256, Hp / 4, Wp / 4, ST_NHWC, c8, 128, Hp / 4, Wp / 4);
        q.conv8("hg.Up_conv4", c8, 128, nullptr, 0, Hp / 4, Wp / 4, ST_PS, u4q, 128, Hp / 2, Wp / 2);
        q.conv8("hg.conv9", u4q, 128, c2q, 128, Hp / 2, Wp / 2, ST_NHWC, c9q, 64, Hp / 2, Wp / 2);
        // Up_conv5 -> pixel shuffle -> ReLU -> first half of conv10, fused: 3 partial sums per pixel leave the kernel
        q.conv8("hg.Up_conv5", c9q, 64, nullptr, 0, Hp / 2, Wp / 2, ST_PS_DOT3, nullptr, 64, Hp, Wp, wtp<float>(c, c->hg_w10a), part);
    } else {
        f16 *p1 = wsp<f16>(c, "hg.p1"), *c2 = wsp<f16>(c, "hg.conv2"), *u4 = wsp<f16>(c, "hg.up4"), *c9 = wsp<f16>(c, "hg.conv9");
        q.c3("hg.conv1", img, Hp, Wp, ACT_RELU, nullptr, p1, 0.f, 0.f, w2frag, part2);
        q.conv("hg.conv2", p1, 64, nullptr, 0, Hp / 2, Wp / 2, ACT_RELU, ST_NHWC, c2, 128, Hp / 2, Wp / 2);
        f16 *p3 = wsp<f16>(c, "hg.p3"), *c3 = wsp<f16>(c, "hg.conv3_2"), *p4 = wsp<f16>(c, "hg.p4"), *c4 = wsp<f16>(c, "hg.conv4_2"),
            *p5 = wsp<f16>(c, "hg.p5"), *c5 = wsp<f16>(c, "hg.conv5_2"), *pc = wsp<f16>(c, "hg.pc"), *code = wsp<f16>(c, "hg.conv_code2");
        f16 *u1 = wsp<f16>(c, "hg.up1"), *c6 = wsp<f16>(c, "hg.conv6"), *u2 = wsp<f16>(c, "hg.up2"), *c7 = wsp<f16>(c, "hg.conv7"),
            *u3 = wsp<f16>(c, "hg.up3"), *c8 = wsp<f16>(c, "hg.conv8");
        q.conv("hg.conv3_1", c2, 128, nullptr, 0, Hp / 2, Wp / 2, ACT_RELU, ST_POOL, p3, 256, Hp / 4, Wp / 4);
        q.conv("hg.conv3_2", p3, 256, nullptr, 0, Hp / 4, Wp / 4, ACT_RELU, ST_NHWC, c3, 256, Hp / 4, Wp / 4);
        q.conv("hg.conv4_1", c3, 256, nullptr, 0, Hp / 4, Wp / 4, ACT_RELU, ST_POOL, p4, 512, Hp / 8, Wp / 8);
        q.conv("hg.conv4_2", p4, 512, nullptr, 0, Hp / 8, Wp / 8, ACT_RELU, ST_NHWC, c4, 512, Hp / 8, Wp / 8);
        q.conv("hg.conv5_1", c4, 512, nullptr, 0, Hp / 8, Wp / 8, ACT_RELU, ST_POOL, p5, 512, Hp / 16, Wp / 16);
        q.conv("hg.conv5_2", p5, 512, nullptr, 0, Hp / 16, Wp / 16, ACT_RELU, ST_NHWC, c5, 512, Hp / 16, Wp / 16);
        q.conv("hg.conv_code1", c5, 512, nullptr, 0, Hp / 16, Wp / 16, ACT_RELU, ST_POOL, pc, 512, Hp / 32, Wp / 32);
        q.conv("hg.conv_code2", pc, 512, nullptr, 0, Hp / 32, Wp / 32, ACT_RELU, ST_NHWC, code, 512, Hp / 32, Wp / 32);
        q.conv("hg.Up_conv1", code, 512, nullptr, 0, Hp / 32, Wp / 32, ACT_RELU, ST_PS, u1, 512, Hp / 16, Wp / 16);
        q.conv("hg.conv6", u1, 512, c5, 512, Hp / 16, Wp / 16, ACT_NONE, ST_NHWC, c6, 512, Hp / 16, Wp / 16);
        q.conv("hg.Up_conv2", c6, 512, nullptr, 0, Hp / 16, Wp / 16, ACT_RELU, ST_PS, u2, 512, Hp / 8, Wp / 8);
        q.conv("hg.conv7", u2, 512, c4, 512, Hp / 8, Wp / 8, ACT_NONE, ST_NHWC, c7, 256, Hp / 8, Wp / 8);
        q.conv("hg.Up_conv3", c7, 256, nullptr, 0, Hp / 8, Wp / 8, ACT_RELU, ST_PS, u3, 256, Hp / 4, Wp / 4);
        q.conv("hg.conv8", u3, 256, c3, 256, Hp / 4, Wp / 4, ACT_NONE, ST_NHWC, c8, 128, Hp / 4, Wp / 4);
        q.conv("hg.Up_conv4", c8, 128, nullptr, 0, Hp / 4, Wp / 4, ACT_RELU, ST_PS, u4, 128, Hp / 2, Wp / 2);
        q.conv("hg.conv9", u4, 128, c2, 128, Hp / 2, Wp / 2, ACT_NONE, ST_NHWC, c9, 64, Hp / 2, Wp / 2);
        // Up_conv5 -> pixel shuffle -> ReLU -> first half of conv10, fused: 3 partial sums per pixel leave the kernel
        q.conv("hg.Up_conv5", c9, 64, nullptr, 0, Hp / 2, Wp / 2, ACT_RELU, ST_PS_DOT3, nullptr, 64, Hp, Wp, nullptr, nullptr, nullptr,
               nullptr, nullptr, wtp<float>(c, c->hg_w10a), part);
    }
    if (!q.ok()) return q.rc;
    const C3Layer &L1 = c->c3.at("hg.conv1");
    HgFinalFusedArgs fa;
    fa.img = img; fa.mask = mask; fa.part = part; fa.wfrag = wtp<f16>(c, c->hgf_wfrag);
    fa.scale = wtp<float>(c, L1.scale); fa.shift = wtp<float>(c, L1.shift);
    fa.b10 = wtp<float>(c, c->f32v.at("hg.b10"));
    fa.wl = wtp<float>(c, c->f32v.at("hg.wl")); fa.bl = wtp<float>(c, c->f32v.at("hg.bl"));
    fa.out = out; fa.out_f32 = out_f32; fa.H = s.H; fa.W = s.W; fa.Hp = Hp; fa.Wp = Wp;
    if (light)
        q.chk(hg_final_light_launch(fa, part2, q.s), "hg_final", "hg_final_light", (double)s.H * s.W * 6 * 3,
              (double)s.H * s.W * (6 + 1 + 32 + 3 * (out_f32 ? 4 : 2)));
#ifdef HDRTV_AB
    else
        q.chk(hg_final_fused_launch(fa, c->n_cu, q.s), "hg_final", "hg_final_fused", (double)Hp * Wp * (128 * 3 + 6 * 3),
              (double)s.H * s.W * (6 + 1 + 16 + 3 * (out_f32 ? 4 : 2)));
#endif
    return q.rc;
}

}  // namespace

// =========================================================================== exported C ABI
extern "C" {

const char *hdrtv_version(void) { return "hdrtv_mi355x 0.1 gfx950 (MFMA f16 implicit-GEMM, hand-written HIP)"; }

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
    c->n_cu = prop.multiProcessorCount;
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
    if (c->ws.dev) (void)hipFree(c->ws.dev);
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

int hdrtv_infer(hdrtv_ctx *c, void *stream, const void *rgb, const void *cond, int H, int W, void *out, int out_dtype,
                void *agcm_out)
{
    if (!c || !rgb || !cond || !out) return fail(c, HDRTV_EINVAL, "null argument");
    if (c->H != H || c->W != W || !c->ws.dev) return fail(c, HDRTV_ESTATE, "call hdrtv_reserve(%d,%d) first", H, W);
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
    if (!c || c->ring.empty()) return fail(c, HDRTV_ESTATE, "ring not created");
    std::unique_lock<std::mutex> lk(c->ring_mu);
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
        const bool got = c->ring_cv.wait_for(lk, std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms),
                                             [&] { return (i = pick()) >= 0; });
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
    }
    return HDRTV_OK;
}

int hdrtv_ring_wait(hdrtv_ctx *c, int slot)
{
    if (!c) return HDRTV_EINVAL;
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(c->ring_mu);
        if (slot < 0 || slot >= (int)c->ring.size()) return fail(c, HDRTV_EINVAL, "bad ring slot");
        // an acquired-but-uncommitted slot has no copy in flight: its event was never recorded (or belongs to the slot's
        // previous frame) and hipEventSynchronize would return at once on stale pixels
        if (c->ring[slot].state != 2) return fail(c, HDRTV_ESTATE, "ring slot %d is not committed (state %d)", slot, c->ring[slot].state);
        ev = c->ring[slot].ev;
    }
    const hipError_t e = hipEventSynchronize(ev);          // outside the lock: blocks until the copy has landed
    if (e != hipSuccess) {
        std::lock_guard<std::mutex> lk(c->ring_mu);
        return fail(c, HDRTV_EHIP, "ring wait failed: %s", hipGetErrorString(e));
    }
    return HDRTV_OK;
}

int hdrtv_ring_release(hdrtv_ctx *c, int slot)
{
    if (!c) return HDRTV_EINVAL;
    {
        std::lock_guard<std::mutex> lk(c->ring_mu);
        if (slot < 0 || slot >= (int)c->ring.size()) return fail(c, HDRTV_EINVAL, "bad ring slot");
        if (c->ring[slot].state == 0) return fail(c, HDRTV_ESTATE, "ring slot %d is already free", slot);
        c->ring[slot].state = 0;
    }
    c->ring_cv.notify_all();
    return HDRTV_OK;
}

int hdrtv_ring_destroy(hdrtv_ctx *c)
{
    if (!c) return HDRTV_OK;
    std::lock_guard<std::mutex> lk(c->ring_mu);
    for (auto &sl : c->ring) {
        if (sl.ev) (void)hipEventDestroy(sl.ev);
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.dev) (void)hipFree(sl.dev);
    }
    c->ring.clear();
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

const char *hdrtv_last_error(const hdrtv_ctx *c) { return c ? c->err.c_str() : "null context"; }

}  // extern "C"
