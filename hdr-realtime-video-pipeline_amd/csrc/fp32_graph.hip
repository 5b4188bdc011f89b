// fp32_graph.hip -- precision="fp32": weights and the layer graph (kernels: fp32_ops.hip; interface: api.h).
//
// The reference's fp32 preset (hdrtvnet_torch.py:1694-1712) runs the very same modules on fp32 tensors.  Here the graph is
// spelled out layer by layer on planar CHW fp32 tensors -- one generic convolution kernel with the per-layer epilogue the
// module applies (BatchNorm / GFM / activation / residual / PixelShuffle), plus a handful of element-wise kernels -- in the
// reference's operation order, so that the result tracks the reference's own CPU fp32 run to fp32 rounding
// (tests/test_gpu_fp32.py: floats <= 2e-5, almost every RGB48 integer identical).
#include "api.h"

namespace hdrtv_host {

namespace {
struct F32T {
    float *p = nullptr;
    int C = 0, H = 0, W = 0;
    size_t n() const { return (size_t)C * H * W; }
};

// every 4-D ".weight" of a pack becomes a conv layer [cout group][cin][tap][cot]; BatchNorm2d(eval) vectors of HG's conv
// blocks (name.0 conv, name.1 BN: Hallucination_arch.py:24-29) become scale / shift; 1-D / 2-D tensors are kept raw
bool pack_f32(hdrtv_ctx *c, const Pack &pk, const std::string &prefix)
{
    for (const auto &kv : pk.e) {
        const std::string &name = kv.first;
        const PackEntry &pe = kv.second;
        if (pe.dtype == 2) { c->err = "INT8 checkpoint with precision 'fp32' (" + name + "): use precision='int8-full'/'int8-mixed'"; return false; }
        if (pe.dtype == 3) continue;                               // num_batches_tracked
        std::vector<float> v;
        if (!pk.get(name, pe.numel(), v, c->err)) return false;
        const std::string suffix = ".weight";
        const bool is_w = name.size() > suffix.size() && name.compare(name.size() - suffix.size(), suffix.size(), suffix) == 0;
        if (!(is_w && pe.ndim == 4)) {
            c->f32v[prefix + name] = c->wts.put(v.data(), v.size() * 4);
            continue;
        }
        F32Layer L;
        L.cout = pe.dims[0]; L.cin = pe.dims[1]; L.ks = pe.dims[2];
        if (pe.dims[3] != L.ks || (L.ks != 1 && L.ks != 3)) { c->err = "unsupported kernel shape: " + name; return false; }
        L.cot = L.cout <= 8 ? 8 : 32;
        const int groups = (L.cout + L.cot - 1) / L.cot, kk = L.ks * L.ks;
        std::vector<float> w((size_t)groups * L.cin * kk * L.cot, 0.f);
        for (int co = 0; co < L.cout; ++co)
            for (int ci = 0; ci < L.cin; ++ci)
                for (int t = 0; t < kk; ++t)
                    w[(((size_t)(co / L.cot) * L.cin + ci) * kk + t) * L.cot + co % L.cot] = v[((size_t)co * L.cin + ci) * kk + t];
        L.w = c->wts.put(w.data(), w.size() * 4);
        const std::string layer = name.substr(0, name.size() - suffix.size());
        std::vector<float> b;
        if (!pk.get(layer + ".bias", (size_t)L.cout, b, c->err)) return false;
        L.b = c->wts.put(b.data(), b.size() * 4);
        if (layer.size() > 2 && layer.compare(layer.size() - 2, 2, ".0") == 0) {
            const std::string bn = layer.substr(0, layer.size() - 2) + ".1";
            if (pk.has(bn + ".running_mean")) {
                std::vector<float> g, be, mu, var, s(L.cout), t(L.cout);
                if (!pk.get(bn + ".weight", (size_t)L.cout, g, c->err) || !pk.get(bn + ".bias", (size_t)L.cout, be, c->err) ||
                    !pk.get(bn + ".running_mean", (size_t)L.cout, mu, c->err) || !pk.get(bn + ".running_var", (size_t)L.cout, var, c->err))
                    return false;
                for (int i = 0; i < L.cout; ++i) {
                    const double sc = (double)g[i] / std::sqrt((double)var[i] + 1e-5);
                    s[i] = (float)sc;
                    t[i] = (float)((double)be[i] - (double)mu[i] * sc);
                }
                L.bn_s = c->wts.put(s.data(), s.size() * 4);
                L.bn_t = c->wts.put(t.data(), t.size() * 4);
                L.bn = true;
            }
        }
        c->conv32f[prefix + layer] = L;
    }
    return true;
}

}  // namespace

bool build_weights_f32(hdrtv_ctx *c, const Pack &hr, const Pack *hg)
{
    const std::vector<unsigned char> z(256, 0);
    c->zeros_off = c->wts.put(z.data(), z.size());
    c->dump_off = c->wts.put(z.data(), z.size());
    if (!pack_f32(c, hr, "")) return false;
    if (hg && !pack_f32(c, *hg, "hg.")) return false;
    static const char *need[] = {"AGCM.conv_first", "AGCM.HRconv", "AGCM.conv_last", "AGCM.classifier.model.20", "LE.conv_first",
                                 "LE.HR_conv1", "LE.conv_last", "LE.cond_first.0", "LE.CondNet4.4", "LE.recon_trunk3.3.conv2"};
    for (const char *n : need)
        if (!c->conv32f.count(n)) { c->err = std::string("tensor missing from weight pack: ") + n + ".weight"; return false; }
    if (hg && (!c->conv32f.count("hg.conv1.0") || !c->conv32f.count("hg.conv_last") || !c->conv32f.count("hg.Up_conv5.0"))) {
        c->err = "HG weight pack is not a Hallucination_Generator state";
        return false;
    }
    return true;
}

namespace {
struct F32Run {
    hdrtv_ctx *c;
    Seq *q;
    bool plan;                          // true: only register the tensors (hdrtv_reserve); false: launch
    bool ok() const { return plan || q->ok(); }
    hipStream_t s() const { return q->s; }

    F32T T(const std::string &name, int C, int H, int W)
    {
        F32T t;
        t.C = C; t.H = H; t.W = W;
        if (plan) {
            auto it = c->t.find(name);
            if (it == c->t.end()) ws_add(c, name, C, H, W, 2);
            else if (it->second.C != C || it->second.H != H || it->second.W != W) { fprintf(stderr, "hdrtv: fp32 plan: %s re-declared with another shape\n", name.c_str()); abort(); }
            return t;
        }
        t.p = wsp<float>(c, name);
        return t;
    }
    // scratch shared by every use of the same shape (the stream orders them)
    F32T tmp(const char *tag, int C, int H, int W)
    {
        char nm[96];
        snprintf(nm, sizeof nm, "f32.%s.%dx%dx%d", tag, C, H, W);
        return T(nm, C, H, W);
    }
    const float *vec(const std::string &name)
    {
        auto it = c->f32v.find(name);
        if (it == c->f32v.end()) { if (!plan && q->ok()) q->rc = fail(c, HDRTV_ESTATE, "no fp32 vector %s", name.c_str()); return nullptr; }
        return wtp<float>(c, it->second);
    }

    struct Opt {
        int stride = 1, act = 0;
        float slope = 0.f;
        const F32T *x2 = nullptr;        // channel concat
        const float *res = nullptr;
        bool ps = false;
        const float *gfm_s = nullptr, *gfm_t = nullptr;
    };
    F32T conv(const std::string &layer, const std::string &out_name, const F32T &x, const Opt &o)
    {
        F32T y;
        auto it = c->conv32f.find(layer);
        if (it == c->conv32f.end()) { if (ok() && !plan) q->rc = fail(c, HDRTV_ESTATE, "no fp32 layer %s", layer.c_str()); return y; }
        const F32Layer &L = it->second;
        const int pad = L.ks / 2;
        const int Ho = (x.H + 2 * pad - L.ks) / o.stride + 1, Wo = (x.W + 2 * pad - L.ks) / o.stride + 1;
        y = o.ps ? T(out_name, L.cout / 4, 2 * Ho, 2 * Wo) : T(out_name, L.cout, Ho, Wo);
        if (plan || !q->ok()) return y;
        if (x.C + (o.x2 ? o.x2->C : 0) != L.cin || (o.x2 && (o.x2->H != x.H || o.x2->W != x.W))) {
            q->rc = fail(c, HDRTV_ESTATE, "fp32 layer %s: input shape mismatch", layer.c_str());
            return y;
        }
        F32ConvParams p;
        memset(&p, 0, sizeof p);
        p.x0 = x.p; p.c0 = x.C; p.x1 = o.x2 ? o.x2->p : nullptr; p.c1 = o.x2 ? o.x2->C : 0;
        p.Hi = x.H; p.Wi = x.W; p.Ho = Ho; p.Wo = Wo;
        p.w = wtp<float>(c, L.w); p.bias = wtp<float>(c, L.b);
        if (L.bn) { p.bn_s = wtp<float>(c, L.bn_s); p.bn_t = wtp<float>(c, L.bn_t); }
        p.gfm_s = o.gfm_s; p.gfm_t = o.gfm_t; p.res = o.res; p.y = y.p;
        p.narrow_below = c->var.at("f32_narrow_below");
        p.no_mfma = c->var.at("f32_mfma") ? 0 : 1;
        p.cout = L.cout; p.cot = L.cot; p.pad = pad; p.act = o.act; p.slope = o.slope; p.ps = o.ps ? 1 : 0;
        const double macs = (double)L.cout * L.cin * L.ks * L.ks * Ho * Wo;
        const bool mfma = o.stride == 1 && conv_f32_on_mfma(p, L.ks, c->n_cu);
        q->chk(conv_f32_launch(p, L.ks, o.stride, c->n_cu, q->s), layer.c_str(), mfma ? "conv_f32_mfma" : "conv_f32", macs,
               4.0 * ((double)L.cin * x.H * x.W + (double)L.cout * Ho * Wo));
        return y;
    }
    static Opt act(int a, float slope = 0.f, int stride = 1)
    {
        Opt o;
        o.act = a; o.slope = slope; o.stride = stride;
        return o;
    }
    void ew(int op, const char *label, const F32T &a, const float *b, const float *cc, const F32T &y)
    {
        if (plan || !q->ok()) return;
        q->chk(ew_f32_launch(op, a.p, b, cc, y.p, a.n(), q->s), label, "ew_f32", 0.0, 4.0 * a.n() * (op ? 4 : 3));
    }
    F32T add(const char *label, const std::string &out_name, const F32T &a, const F32T &b)
    {
        F32T y = T(out_name, a.C, a.H, a.W);
        ew(0, label, a, b.p, nullptr, y);
        return y;
    }
    // HDRUNet3T1._align_to (HDRUNet3T1_arch.py:79-104); a no-op when the shapes already agree
    F32T align(const char *label, const std::string &out_name, const F32T &x, int H, int W)
    {
        if (x.H == H && x.W == W) return x;
        F32T y = T(out_name, x.C, H, W);
        if (!plan && q->ok()) q->chk(window_f32_launch(x.p, y.p, x.C, x.H, x.W, H, W, 0, q->s), label, "window_f32");
        return y;
    }
    // SFTLayer.forward (arch_util.py:68-72): x * (scale + 1) + shift, both from two 1x1 convs on the condition map
    F32T sft(const std::string &name, const std::string &out_name, const F32T &x, const F32T &cond)
    {
        const F32Layer *l0 = c->conv32f.count(name + ".SFT_scale_conv0") ? &c->conv32f.at(name + ".SFT_scale_conv0") : nullptr;
        const int hid = l0 ? l0->cout : 1;
        F32T h = conv(name + ".SFT_scale_conv0", tmp_name("sfth", hid, cond.H, cond.W), cond, act(2, 0.1f));
        F32T scale = conv(name + ".SFT_scale_conv1", tmp_name("sfts", x.C, x.H, x.W), h, Opt());
        h = conv(name + ".SFT_shift_conv0", tmp_name("sfth", hid, cond.H, cond.W), cond, act(2, 0.1f));
        F32T shift = conv(name + ".SFT_shift_conv1", tmp_name("sftt", x.C, x.H, x.W), h, Opt());
        F32T y = T(out_name, x.C, x.H, x.W);
        ew(1, name.c_str(), x, scale.p, shift.p, y);
        return y;
    }
    std::string tmp_name(const char *tag, int C, int H, int W)
    {
        char nm[96];
        snprintf(nm, sizeof nm, "f32.%s.%dx%dx%d", tag, C, H, W);
        return nm;
    }
    // ResBlock_with_SFT.forward (arch_util.py:89-95)
    F32T resblock(const std::string &name, const std::string &out_name, const F32T &x, const F32T &cond)
    {
        F32T a = sft(name + ".sft1", tmp_name("rba", x.C, x.H, x.W), x, cond);
        F32T b = conv(name + ".conv1", tmp_name("rbb", x.C, x.H, x.W), a, act(1));
        a = sft(name + ".sft2", tmp_name("rba", x.C, x.H, x.W), b, cond);
        Opt o;
        o.res = x.p;
        return conv(name + ".conv2", out_name, a, o);
    }
};

// Color_Condition.forward + ConditionNet.forward's dynamic branch (Condition_arch.py:19-35, 559-585)
F32T f32_agcm(F32Run &r, const F32T &tensor, const F32T &cond, float *agcm_out)
{
    static const int idx[5] = {0, 4, 8, 12, 16};
    F32T x = cond;
    char nm[96], out[64];
    for (int i = 0; i < 5; ++i) {
        snprintf(nm, sizeof nm, "AGCM.classifier.model.%d", idx[i]);
        snprintf(out, sizeof out, "agcm32.c%d", i);
        F32T t = r.conv(nm, out, x, F32Run::Opt());
        snprintf(out, sizeof out, "agcm32.p%d", i);
        F32T p = r.T(out, t.C, (t.H - 1) / 2 + 1, (t.W - 1) / 2 + 1);
        if (!r.plan && r.q->ok()) {
            r.q->chk(avgpool3s2_leaky_f32_launch(t.p, p.p, t.C, t.H, t.W, 0.2f, r.s()), nm, "avgpool3s2_leaky_f32");
            if (i < 4) {
                snprintf(nm, sizeof nm, "AGCM.classifier.model.%d", idx[i] + 3);
                const float *g = r.vec(std::string(nm) + ".weight"), *b = r.vec(std::string(nm) + ".bias");
                if (r.q->ok()) r.q->chk(instnorm_f32_launch(p.p, g, b, p.C, p.H * p.W, 1e-5f, r.s()), nm, "instnorm_f32");
            }
        }
        x = p;
    }
    F32T f = r.conv("AGCM.classifier.model.20", "agcm32.c5", x, F32Run::Opt());
    F32T fea = r.T("agcm32.fea6", 6, 1, 1);
    F32T gfm = r.T("agcm32.gfm", 6 * 64, 1, 1);
    if (!r.plan && r.q->ok()) {
        r.q->chk(plane_mean_f32_launch(f.p, fea.p, 6, f.H * f.W, r.s()), "AGCM.classifier.mean", "plane_mean_f32");
        F32GfmParams g;
        static const char *heads[6] = {"AGCM.cond_scale_first", "AGCM.cond_scale_HR", "AGCM.cond_scale_last",
                                       "AGCM.cond_shift_first", "AGCM.cond_shift_HR", "AGCM.cond_shift_last"};
        static const int hn[6] = {64, 64, 3, 64, 64, 3};
        for (int i = 0; i < 6; ++i) {
            g.w[i] = r.vec(std::string(heads[i]) + ".weight");
            g.b[i] = r.vec(std::string(heads[i]) + ".bias");
            g.n[i] = hn[i];
        }
        g.fea = fea.p; g.out = gfm.p;
        if (r.q->ok()) r.q->chk(gfm_heads_f32_launch(g, r.s()), "AGCM.cond_scale/shift", "gfm_heads_f32");
    }
    F32Run::Opt o = F32Run::act(1);
    o.gfm_s = gfm.p; o.gfm_t = gfm.p ? gfm.p + 3 * 64 : nullptr;
    F32T a = r.conv("AGCM.conv_first", "agcm32.o1", tensor, o);
    o.gfm_s = gfm.p ? gfm.p + 64 : nullptr; o.gfm_t = gfm.p ? gfm.p + 4 * 64 : nullptr;
    F32T b = r.conv("AGCM.HRconv", "agcm32.o2", a, o);
    o.act = 0;
    o.gfm_s = gfm.p ? gfm.p + 2 * 64 : nullptr; o.gfm_t = gfm.p ? gfm.p + 5 * 64 : nullptr;
    F32T y = r.conv("AGCM.conv_last", "agcm32.out", b, o);
    if (!r.plan && r.q->ok() && agcm_out)
        if (hipMemcpyAsync(agcm_out, y.p, y.n() * 4, hipMemcpyDeviceToDevice, r.s()) != hipSuccess)
            r.q->rc = fail(r.c, HDRTV_EHIP, "fp32 agcm_out copy failed");
    return y;
}

// HDRUNet3T1._forward_safe_aligned (HDRUNet3T1_arch.py:152-206) with x = [img, img] (Ensemble_AGCM_LE_arch.py:890-893)
F32T f32_le(F32Run &r, const F32T &img, const std::string &out_name)
{
    typedef F32Run::Opt Opt;
    const std::string L = "LE.";
    F32T cond = r.conv(L + "cond_first.0", "le32.cfa", img, F32Run::act(2, 0.1f));
    cond = r.conv(L + "cond_first.2", "le32.cfb", cond, F32Run::act(2, 0.1f));
    cond = r.conv(L + "cond_first.4", "le32.cond", cond, F32Run::act(2, 0.1f));
    F32T cn[5];
    static const int strides[5][3] = {{0, 0, 0}, {1, 1, 1}, {2, 1, 1}, {2, 2, 1}, {2, 2, 2}};
    for (int i = 1; i <= 4; ++i) {
        char nm[64], out[64];
        snprintf(nm, sizeof nm, "LE.CondNet%d", i);
        snprintf(out, sizeof out, "le32.cn%da", i);
        F32T y = r.conv(std::string(nm) + ".0", out, cond, F32Run::act(2, 0.1f, strides[i][0]));
        snprintf(out, sizeof out, "le32.cn%db", i);
        y = r.conv(std::string(nm) + ".2", out, y, F32Run::act(2, 0.1f, strides[i][1]));
        snprintf(out, sizeof out, "le32.cond%d", i);
        cn[i] = r.conv(std::string(nm) + ".4", out, y, F32Run::act(0, 0.f, strides[i][2]));
    }
    F32T f = r.conv(L + "conv_first", "le32.cf", img, F32Run::act(1));
    f = r.sft(L + "SFT_layer1", "le32.s1", f, cn[1]);
    F32T fea0 = r.conv(L + "HR_conv1", "le32.fea0", f, F32Run::act(1));
    F32T d1 = r.conv(L + "down_conv1", "le32.d1", fea0, F32Run::act(1, 0.f, 2));
    F32T fea1 = r.resblock(L + "recon_trunk1.0", "le32.fea1", d1, cn[2]);
    F32T d2 = r.conv(L + "down_conv2", "le32.d2", fea1, F32Run::act(1, 0.f, 2));
    F32T fea2 = r.resblock(L + "recon_trunk2.0", "le32.fea2", d2, cn[3]);
    F32T fea3 = r.conv(L + "down_conv3", "le32.fea3", fea2, F32Run::act(1, 0.f, 2));
    F32T out = fea3;
    for (int b = 0; b < 4; ++b) {
        char nm[64];
        snprintf(nm, sizeof nm, "LE.recon_trunk3.%d", b);
        out = r.resblock(nm, (b & 1) ? "le32.t3b" : "le32.t3a", out, cn[4]);
    }
    out = r.add("LE.trunk3+fea3", "le32.t3s", out, fea3);

    Opt up = F32Run::act(1);
    up.ps = true;
    F32T u = r.conv(L + "up_conv1.0", "le32.up1", out, up);
    u = r.align("LE.align1", "le32.up1a", u, fea2.H, fea2.W);
    out = r.resblock(L + "recon_trunk4.0", "le32.o2", r.add("LE.up1+fea2", "le32.s2", u, fea2), cn[3]);
    u = r.conv(L + "up_conv2.0", "le32.up2", out, up);
    u = r.align("LE.align2", "le32.up2a", u, fea1.H, fea1.W);
    out = r.resblock(L + "recon_trunk5.0", "le32.o1", r.add("LE.up2+fea1", "le32.s1b", u, fea1), cn[2]);
    u = r.conv(L + "up_conv3.0", "le32.up3", out, up);
    u = r.align("LE.align3", "le32.up3a", u, fea0.H, fea0.W);
    out = r.sft(L + "SFT_layer2", "le32.s2s", r.add("LE.up3+fea0", "le32.s0", u, fea0), cn[1]);
    out = r.conv(L + "HR_conv2", "le32.hr2", out, F32Run::act(1));
    F32T last = r.conv(L + "conv_last", "le32.last", out, Opt());
    last = r.align("LE.align_out", "le32.lasta", last, img.H, img.W);
    return r.add("LE.img+out", out_name, img, last);
}

// HG_Composite.forward (HG_Composite_arch.py:86-107) + Hallucination_Generator.forward (Hallucination_arch.py:97-137)
void f32_hg(F32Run &r, const F32T &base, float mask_r, float *out)
{
    typedef F32Run::Opt Opt;
    const int H = base.H, W = base.W, Hp = (H + 31) / 32 * 32, Wp = (W + 31) / 32 * 32;
    F32T mask = r.T("hg32.mask", 1, H, W);
    if (!r.plan && r.q->ok()) r.q->chk(hg_mask_f32_launch(base.p, mask.p, (size_t)H * W, mask_r, r.s()), "hg.mask", "hg_mask_f32");
    F32T img = base, m = mask;
    if (Hp != H || Wp != W) {
        img = r.T("hg32.img", 3, Hp, Wp);
        m = r.T("hg32.maskp", 1, Hp, Wp);
        if (!r.plan && r.q->ok()) {
            r.q->chk(window_f32_launch(base.p, img.p, 3, H, W, Hp, Wp, 1, r.s()), "hg.pad", "window_f32");
            r.q->chk(window_f32_launch(mask.p, m.p, 1, H, W, Hp, Wp, 1, r.s()), "hg.pad(mask)", "window_f32");
        }
    }
    auto block = [&](const char *name, const F32T &x) {
        return r.conv(std::string("hg.") + name + ".0", std::string("hg32.") + name, x, F32Run::act(1));
    };
    auto pool = [&](const char *name, const F32T &x) {
        F32T y = r.T(std::string("hg32.") + name, x.C, x.H / 2, x.W / 2);
        if (!r.plan && r.q->ok()) r.q->chk(maxpool2_f32_launch(x.p, y.p, x.C, x.H, x.W, r.s()), name, "maxpool2_f32");
        return y;
    };
    auto upb = [&](const char *name, const F32T &x) {
        Opt o = F32Run::act(1);
        o.ps = true;
        return r.conv(std::string("hg.") + name + ".0", std::string("hg32.") + name, x, o);
    };
    auto fuse = [&](const char *name, const F32T &a, const F32T &b) {
        Opt o;
        o.x2 = &b;
        return r.conv(std::string("hg.") + name, std::string("hg32.") + name, a, o);
    };
    F32T c1 = block("conv1", img);
    F32T c2 = block("conv2", pool("p1", c1));
    F32T c3 = block("conv3_2", pool("p2", block("conv3_1", c2)));
    F32T c4 = block("conv4_2", pool("p3", block("conv4_1", c3)));
    F32T c5 = block("conv5_2", pool("p4", block("conv5_1", c4)));
    F32T code = block("conv_code2", pool("p5", block("conv_code1", c5)));
    F32T c6 = fuse("conv6", upb("Up_conv1", code), c5);
    F32T c7 = fuse("conv7", upb("Up_conv2", c6), c4);
    F32T c8 = fuse("conv8", upb("Up_conv3", c7), c3);
    F32T c9 = fuse("conv9", upb("Up_conv4", c8), c2);
    F32T c10 = fuse("conv10", upb("Up_conv5", c9), c1);
    F32T t = fuse("conv_last", c10, img);
    if (!r.plan && r.q->ok())
        r.q->chk(hg_blend_f32_launch(t.p, img.p, m.p, out, H, W, Hp, Wp, r.s()), "hg.blend", "hg_blend_f32", 0.0, 4.0 * 10 * H * W);
}

// the whole fp32 graph; plan = true registers its tensors in the workspace (hdrtv_reserve), false launches it
}  // namespace

int run_f32(hdrtv_ctx *c, Seq &q, bool plan, int H, int W, const float *rgb, const float *cond, float *out, float *agcm_out)
{
    F32Run r{c, &q, plan};
    const Shapes sh = shapes_for(H, W);
    F32T tensor, cd;
    tensor.p = const_cast<float *>(rgb); tensor.C = 3; tensor.H = sh.H; tensor.W = sh.W;
    cd.p = const_cast<float *>(cond); cd.C = 3; cd.H = sh.h4; cd.W = sh.w4;
    F32T a = f32_agcm(r, tensor, cd, agcm_out);
    F32T base = f32_le(r, a, "le32.out");
    if (c->has_hg) {
        f32_hg(r, base, c->mask_r, out);
    } else if (!plan && q.ok()) {
        if (hipMemcpyAsync(out, base.p, base.n() * 4, hipMemcpyDeviceToDevice, q.s) != hipSuccess)
            q.rc = fail(c, HDRTV_EHIP, "fp32 output copy failed");
    }
    return q.rc;
}

}  // namespace hdrtv_host
