// api_pack.hip -- the reference's state_dict -> MFMA operand layouts in the weight arena (hdrtv_create).
// Only build_weights() is visible to the other translation units (api.h); the per-layer packers are file-local.
#include "api.h"

namespace hdrtv_host {
namespace {

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
bool pack_conv32_i8(hdrtv_ctx *c, const Pack &pk, const std::string &key, int co, int ps_cps, const std::string &store_as = "", int stride = 1)
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
    L.q = r.q; L.cin = 32; L.cout = co; L.coutPad = coP; L.ks = 3; L.stride = stride;
    L.wpk8 = c->wts.put(wp.data(), wp.size());
    L.scale = c->wts.put(scale.data(), scale.size() * 4);
    L.shift = c->wts.put(shift.data(), shift.size() * 4);
    c->q32[store_as.empty() ? key : store_as] = L;
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

}  // namespace

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
    // A W8A8 layer (weight_int8 + x_scale in the pack: the reference's `predequantize` off) is packed for the int8-MFMA kernel that
    // serves it; the pack is rejected when there is none -- no layer is silently computed in fp16 without its quantiser.  The 3x3
    // 32-channel layers are ALSO packed as dequantised fp16 weights ("#fq"): a fused LE chain that mixes W8A8 and other layers (the
    // mixed recipe) runs them in W8A8Conv2d.forward's own fake-quant form (le_rows.hip, variant le_rows_fq); chains whose layers are
    // all W8A8 run on int8 MFMA (le_rows_i8.hip).  Which form ran is on the launch profile (kernel tags <i8> / <fq>), never assumed.
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
            // down_conv1 inside the int8 head row kernel (le_rows_i8.hip) reads the code ring HR_conv1's epilogue writes: the K order
            // of pack_conv32_i8 (its tables are conv_q8's: the same taps fall outside the image per border class)
            if (std::string(s.name) == "LE.down_conv1" && !pack_conv32_i8(c, hr, s.name, s.co, 0, "LE.down_conv1#rows8", 2)) return false;
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

}  // namespace hdrtv_host
