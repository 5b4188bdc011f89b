// api_graph.hip -- launch sequencing of one hdrtv_infer: AGCM, LE and HG layer by layer (struct Seq: api.h).
#include "api.h"

namespace hdrtv_host {

// ----------------------------------------------------------------------- Seq: one launch per call, counted, checked, profiled
void Seq::conv(const std::string &key, const f16 *src0, int c0, const f16 *src1, int c1, int Hi, int Wi, int act, int mode, f16 *dst, int dstC, int Hd, int Wd, const f16 *res1, const f16 *res2, f16 *dst_full, f16 *dst_planar, const f16 *res_planar, const float *dotw, float *dst_dot, int s0_stride)
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
    p.tail_w = tail_w; p.tail_b = tail_b; p.tail_s = tail_s; p.tail_out = tail_out;
    tail_w = nullptr; tail_b = nullptr; tail_s = nullptr; tail_out = nullptr;
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
    // a fused 1x1 tail (run_le sets it and then skips the tail's own launch) exists in conv3x3s2_preg's epilogue only
    if (p.tail_w && !s2g) { rc = fail(c, HDRTV_ESTATE, "conv %s: a fused CondNet tail was requested but the layer does not run on conv3x3s2_preg", key.c_str()); return; }
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
    else if (s2g) snprintf(tag, sizeof tag, p.tail_w ? "conv3x3s2_preg<%d>+tail" : "conv3x3s2_preg<%d>", L.coutPad);
    else if (pglds) snprintf(tag, sizeof tag, "%s<%s>", prw ? (prw_th == 8 ? "conv_prw8" : "conv_prw") : "conv_pglds", mode == ST_POOL ? "pool" : (mode == ST_PS ? "ps" : (mode == ST_PS_DOT3 ? "ps_dot3" : "nhwc")));
    else if (glds1) snprintf(tag, sizeof tag, "conv_glds1");
    else snprintf(tag, sizeof tag, "conv_igemm<%d,%d,%d,%d>", L.cin_t, L.bn, L.ks, L.stride);
    double macs = (double)p.Ho * p.Wo * L.cin * L.ks * L.ks * L.cout;
    double bytes = 2.0 * Hi * Wi * L.cin + 2.0 * L.ks * L.ks * L.cin * L.coutPad;
    if (s2g && p.tail_w) {          // a fused 1x1 tail: + its MACs, 16 channels out instead of the 64 it consumes
        macs += (double)p.Ho * p.Wo * (p.tail_s ? 64 * 16 : 64 * 64 + 64 * 16);
        bytes -= 2.0 * p.Ho * p.Wo * (64 - 16);
    }
    const double outel = mode == ST_POOL ? (double)Hd * Wd * L.cout + (dst_full ? (double)p.Ho * p.Wo * L.cout : 0.0)
                                          : (mode == ST_PLANAR3 ? 3.0 * Hd * Wd
                                                                : (mode == ST_PS_DOT3 ? 8.0 * Hd * Wd : (double)p.Ho * p.Wo * L.cout));
    bytes += 2.0 * outel * (1 + (res1 ? 1 : 0) + (res2 ? 1 : 0) + (res_planar ? 1 : 0));
    chk(t16 ? conv_t16_launch(p, s, c->n_cu) : s2g ? conv3x3s2_preg_launch(p, c->n_cu, s)
            : (pglds ? (prw ? conv_prw_launch(p, prw_th, c->n_cu, s) : conv_pglds_launch(p, c->n_cu, s))
                     : (glds1 ? conv_glds1_launch(p, s, c->n_cu, c->var.at("glds1_old") != 0) : conv_igemm_launch(p, L.cin_t, L.bn, L.ks, L.stride, s))),
        key.c_str(), tag, macs, bytes);
}


void Seq::conv8(const std::string &key, const int8_t *src0, int c0, const int8_t *src1, int c1, int Hi, int Wi, int mode, void *dst, int dstC, int Hd, int Wd, const float *dotw, float *dst_dot)
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


void Seq::convq8(const std::string &key, const void *src, bool src_i8, int src_stride, int Hi, int Wi, int act, void *dst, int dstC, const ActQf *oq)
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


void Seq::c3(const std::string &key, const f16 *in, int H, int W, int act, f16 *out, f16 *out_pool, float pool_q_inv, float pool_q_zero, const f16 *w2frag, float *part2)
{
    if (!ok()) return;
    const C3Layer &L = c->c3.at(key);
    chk(conv_c3_launch(in, H, W, wtp<f16>(c, L.wfrag), wtp<float>(c, L.scale), wtp<float>(c, L.shift), L.cout, act, out,
                       out_pool, c->n_cu, s, pool_q_inv, pool_q_zero, w2frag, part2), key.c_str(),
        part2 ? "conv_c3<64,dot3>" : (L.cout == 64 ? "conv_c3<64>" : "conv_c3<32>"), (double)H * W * (27 * L.cout + (part2 ? 192 : 0)),
        (double)H * W * (6.0 + (out ? 2.0 * L.cout : 0.0) + (out_pool ? (pool_q_inv > 0.f ? 0.25 : 0.5) * L.cout : 0.0) + (part2 ? 16.0 : 0.0)));
}


void Seq::conv32(const std::string &key, const f16 *src, const f16 *cond, const std::string &sft_key, int H, int W, int act, int mode, f16 *dst, int dstC, int Hd, int Wd, const f16 *res1, const f16 *res2, f16 *dst_planar, const f16 *res_planar, const f16 *c3_img, const std::string &c3_key)
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


void Seq::resblock(const std::string &base, const f16 *x, const f16 *cond, int H, int W, f16 *tb, f16 *y, const f16 *extra)
{
    // every layer of the block W8A8 (the full-QAT recipe): the row-streaming kernel on int8 MFMA (le_rows_i8.hip), the rings hold
    // the layers' int8 codes; bit-identical to the two conv32s<sft-i8, i8> launches at the bottom (variant le_rows_i8 = 0: the
    // fake-quant form below, or those launches)
    if (ok() && c->var.at("le_rows") && c->var.at("le_rows_i8") && !extra && rows_fit(H, W)) {
        auto q1 = c->q32.find(base + ".conv1"), q2 = c->q32.find(base + ".conv2");
        const SftLayer &S1 = c->sft.at(base + ".sft1"), &S2 = c->sft.at(base + ".sft2");
        if (q1 != c->q32.end() && q2 != c->q32.end() && S1.q && S2.q && q1->second.coutPad == 32 && q2->second.coutPad == 32) {
            RowsRbI8Params p;
            memset(&p, 0, sizeof p);
            auto cv = [&](const QLayer &Q) { return RowsConvI8{wtp<int8_t>(c, Q.wpk8), wtp<float>(c, Q.scale), wtp<float>(c, Q.shift), Q.q.inv(), Q.q.zoff()}; };
            auto sv = [&](const SftLayer &S) {
                RowsSftI8 r{wtp<int8_t>(c, S.qfrag), wtp<float>(c, S.qconst), {S.inv[0], S.inv[1]}, {S.zoff[0], S.zoff[1]}, {S.hzoff[0], S.hzoff[1]}};
                return r;
            };
            p.x = x; p.cond = cond; p.c1 = cv(q1->second); p.c2 = cv(q2->second); p.s1 = sv(S1); p.s2 = sv(S2);
            p.slope1 = act_slope(ACT_RELU);
            p.dst = y; p.trash = const_cast<char *>(wtp<char>(c, c->dump_off)); p.H = H; p.W = W;
            const double npx = (double)H * W;
            chk(le_rb_rows_i8_launch(p, c->n_cu, s), base.c_str(), "le_rb_rows<i8>",
                npx * (2.0 * 32 * 9 * 32 + 4.0 * (16 * 16 + 16 * 32)), npx * (64 + 32 + 64) + 2.0 * 9 * 32 * 32);
            return;
        }
    }
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
    bool cond2_fused = false;
    if (c->conv.find("LE.CondNet234.0") != c->conv.end()) {
        f16 *x192 = wsp<f16>(c, "le.x192");
        // CondNet2.2 + .4 ride in that launch's epilogue (variant cond2_fused; a W8A8 CondNet2.4 keeps its own kernel): the first 64 of
        // the 192 channels then never reach HBM
        cond2_fused = c->var.at("cond2_fused") != 0 && !c->tail_q8 && !c->q_tail2.on;
        if (cond2_fused) { q.tail_w = wtp<f16>(c, c->tail_wfrag); q.tail_b = wtp<float>(c, c->tail_bias); q.tail_out = cond2; }
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
    } else if (q.ok() && !cond2_fused)
        q.chk(cond_tail_launch(a2, astride, (size_t)s.H1 * s.W1, wtp<f16>(c, c->tail_wfrag), wtp<float>(c, c->tail_bias), cond2, c->n_cu, q.s,
                               qlast(c->q_tail2, qa2)),
              "LE.CondNet2.2+4", c->q_tail2.on ? "cond_tail<q2>" : "cond_tail", (double)s.H1 * s.W1 * (64 * 64 + 64 * 16), (double)s.H1 * s.W1 * (128 + 32));
    // CondNet3 / CondNet4: .2 (3x3 / stride 2, 64 -> 64, LeakyReLU) then .4 (1x1 resp. 3x3 / stride 2, 64 -> 16)
    {
        const f16 *af[2] = {a3, a4};
        const int8_t *aq[2] = {a3q, a4q};
        f16 *h2[2] = {h2a, h2b}, *cout[2] = {cond3, cond4};
        const char *l2[2] = {"LE.CondNet3.2", "LE.CondNet4.2"}, *l4[2] = {"LE.CondNet3.4", "LE.CondNet4.4"};
        bool tail4 = false;
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
                // CondNet3.4 (1x1, no activation) rides in CondNet3.2's epilogue (variant cond3_fused): its 64-channel input never reaches HBM
                auto t4 = c->conv.find(l4[i]);
                if (i == 0 && c->var.at("cond3_fused") && !isq8(l4[i]) && t4 != c->conv.end() && t4->second.ks == 1 && t4->second.cin == 64 &&
                    t4->second.cout == 16 && t4->second.coutPad == 32 && t4->second.cin_t == 64) {
                    q.tail_w = wtp<f16>(c, t4->second.wpk); q.tail_s = wtp<float>(c, t4->second.scale); q.tail_b = wtp<float>(c, t4->second.shift);
                    q.tail_out = cout[i];
                    tail4 = true;
                }
                q.conv(l2[i], af[i], 64, nullptr, 0, s.H1, s.W1, ACT_LRELU01, ST_NHWC, h2[i], 64, s.H2, s.W2, nullptr, nullptr, nullptr,
                       nullptr, nullptr, nullptr, nullptr, astride);
            }
            if (tail4) { tail4 = false; continue; }
            if (isq8(l4[i])) q.convq8(l4[i], h, h_i8, 64, s.H2, s.W2, ACT_NONE, cout[i], 16, nullptr);
            else q.conv(l4[i], h2[i], 64, nullptr, 0, s.H2, s.W2, ACT_NONE, ST_NHWC, cout[i], 16, i ? s.H3 : s.H2, i ? s.W3 : s.W2);
        }
    }
    // main branch: every SFT is fused into the 3x3 conv that follows it
    f16 *f0a = wsp<f16>(c, "le.f0a"), *f0b = wsp<f16>(c, "le.f0b"), *fea0 = wsp<f16>(c, "le.fea0"), *up3 = wsp<f16>(c, "le.up3");
    bool head_fused = false;
    // conv_first .. down_conv1 in one row-streaming launch (le_rows.hip) when the shapes are even and every layer is fp16 or a W8A8
    // layer the kernel runs as fake-quant (variant le_rows_fq)
    // every layer of the head W8A8 (the full-QAT recipe): the row kernel on int8 MFMA (le_rows_i8.hip), bit-identical to conv_c3_q8 +
    // conv32s<sft-i8, i8> + conv_q8<32,3,2>
    if (q.ok() && c->var.at("le_rows") && c->var.at("le_rows_i8") && !c->var.at("no_c3q8") && !(H & 1) && !(W & 1) && q.rows_fit(H, W)) {
        auto qc = c->q8.find("LE.conv_first#c3");
        auto qh = c->q32.find("LE.HR_conv1"), qd = c->q32.find("LE.down_conv1#rows8");
        const SftLayer &S1 = c->sft.at("LE.SFT_layer1");
        if (qc != c->q8.end() && qh != c->q32.end() && qd != c->q32.end() && S1.q && qh->second.coutPad == 32 && qd->second.coutPad == 32) {
            RowsHeadI8Params p;
            memset(&p, 0, sizeof p);
            auto cv = [&](const QLayer &Q) { return RowsConvI8{wtp<int8_t>(c, Q.wpk8), wtp<float>(c, Q.scale), wtp<float>(c, Q.shift), Q.q.inv(), Q.q.zoff()}; };
            p.img = img; p.cond = cond1; p.fea0 = fea0; p.fea1 = wsp<f16>(c, "le.fea1a"); p.H = H; p.W = W;
            p.cf = cv(qc->second); p.hr = cv(qh->second); p.dn = cv(qd->second);
            p.s = RowsSftI8{wtp<int8_t>(c, S1.qfrag), wtp<float>(c, S1.qconst), {S1.inv[0], S1.inv[1]}, {S1.zoff[0], S1.zoff[1]}, {S1.hzoff[0], S1.hzoff[1]}};
            p.slope_relu = act_slope(ACT_RELU);
            p.trash = const_cast<char *>(wtp<char>(c, c->dump_off));
            const double npx = (double)H * W;
            q.chk(le_head_rows_i8_launch(p, c->n_cu, q.s), "LE.head", "le_head_rows<i8>",
                  npx * (27.0 * 32 + 2.0 * (16 * 16 + 16 * 32) + 32.0 * 9 * 32 + 32.0 * 9 * 32 / 4), npx * (6 + 32 + 64 + 16) + 2.0 * 9 * 32 * 32);
            head_fused = true;
        }
    }
    if (!head_fused && q.ok() && c->var.at("le_rows") && !c->var.at("no_c3fuse") && !c->var.at("conv32_old") && !(H & 1) && !(W & 1) && q.rows_fit(H, W)) {
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
    if (q.ok() && c->var.at("le_rows") && c->var.at("le_rows_i8") && !(H & 1) && !(W & 1) && s.H1 * 2 == H && s.W1 * 2 == W && q.rows_fit(H, W)) {
        // every layer of the tail W8A8 (the full-QAT recipe): the row kernel on int8 MFMA (le_rows_i8.hip), bit-identical to the three
        // per-layer int8 launches at the bottom
        auto qu = c->q32.find("LE.up_conv3.0"), qh = c->q32.find("LE.HR_conv2"), ql = c->q32.find("LE.conv_last");
        const SftLayer &S2 = c->sft.at("LE.SFT_layer2");
        if (qu != c->q32.end() && qh != c->q32.end() && ql != c->q32.end() && S2.q && qu->second.coutPad == 128 &&
            qh->second.coutPad == 32 && ql->second.coutPad == 32) {
            RowsTailI8Params p;
            memset(&p, 0, sizeof p);
            auto cv = [&](const QLayer &Q) { return RowsConvI8{wtp<int8_t>(c, Q.wpk8), wtp<float>(c, Q.scale), wtp<float>(c, Q.shift), Q.q.inv(), Q.q.zoff()}; };
            p.u = t5; p.fea0 = fea0; p.cond = cond1; p.res_planar = img; p.dst_planar = out_planar; p.H = H; p.W = W;
            p.up = cv(qu->second); p.hr = cv(qh->second); p.last = cv(ql->second);
            p.s = RowsSftI8{wtp<int8_t>(c, S2.qfrag), wtp<float>(c, S2.qconst), {S2.inv[0], S2.inv[1]}, {S2.zoff[0], S2.zoff[1]}, {S2.hzoff[0], S2.hzoff[1]}};
            p.slope_relu = act_slope(ACT_RELU);
            p.trash = const_cast<char *>(wtp<char>(c, c->dump_off));
            const double npx = (double)H * W;
            q.chk(le_tail_rows_i8_launch(p, c->n_cu, q.s), "LE.tail", "le_tail_rows<i8>",
                  npx * (32.0 * 9 * 128 / 4 + 2.0 * (16 * 16 + 16 * 32) + 32.0 * 9 * 32 + 32.0 * 9 * 3), npx * (16 + 64 + 32 + 6 + 6) + 9.0 * 32 * (128 + 32 + 32));
            return q.rc;
        }
    }
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

}  // namespace hdrtv_host
