// api_workspace.hip -- the per-resolution workspace arena: every intermediate a named tensor (hdrtv_reserve, hdrtv_get_tap).
#include "api.h"

namespace hdrtv_host {

// ------------------------------------------------------------------------------- workspace
namespace {
inline int half_up(int n) { return (n - 1) / 2 + 1; }

}  // namespace

Tensor &ws_add(hdrtv_ctx *c, const std::string &name, int C, int H, int W, int layout)
{
    Tensor t;
    t.C = C; t.H = H; t.W = W; t.layout = layout;
    t.off = c->ws.reserve(t.bytes() + 256);
    c->t[name] = t;
    return c->t[name];
}

namespace {
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

}  // namespace

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

void free_workspaces(hdrtv_ctx *c)
{
    for (unsigned char *p : c->lane_ws)
        if (p) (void)hipFree(p);
    c->lane_ws.clear();
    c->ws.dev = nullptr;
    c->H = c->W = 0;
}

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
    free_workspaces(c);                    // no valid workspace until every step below has succeeded
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
        if (int rc = f32_plan(c, H, W)) return rc;
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
    // one workspace per lane (hdrtv_set_lanes): lane 0 is initialised, the others start as copies of it
    c->lane_ws.assign((size_t)c->lanes, nullptr);
    for (int l = 0; l < c->lanes; ++l)
        if (hipMalloc((void **)&c->lane_ws[l], c->ws.size + 4096) != hipSuccess) {
            c->lane_ws[l] = nullptr;
            free_workspaces(c);
            return fail(c, HDRTV_ENOMEM, "workspace allocation of %d x %zu bytes failed", c->lanes, c->ws.size);
        }
    c->ws.dev = c->lane_ws[0];
    hipError_t e = hipMemset(c->ws.dev, 0, c->ws.size + 4096);
    auto up = [&](const char *name, const void *src, size_t bytes) {
        if (e == hipSuccess) e = hipMemcpy(wsp<char>(c, name), src, bytes, hipMemcpyHostToDevice);
    };
    up("aa.wx", wx.data(), wx.size() * 4); up("aa.wy", wy.data(), wy.size() * 4);
    up("aa.xmn", xmn.data(), xmn.size() * 4); up("aa.xns", xns.data(), xns.size() * 4);
    up("aa.ymn", ymn.data(), ymn.size() * 4); up("aa.yns", yns.data(), yns.size() * 4);
    for (int l = 1; l < c->lanes && e == hipSuccess; ++l) e = hipMemcpy(c->lane_ws[l], c->ws.dev, c->ws.size + 4096, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) {                 // leave no half-initialised workspace behind a size that looks reserved
        free_workspaces(c);
        return fail(c, HDRTV_EHIP, "workspace initialisation failed: %s", hipGetErrorString(e));
    }
    c->H = H; c->W = W;
    return HDRTV_OK;
}

}  // namespace hdrtv_host
