// api_util.hip -- error reporting and the developer-variant table of a context (api.h).
#include "api.h"

namespace hdrtv_host {

int fail(hdrtv_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) { std::lock_guard<std::mutex> lk(c->err_mu); c->err = buf; }
    return code;
}

// ---- developer variants: name -> default.  hdrtv_set_variant changes one on a context; HDRTV_VARIANTS="a=1,b=0" seeds them at
// hdrtv_create.  The launch path reads c->var only.
namespace {
const std::pair<const char *, int> k_variants[] = {
    {"le_rows", 1},          // fused row-streaming LE kernels (le_rows.hip); 0 = the per-layer 16x16-tile kernels
    {"le_rows_min", 8},      // ... when a strip segment has at least this many rows (it pays 4 .. 6 warm-up rows; 8: the 1/8-resolution ResBlocks of a
                             // 3840x2160 frame -- 9-row segments -- still gain 10 - 15 % over their two per-layer launches, profiles/r05_le_rows_min_ab.txt)
    {"le_rows_i8", 1},       // ... chains whose layers are all W8A8 on the int8-MFMA row kernels (le_rows_i8.hip); 0 = the fake-quant form (le_rows_fq) or the per-layer int8 kernels
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
    {"cond3_fused", 1},      // CondNet3.4 in CondNet3.2's (conv3x3s2_preg<64>) epilogue; 0 = its own conv_igemm launch
    {"cond2_fused", 1},      // CondNet2.2 + .4 in conv3x3s2_preg<192>'s epilogue; 0 = the separate cond_tail launch
    {"pre_split", 0},        // 1: preprocess as two kernels (unpack, condition resize)
    {"force_ncu", 0},        // > 0: pretend the device has this many CUs (persistent grids)
    {"f32_narrow_below", 0}, // precision="fp32": workgroups per CU below which conv_f32 runs 8 channels per lane (0 = 3)
    {"f32_mfma", 1},         // precision="fp32": 3x3 / stride-1 layers on v_mfma_f32_32x32x2_f32 (conv_f32_mfma); 0 = every layer on the vector-FMA kernel
};
// variants whose non-default settings select kernels that exist in the A/B library only (make AB=1 -> libhdrtv_mi355x_ab.so):
// superseded schedules kept as bit-identity yardsticks of the shipped ones
const char *const k_ab_only[] = {"conv32_old", "conv32_nosplit", "conv32_nw", "glds1_old", "final_recompute"};
}  // namespace
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

}  // namespace hdrtv_host
