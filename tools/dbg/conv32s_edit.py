"""Source edits of conv32s.hip for tools/dbg/slp_bisect.sh experiments.  usage: conv32s_edit.py <path to conv32s.hip> fma_h4|fmaf|both"""
import re, sys
p, what = sys.argv[1], sys.argv[2]
s = open(p).read()
if what in ("fma_h4", "both"):
    old = "                    y = y * s1p[gi][qd] + s0p[gi][qd];"
    assert old in s
    s = s.replace(old, "                    y = __builtin_elementwise_fma(y, s1p[gi][qd], s0p[gi][qd]);")
if what in ("fmaf", "both"):
    tok = r"[A-Za-z_][A-Za-z_0-9]*(?:\[[^\]]*\])*(?:\.[xyzw])?"
    pat = re.compile(r"((?:\(float\))?" + tok + r") \* (" + tok + r") \+ (" + tok + r")")
    L = s.split("\n")
    for i, ln in enumerate(L):
        if ("hacc[4 * g" in ln or "a1[4 * g" in ln or "a2[4 * g" in ln or "act_fast(acc[g]" in ln) and " * " in ln:
            L[i] = pat.sub(r"__builtin_fmaf(\1, \2, \3)", ln)
    s = "\n".join(L)
if what == "cvtpk":
    # f32 -> f16 through the packed convert (two values per instruction), as le_rows_i8.hip's cvt4: the scalar casts let hipcc fuse a
    # preceding FMA and the conversion into v_fma_mixlo_f16
    old = "                for (int k = 0; k < 4; ++k) { s1p[gi][qd][k] = (f16)sc[gi][4 * qd + k]; s0p[gi][qd][k] = (f16)sh[gi][4 * qd + k]; }"
    assert old in s
    s = s.replace("#pragma unroll\n" + old, old)
    s = s.replace(old, "                { s1p[gi][qd] = cvt_h4(sc[gi][4 * qd], sc[gi][4 * qd + 1], sc[gi][4 * qd + 2], sc[gi][4 * qd + 3]); s0p[gi][qd] = cvt_h4(sh[gi][4 * qd], sh[gi][4 * qd + 1], sh[gi][4 * qd + 2], sh[gi][4 * qd + 3]); }")
    s = s.replace("namespace {", """typedef float f32x2_ __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x4 cvt_h4(float a, float b, float c, float d)
{
    const f16x2_ lo = __builtin_convertvector(f32x2_{a, b}, f16x2_), hi = __builtin_convertvector(f32x2_{c, d}, f16x2_);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3);
}
namespace {""", 1)
open(p, "w").write(s)
