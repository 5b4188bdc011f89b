"""Compiler-output contracts two kernels rely on (hipcc cross-compiles gfx950 without a GPU).

conv3x3_pglds.hip counts vector-memory operations by hand: its `s_waitcnt vmcnt(N)` after a tile boundary
assumes every wave issued exactly NStores<MODE>::N global stores in the epilogue.  If a compiler change merged
two stores (fewer than assumed) the wait would let a weight DMA be read before it landed, so the count is pinned
here.  conv32p.hip issues its planar-residual loads as inline asm (the compiler does not know they are
outstanding): the loaded registers must not be read before the `s_waitcnt vmcnt(0)` that precedes barrier 1."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "hdr-realtime-video-pipeline_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _asm(src, tmp_path):
    out = tmp_path / (src + ".s")
    # -DHDRTV_AB: the A/B library's superset (the shipped kernels + the superseded ones the GPU bit-identity tests run)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "-fno-vectorize", "-DHDRTV_AB", "-S", "--cuda-device-only",
                    os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
        kernels[m.group(1)] = m.group(2)
    return kernels


def test_pglds_store_counts_match_the_counted_waits(tmp_path):
    kernels = _asm("conv3x3_pglds.hip", tmp_path)
    src = open(os.path.join(CSRC, "conv3x3_pglds.hip")).read()
    m = re.search(r"MODE == ST_POOL \? (\d+) : \(MODE == ST_PS_DOT3 \? (\d+) : (\d+)\)", src)
    n_pool, n_dot3, n_other = (int(v) for v in m.groups())
    expect = {0: n_other, 1: n_other, 2: n_pool, 4: n_dot3}          # ST_NHWC, ST_PS, ST_POOL, ST_PS_DOT3 (common.h)
    seen = 0
    for name, body in kernels.items():
        km = re.search(r"conv_pglds_kernelILi(\d+)E", name)
        if not km:
            continue
        mode = int(km.group(1))
        stores = len(re.findall(r"^\s*(?:global|buffer|flat)_store", body, re.M))
        assert stores == expect[mode], (name, stores, expect[mode])
        waits = set(int(v) for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
        assert expect[mode] + 2 in waits, (name, sorted(waits))
        seen += 1
    assert seen == 4


def test_pglds_i8_store_counts_match_the_counted_waits(tmp_path):
    """The int8 twin (conv3x3_pglds_i8.hip) keeps the same hand-counted scheme with its own store counts."""
    kernels = _asm("conv3x3_pglds_i8.hip", tmp_path)
    src = open(os.path.join(CSRC, "conv3x3_pglds_i8.hip")).read()
    m = re.search(r"N = \(MODE == ST_POOL \|\| MODE == ST_PS_DOT3\) \? (\d+) : (\d+);", src)
    n_pool, n_other = (int(v) for v in m.groups())
    seen = 0
    for name, body in kernels.items():
        km = re.search(r"conv_pglds_i8_kernelILi(\d+)ELb\d", name)
        if not km:
            continue
        want = n_pool if int(km.group(1)) in (2, 4) else n_other
        # stores per tile; hipcc may peel the first tile off the tile loop, which puts a second copy of the epilogue into the text
        stores = len(re.findall(r"^\s*(?:global|buffer|flat)_store", body, re.M))
        assert stores in (want, 2 * want), (name, stores, want)
        waits = set(int(v) for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
        assert want + 2 in waits, (name, sorted(waits))
        assert "v_mfma_i32_16x16x64_i8" in body and "scratch_" not in body
        seen += 1
    assert seen == 5


def test_conv32p_asm_loads_are_read_only_after_the_wait(tmp_path):
    kernels = _asm("conv32p.hip", tmp_path)
    checked = 0
    for name, body in kernels.items():
        if "conv32p_kernel" not in name:
            continue
        lines = body.split("\n")
        for i, ln in enumerate(lines):
            m = re.match(r"\s*global_load_ushort (v\d+),", ln)
            if not m:
                continue
            reg = m.group(1)
            waited = False
            for later in lines[i + 1:]:
                if re.search(r"s_waitcnt vmcnt\(0\)", later):
                    waited = True
                if re.search(rf"\b{reg}\b", later) and not re.match(r"\s*global_load_ushort", later):
                    assert waited, (name, reg, later.strip())
                    break
            checked += 1
    assert checked >= 2


def test_conv32s_counted_waits(tmp_path):
    """conv32s.hip (one barrier per tile) waits for the next tile's LDS-DMA by COUNT: per tile and wave it issues exactly
    NPIECES LDS-DMA pieces (buffer_load_dwordx4 ... lds) and then, on each of the two phase orders, NSTORE global stores, and the loop's closing wait is
    vmcnt(NSTORE).  hipcc's own vmcnt(0) (residuals, first nameable LDS read with a DMA in flight) must sit in the epilogue,
    behind every MFMA of the tile -- in the MFMA or SFT phase it would stall the wave on the DMA it has just issued."""
    kernels = _asm("conv32s.hip", tmp_path)
    seen = 0
    for name, body in kernels.items():
        m = re.search(r"conv32s_kernelILb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
        if not m:
            continue
        sft, i8, sq, planar, c3, split = (int(v) for v in m.groups())
        # DMA pieces of an issuing wave (0-3): halo tile (not with conv_first fused in: computed in the kernel) + condition tile
        npieces, nstore = (0 if c3 else 6) + (3 if sft else 0), 3 if planar else 2
        assert "scratch_" not in body, name
        lines = [ln for ln in body.split("\n") if ln.strip() and not ln.strip().startswith(";")]
        # the steady-state loop = the basic blocks hipcc labels as belonging to the loop that holds the barrier
        groups, cur = {}, None
        for ln in lines:
            lm = re.match(r"\.L(BB\d+_\d+):", ln)
            if lm:
                hm = re.search(r"Header=(BB\d+_\d+)", ln)
                cur = hm.group(1) if hm else (lm.group(1) if "Loop Header" in ln else None)
            if cur:
                groups.setdefault(cur, []).append(ln)
        main = [g for g in groups.values() if any(re.match(r"\s*s_barrier", ln) for ln in g)]
        assert len(main) == 1, (name, len(main))
        loop = main[0]
        # the issue exists twice in the code, for interior tiles (no per-lane image test) and for border tiles; a wave runs one
        # (hipcc may merge the tails of the two paths: between one and two copies of every piece in the text)
        assert npieces <= len([ln for ln in loop if re.search(r"buffer_load_dwordx4 .* lds", ln)]) <= 2 * npieces, name
        # NSTORE per phase order; the epilogue is shared by both orders
        nst = len([ln for ln in loop if re.match(r"\s*global_store", ln)])
        if split:
            # role split: waves 0-3 convolve two 32-pixel groups (twice the stores), waves 4-7 issue the DMA and run the P pass.
            # The conv waves close the tile on lgkmcnt(0) only (they issue no DMA, stores are never waited for), the prep
            # waves on vmcnt(0) lgkmcnt(0) (their DMA has landed)
            assert nst == 2 * nstore, (name, nst)
            closing = [ln for ln in loop if re.search(r"s_waitcnt vmcnt\(0\) lgkmcnt\(0\)", ln)]
            assert closing, name
            seen += 1
            continue
        assert nst == nstore, (name, nst)
        bar = max(i for i, ln in enumerate(loop) if re.match(r"\s*s_barrier", ln))
        loop = loop[bar + 1:] + loop[:bar + 1]        # hipcc may rotate the loop: read it from behind the barrier to the barrier
        bar = len(loop) - 1
        assert re.search(rf"s_waitcnt vmcnt\({nstore}\) lgkmcnt\(0\)", loop[bar - 1]), (name, loop[bar - 1])
        w0 = [i for i, ln in enumerate(loop) if re.search(r"s_waitcnt vmcnt\(0\)", ln)]
        assert len(w0) == 1 and w0[0] < bar, (name, w0)
        assert not any("v_mfma" in ln for ln in loop[w0[0]:bar]), name
        seen += 1
    assert seen == 13


def test_glds1p_store_count_matches_the_counted_wait(tmp_path):
    """conv_glds1p (persistent 1x1 fuse convs) waits vmcnt(6 + P_NST) at a tile's first chunk: P_NST stores per wave and tile."""
    kernels = _asm("conv1x1_glds.hip", tmp_path)
    src = open(os.path.join(CSRC, "conv1x1_glds.hip")).read()
    nst = int(re.search(r"constexpr int P_NST = (\d+);", src).group(1))
    body = [b for n, b in kernels.items() if "conv_glds1p_kernel" in n]
    assert len(body) == 1
    body = body[0]
    assert len(re.findall(r"^\s*(?:global|buffer|flat)_store", body, re.M)) == nst
    waits = set(int(v) for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
    assert {6, 6 + nst} <= waits, sorted(waits)
    assert "scratch_" not in body


def test_prw_has_no_scratch_and_the_waits_it_counts_on(tmp_path):
    """conv3x3_prw.hip sits at ~250 VGPRs with LDS-DMA in flight all the time: a single spill puts scratch loads (and the
    vmcnt(0) hipcc guards them with) into the MFMA stream.  Its only counted waits are for the halo pieces of tap 1
    (A_PIECES_PER_WAVE, + 1 with the scale/shift piece): pin them, the DMA piece counts and the MFMA count per kernel."""
    kernels = _asm("conv3x3_prw.hip", tmp_path)
    src = open(os.path.join(CSRC, "conv3x3_prw.hip")).read()
    assert "A_PIECES_PER_WAVE = (A_PIECES + 7) / 8" in src and "(TH + 2) * HW" in src
    seen = 0
    for name, body in kernels.items():
        km = re.search(r"conv_prw_kernelILi(\d+)ELi(\d+)E", name)
        if not km:
            continue
        th = int(km.group(2))
        per_wave = (((th + 2) * 18 * 128 + 1023) // 1024 + 7) // 8        # halo pieces per wave: 6 (16-row tiles) or 3
        assert "scratch_" not in body, name
        dot3 = 2 * th if km.group(1) == "4" else 0            # ST_PS_DOT3: the 64 -> 3 dot products of a pixel row = one MFMA per weight half
        assert len(re.findall(r"v_mfma_f32_16x16x32_f16", body)) == 9 * 4 * th + dot3, name  # nine unrolled taps x (2 x TH x 2 k-steps)
        # (PS_DOT3 builds its dot-product fragments from global weights in the prologue: counted waits in front of the first barrier)
        loop = body.split("s_barrier", 1)[1] if dot3 else body
        waits = set(int(v) for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)", loop))
        assert waits == {0, per_wave, per_wave + 1}, (name, sorted(waits))
        assert len(re.findall(r"s_barrier", body)) == 2, name                               # prologue + one per chunk (tap 8)
        # LDS-DMA must be the BUFFER form: with the FLAT-encoded global_load_lds in a kernel hipcc stops counting and every
        # wait becomes lgkmcnt(0) / vmcnt(0)
        assert "global_load_lds" not in body and len(re.findall(r"buffer_load_dwordx4 .* lds", body)) >= 36 + per_wave, name
        counted = [int(v) for v in re.findall(r"lgkmcnt\((\d+)\)", body)]
        assert sum(v > 0 for v in counted) > sum(v == 0 for v in counted), name
        seen += 1
    assert seen == 7          # NHWC / PS / POOL x {16, 8}-row tiles + PS_DOT3


@pytest.mark.parametrize("src", ["conv3x3_prw.hip", "conv3x3_pglds.hip", "conv3x3_pglds_i8.hip", "conv1x1_glds.hip", "conv_i8_misc.hip",
                                 "conv3x3s2_preg.hip", "conv32s.hip", "conv32p.hip", "conv_tile_f16.hip", "le_rows.hip"])
def test_lds_dma_kernels_spill_nothing_and_use_the_buffer_form(src, tmp_path):
    """Every kernel that stages through LDS-DMA: (1) no scratch -- a scratch load with a DMA in flight is guarded by
    vmcnt(0), i.e. it drains the DMA queue in the middle of the pipeline (round 2's conv3x3s2_preg<12> spilled 3 VGPRs);
    (2) the DMA is `buffer_load_dwordx4 ... lds`, never the FLAT-encoded global_load_lds, after which hipcc's waitcnt pass
    stops counting."""
    out = tmp_path / (src + ".s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "-fno-vectorize", "-DHDRTV_AB", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)],
                   check=True, capture_output=True)
    text = out.read_text()
    spills = [int(v) for v in re.findall(r"\.vgpr_spill_count:\s*(\d+)", text)]
    assert spills and max(spills) == 0, (src, spills)
    assert "global_load_lds" not in text and re.search(r"buffer_load_dwordx4 .* lds", text), src
    assert not re.search(r"^\s*scratch_", text, re.M), src


def test_le_rows_step_loops_never_drain_the_dma_queue(tmp_path):
    """le_rows.hip keeps LDS-DMA two to three steps (4-6 rows) ahead of its consumers and counts its waits by hand.  (1) Every
    s_waitcnt vmcnt inside a step loop must be one of the counted ones: hipcc guards an LDS load that carries no alias metadata
    (HIP's struct vector types) and a plain global load's first use with vmcnt(0) while a DMA is in flight, which would drain
    the prefetch queue every step -- the kernels read LDS through clang ext_vector types only and fetch the tail's residual
    planes by DMA for that reason.  (2) The DMA-issuing role issues exactly the piece count its closing wait assumes, on every
    control-flow path (hipcc turns the wave-uniform row tests into branches).  (3) MFMA counts, no scratch."""
    kernels = _asm("le_rows.hip", tmp_path)
    # kernel -> (MFMAs in the text, counted closing waits of its DMA role(s))
    # (the <.., true> instances -- W8A8 layers as fake-quant -- carry the SFT hidden-layer MFMA twice where a layer's input may
    # or may not be quantised, the mixed recipe: a wave-uniform branch)
    want = {"le_rb_rows_kernel": ((2 * 18 + 2 * 3, 2 * 18 + 2 * 3 + 4), {12}), "le_tail_rows_kernel": ((3 * 18 + 3, 3 * 18 + 3 + 2), {12}),
            "le_head_rows_kernel": ((2 * 18 + 3 + 3, 2 * 18 + 3 + 3 + 2), {8})}
    seen = 0
    for name, body in kernels.items():
        key = next((k for k in want if k in name), None)
        if key is None:
            continue
        n_mfma, counted = want[key]
        n_mfma = n_mfma[1 if "Lb1E" in name else 0]
        assert "scratch_" not in body and "global_load_lds" not in body, name
        assert len(re.findall(r"v_mfma_f32_32x32x16_f16", body)) == n_mfma, (name, len(re.findall(r"v_mfma_f32_32x32x16_f16", body)))
        lines = [ln for ln in body.split("\n") if ln.strip() and not ln.strip().startswith(";")]
        # step loops = the loops that hold an s_barrier
        groups, cur = {}, None
        for ln in lines:
            lm = re.match(r"\.L(BB\d+_\d+):", ln)
            if lm:
                hm = re.search(r"Header=(BB\d+_\d+)", ln)
                cur = hm.group(1) if hm else (lm.group(1) if "Loop Header" in ln else None)
            if cur:
                groups.setdefault(cur, []).append(ln)
        loops = [g for g in groups.values() if any(re.match(r"\s*s_barrier", ln) for ln in g)]
        assert len(loops) == 2, (name, len(loops))               # one per role
        waits = set()
        for loop in loops:
            w = [int(v) for ln in loop for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)", ln)]
            if key != "le_head_rows_kernel" or any("buffer_load" in ln for ln in loop):
                # (the head's first role stages the image patch with plain loads and issues no DMA: hipcc's own waits are right there)
                waits |= set(w)
        assert waits == counted, (name, sorted(waits), sorted(counted))
        seen += 1
    assert seen == 6          # three kernels x {fp16, fake-quant} instances


def test_prw_dot3_strip_reads_stay_behind_the_flag_poll(tmp_path):
    """conv_prw<ST_PS_DOT3>: the even wave polls an LDS flag and then reads its partner's partial sums from the partner's
    strip with plain loads.  An acquire fence behind both polls keeps hipcc from hoisting those reads (or the odd wave's DMA
    into the strip) above the loop; here: no LDS read between a poll and the loop's closing branch, and the strip reads
    (TH / 4 ds_read_b96: three of a float4) behind it."""
    kernels = _asm("conv3x3_prw.hip", tmp_path)
    body = [b for n, b in kernels.items() if re.search(r"conv_prw_kernelILi4ELi16E", n)]
    assert len(body) == 1
    lines = [ln.strip() for ln in body[0].split("\n") if ln.strip() and not ln.strip().startswith(";")]
    polls = [i for i, ln in enumerate(lines) if ln.startswith("s_sleep")]
    assert len(polls) == 2, polls
    for i in polls:
        j = next(k for k in range(i, len(lines)) if lines[k].startswith("s_cbranch_execnz"))
        assert not any(ln.startswith("ds_read") for ln in lines[i:j]), lines[i:j]
    j = next(k for k in range(polls[1], len(lines)) if lines[k].startswith("s_cbranch_execnz"))
    nxt = lines[j + 1:j + 40]
    assert sum(ln.startswith("ds_read_b96") or ln.startswith("ds_read_b128") for ln in nxt) >= 4, nxt      # (x, y, z of a float4: b96)


def test_fp32_matrix_pipe_conv_issues_its_mfmas_and_spills_nothing(tmp_path):
    """conv_f32_mfma (fp32_ops.hip): 72 (3x3, 16-channel chunks) / 16 (1x1, 32-channel chunks) v_mfma_f32_32x32x2_f32 per chunk and
    accumulator tile in the unrolled chunk loop, staged through registers without scratch (an array of HIP's float4 struct for the
    weight staging landed in scratch: clang vectors only)."""
    kernels = _asm("fp32_ops.hip", tmp_path)
    seen = 0
    for name, body in kernels.items():
        m = re.search(r"conv_f32_mfma_kernelILi(\d)ELi(\d)E", name)
        if not m:
            continue
        mt, ks = int(m.group(1)), int(m.group(2))
        assert len(re.findall(r"v_mfma_f32_32x32x2_f32", body)) == (72 if ks == 3 else 16) * mt, name
        assert "scratch_" not in body, name
        seen += 1
    assert seen == 4


def test_le_rows_i8_step_loops_keep_their_counted_waits(tmp_path):
    """le_rows_i8.hip: the DMA-issuing role of each kernel closes its step with exactly the counted vmcnt (never a drain), every
    role's step loop holds its int8 MFMAs (9 per conv, 3 per SFT layer, 2 for conv_first), nothing spills."""
    kernels = _asm("le_rows_i8.hip", tmp_path)
    want = {"le_rb_rows_i8_kernel": (2 * (9 + 3), {12}), "le_tail_rows_i8_kernel": (9 + 3 + 2 * 9, {10}), "le_head_rows_i8_kernel": (2 + 3 + 9 + 9, {8})}
    seen = 0
    for name, body in kernels.items():
        key = next((k for k in want if k in name), None)
        if key is None:
            continue
        n_mfma, counted = want[key]
        assert "scratch_" not in body and "global_load_lds" not in body, name
        assert len(re.findall(r"v_mfma_i32_32x32x32_i8", body)) == n_mfma, (name, len(re.findall(r"v_mfma_i32_32x32x32_i8", body)))
        lines = [ln for ln in body.split("\n") if ln.strip() and not ln.strip().startswith(";")]
        groups, cur = {}, None
        for ln in lines:
            lm = re.match(r"\.L(BB\d+_\d+):", ln)
            if lm:
                hm = re.search(r"Header=(BB\d+_\d+)", ln)
                cur = hm.group(1) if hm else (lm.group(1) if "Loop Header" in ln else None)
            if cur:
                groups.setdefault(cur, []).append(ln)
        loops = [g for g in groups.values() if any(re.match(r"\s*s_barrier", ln) for ln in g)]
        assert len(loops) == 2, (name, len(loops))
        waits = set()
        for loop in loops:
            if any("buffer_load" in ln and " lds" in ln for ln in loop):          # the role that issues LDS-DMA
                waits |= {int(v) for ln in loop for v in re.findall(r"s_waitcnt vmcnt\((\d+)\)", ln)}
        assert waits == counted, (name, sorted(waits), sorted(counted))
        seen += 1
    assert seen == 3


def test_no_packed_f32_arithmetic_in_any_kernel(tmp_path):
    """csrc/Makefile builds with -fno-slp-vectorize -fno-vectorize: packed f32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, which only the
    SLP and loop vectorisers produce from this code) returned a stale operand in its low half when the wave shared a SIMD with another
    kernel's MFMA waves (pre_fused beside conv1x1_i8 of a frame in flight on another stream: NOTEBOOK.md round 5, section 8).  The
    flag must stay in the Makefile and must keep having that effect on every kernel file."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    assert re.search(r"^CXXFLAGS \+= -fno-slp-vectorize", mk, re.M)
    seen = 0
    # (fp32_ops.hip keeps the vectorisers -- its vector-FMA convolution runs on v_pk_fma_f32 -- and its contexts take one lane only)
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and not f.startswith("api_") and f not in ("hdrtv_api.hip", "fp32_graph.hip", "fp32_ops.hip")):
        for name, body in _asm(src, tmp_path).items():
            packed = re.findall(r"^\s*v_pk_(?:mul|add|fma)_f32", body, re.M)
            assert not packed, (src, name, len(packed))
            seen += 1
    assert seen >= 50
