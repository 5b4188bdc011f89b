#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X-native SDR->HDR hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) it launches the N ranks itself as fresh child
processes (python -m torch.distributed.run, before this process has imported torch or touched a GPU) and relays rank 0's
JSON line.

A step = one pass of the hot path over one batch of synthetic frames: one 3840x2160 frame
per GPU (BASELINE.json configs[2]: full HDRTVNet++ AGCM+LE+HG fp16 + fused RGB48 post;
at N > 1 frame i goes to GPU i mod N = configs[3], no data-path collective).  The u8 frames
are resident in HBM before the timed region; per frame the timed work is
pre_unpack + cond_resize + infer(AGCM, LE, HG) + post_rgb48 + the hand-off into the pinned host
RGB48 ring (hipMemcpyAsync on a copy stream + hipEvent; a consumer waits and releases one frame
behind) -- the path BASELINE.json's north_star names.  `value` = N*K frames / max-over-ranks wall
time.  `value_device_only` leaves the RGB48 frame in HBM; `value_pcie_inclusive` also uploads
every input frame from pinned host memory (never `value`: inputs are resident by contract).

Extra objects on the JSON line:
  roofline     dominant kernel (by summed time) of hdrtv_infer, timed with HIP events on the
               launch stream (hdrtv_profile_*), in K extra steps right after the timed region
               (kept out of it so `value` carries no event overhead): achieved = algorithmic
               FLOPs per launch / average launch duration; peak = 2500 TFLOP/s dense fp16 MFMA.
  cpu_baseline the reference's CPU-eager semantics (the oracle's graphs on PyTorch's CPU kernels,
               fp32) timed on this box's host cores on a bounded sample, rank 0 at N=1 only.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
for _p in (REPO, os.path.join(REPO, "hdr-realtime-video-pipeline_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

MFMA_F16_DENSE_PEAK_TFLOPS = 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--no-hg", action="store_true", help="debug: AGCM+LE only (not the headline config)")
    ap.add_argument("--lanes", type=int, default=1, choices=(1, 2),
                    help="frames in flight on the device for the timed region (hdrtv_set_lanes): frame i runs on lane i mod LANES, each lane with "
                         "its own activation workspace and HIP stream.  Default 1: `value` is one frame at a time; the two-lane rate is reported "
                         "beside it as `two_lanes` (with a byte-for-byte self-check)")
    ap.add_argument("--no-two-lanes", action="store_true", help="skip the `two_lanes` leg")
    ap.add_argument("--int8", action="store_true",
                    help="BASELINE configs[4] instead of the headline fp16 configuration: HR from the reference's INT8-QAT checkpoint "
                         "with its W8A8 layers kept quantised (predequantize off) and the HG head as a W8A8 checkpoint (seeded + calibrated: the "
                         "reference's int8 HG weights are not shipped); which kernels ran -- int8 MFMA or fake-quant on fp16 MFMA -- is read "
                         "from the launch profile and printed in `metric` / `executed`")
    ap.add_argument("--int8-recipe", default="full", choices=("full", "mixed"),
                    help="which shipped HR recipe --int8 runs: full (128 W8A8 layers) or mixed (29 W8A8 + 78 W8A16 + 21 fp16)")
    ap.add_argument("--int8-predequantize", action="store_true",
                    help="with --int8: run the HR checkpoint as the reference does on ROCm (int8 storage, fp16 compute)")
    ap.add_argument("--no-int8-extra", action="store_true",
                    help="skip the extra BASELINE configs[4] measurement (INT8-QAT HR + W8A8 HG) reported beside the headline at N=1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency-tail", action="store_true", help="skip the 300 extra frames behind the timed region that give p99 / 1 %% low a population")
    ap.add_argument("--cpu-sample", default="960x540", help="WxH of the plain-C oracle's extra sample")
    ap.add_argument("--cpu-protocol", default="bounded", choices=("bounded", "full", "quick"),
                    help="cpu_baseline: 'full' = SURVEY 8d to the letter (5 warm-up + 20 timed frames at 960x540 and 1920x1080, 2 timed "
                         "frames at the workload size: ~6 minutes); 'bounded' (default, ~2.5 minutes) = 5 + 20 frames at 960x540, 1 + 5 at "
                         "1920x1080 and ONE unscaled frame at the workload size as `value`; 'quick' = 1 + 3 / 0 + 1 frames, scaled")
    ap.add_argument("--no-dispatcher", action="store_true",
                    help="skip the host-fed in-product dispatcher measurement (hdrtv_mi355x/dispatch.py, one worker process) reported "
                         "beside value_pcie_inclusive at N=1")
    ap.add_argument("--dispatcher", action="store_true", help="only the host-fed dispatcher measurement (its own JSON line)")
    ap.add_argument("--dispatcher-depth", type=int, default=2, help="dispatcher worker: frames in flight (upload + compute + download)")
    ap.add_argument("--dispatcher-slots", type=int, default=3, help="dispatcher: input / output slots per worker")
    ap.add_argument("--dispatcher-lanes", type=int, default=1, help="dispatcher worker: compute lanes (the worker's default is 1)")
    ap.add_argument("--dispatcher-sim", action="store_true",
                    help="CPU only: the dispatcher's host side over --gpus N stand-in workers (memcpys + device_ms of sleep per frame); no GPU is touched")
    ap.add_argument("--sim-fps", type=float, nargs="*", help="with --dispatcher-sim: offered rates to run (default: 100 per worker)")
    ap.add_argument("--sim-numa", action="store_true", help="with --dispatcher-sim: workers pin themselves as they would to their GPU's NUMA node")
    ap.add_argument("--layers", action="store_true", help="print the per-layer profile to stderr")
    return ap.parse_args()


def physical_cores():
    """(physical cores this process may run on, CPU model) from /proc/cpuinfo and the affinity mask."""
    model, cores = "unknown", set()
    try:
        allowed = os.sched_getaffinity(0)
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = (t.strip() for t in line.split(":", 1))
                cur[k] = v
                if k == "model name":
                    model = v
            elif not line.strip() and cur:
                if int(cur.get("processor", -1)) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in allowed:
            cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except OSError:
        pass
    return (len(cores) or (os.cpu_count() or 1)), model


def cpu_baseline(args, use_hg):
    """the reference's CPU-eager semantics -- the network graphs on PyTorch's CPU kernels
    (oracle/aten_backend.py; the reference's own Python cannot travel to the GPU box), fp32, inference mode,
    torch.set_num_threads(physical cores used) -- timed per stage (pre / run / post as process_timed does,
    hdrtvnet_torch.py:2380-2395) at 960x540 and 1920x1080.  `--cpu-protocol full` is the protocol to the letter
    (5 warm-up + 20 timed frames at both sizes and 2 timed frames at the workload size, minutes of CPU time); the default
    keeps the 960x540 part, 1 + 5 frames at 1920x1080 and 1 warm-up + 2 timed frames at the workload size (~2.5 min of CPU
    time for 3840x2160 with HG on 16 cores).  The plain-C oracle is
    timed beside it as an extra."""
    import torch
    from hdrtv_mi355x import weights as W
    from oracle import hdrtvnet_oracle as O
    phys, model = physical_cores()
    cores = max(1, min(16, phys))             # the GPU box's CPU share for one GPU
    torch.set_num_threads(cores)
    O.set_threads(cores)
    hr = W.load_pack(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"))
    hg = W.seeded_hg_state(1234) if use_hg else None
    plan = {"full": [(540, 960, 5, 20), (1080, 1920, 5, 20), (args.height, args.width, 0, 2)],
            "bounded": [(540, 960, 5, 20), (1080, 1920, 1, 5), (args.height, args.width, 1, 2)],
            "quick": [(540, 960, 1, 3), (1080, 1920, 0, 1)]}[args.cpu_protocol]
    seen = set()
    plan = [p for p in plan if not (p[:2] in seen or seen.add(p[:2]))]         # --height/--width equal to a fixed size: once

    def stages(frame):
        t0 = time.perf_counter()
        t, c = O.preprocess(frame)
        t1 = time.perf_counter()
        out = O.hg_composite(hr, hg, t, c)[0] if hg is not None else O.hr_forward(hr, t, c)[0]
        t2 = time.perf_counter()
        O.postprocess_u8(out)
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    runs = []
    O.use_backend("aten")
    try:
        stages(W.synthetic_frame(64, 96, seed=1, kind="noise"))        # page in, spin up threads
        for h, w, nwarm, ntimed in plan:
            frames = [W.synthetic_frame(h, w, seed=1234 + i, kind="noise" if i % 2 == 0 else "gradient") for i in range(2)]
            for i in range(nwarm):
                stages(frames[i % 2])
            acc = np.zeros(3)
            for i in range(ntimed):
                acc += np.array(stages(frames[i % 2]))
            acc /= ntimed
            runs.append({"size": f"{w}x{h}", "warmup": nwarm, "timed_frames": ntimed, "pre_ms": round(acc[0] * 1e3, 2),
                         "run_ms": round(acc[1] * 1e3, 1), "post_ms": round(acc[2] * 1e3, 2), "frames_per_s": round(1.0 / acc.sum(), 5)})
    finally:
        O.use_backend("c")
    last = runs[-1]
    lw, lh = (int(v) for v in last["size"].split("x"))
    scale = (args.height * args.width) / float(lh * lw)
    out = {"value": round(last["frames_per_s"] / scale, 5), "unit": "frames/s", "cores": cores, "cpu_model": model,
           "physical_cores_available": phys, "kind": "port", "backend": "PyTorch CPU eager (ATen / oneDNN), fp32, inference mode",
           "protocol": args.cpu_protocol,
           "sample": f"{last['timed_frames']} timed frame(s) at {last['size']} after {last['warmup']} warm-up, preprocess+AGCM+LE"
                     f"{'+HG' if use_hg else ''}+postprocess" + ("" if scale == 1.0 else f", scaled by pixel count (1/{scale:.1f} of {args.width}x{args.height})"),
           "runs": runs}
    # extra: the plain-C operators of the oracle on one frame of the bounded sample size
    w, h = (int(v) for v in args.cpu_sample.lower().split("x"))
    frame = W.synthetic_frame(h, w, seed=1234, kind="noise")
    t0 = time.perf_counter()
    O.process(hr, frame, hg)
    dt = time.perf_counter() - t0
    sc = (args.height * args.width) / float(h * w)
    out["c_port"] = {"value": round(1.0 / (dt * sc), 5), "unit": "frames/s", "cores": cores,
                     "sample": f"1 frame {w}x{h} through oracle/hdrtv_oracle.c in {dt:.2f} s, scaled by pixel count"}
    return out


# profile tag -> the source file whose change invalidates a committed counter figure for that kernel (besides common.h / launchers.h)
KERNEL_SOURCE = (("conv_prw8_i8", "conv3x3_prw_i8.hip"), ("conv_prw_i8", "conv3x3_prw_i8.hip"), ("conv_prw", "conv3x3_prw.hip"), ("conv_pglds_i8", "conv3x3_pglds_i8.hip"), ("conv_pglds", "conv3x3_pglds.hip"), ("conv_glds1", "conv1x1_glds.hip"),
                 ("le_rb_rows<i8>", "le_rows_i8.hip"), ("le_tail_rows<i8>", "le_rows_i8.hip"), ("le_head_rows<i8>", "le_rows_i8.hip"),
                 ("le_rb_rows", "le_rows.hip"), ("le_tail_rows", "le_rows.hip"), ("le_head_rows", "le_rows.hip"), ("conv32s", "conv32s.hip"),
                 ("conv32p", "conv32p.hip"), ("conv3x3s2_preg", "conv3x3s2_preg.hip"), ("conv1x1_i8", "conv_i8_misc.hip"), ("le_cond_trunk", "le_fused.hip"),
                 ("conv_c3", "le_hg_misc.hip"), ("hg_final", "le_hg_misc.hip"), ("agcm_mlp", "agcm.hip"), ("conv_q8", "conv_q8.hip"))


def traffic_for(kern, measured, doc, doc_path, build_id=None, source_hash=None):
    """(traffic, note): a committed PMC figure is reported only for the code it was measured on -- the library's build id equals
    the file's stamp, or at least the kernel's own source file and the shared headers hash as they did when it was measured
    (tools/pmc_to_json.py records both); otherwise (None, why)."""
    from hdrtv_mi355x import lib as L
    build_id = build_id or L.build_id
    source_hash = source_hash or L.source_hash
    stamp, srcs = doc.get("build_id"), doc.get("sources") or {}
    if not stamp:
        return None, f"{doc_path} carries no build stamp (measured before round 5): not reported"
    try:
        if stamp == build_id():
            return measured, None
        src = next((f for tag, f in KERNEL_SOURCE if kern.startswith(tag)), None)
        files = [src, "common.h", "launchers.h"] if src else []
        if src and all(srcs.get(f) == source_hash(f) for f in files):
            return measured, f"library rebuilt since {doc_path} was measured (build {stamp}); {src} and the shared headers are unchanged"
        return None, f"{doc_path} was measured on library build {stamp}, this is {build_id()} and {src or 'the kernel source'} differs: re-run tools/r05_final.sh"
    except OSError as exc:
        return None, f"cannot verify {doc_path} against the sources: {exc}"


def int8_extra(args, dev, dev_frames, steps=20, warmup=3, recipe="full"):
    """BASELINE configs[4] beside the headline, same frames, same timing method (never `value`): HR from the INT8-QAT
    checkpoint with predequantize off, HG head W8A8; the label states what ran from the launch profile.  `python bench.py --int8` is the full run."""
    import contextlib
    import torch
    from hdrtv_mi355x import lib as L
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    try:
        H, Wd = args.height, args.width
        with contextlib.redirect_stdout(sys.stderr):
            proc = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", f"hr_int8_{recipe}_qat.hdrw"), device=str(dev),
                                  precision=f"int8-{recipe}", predequantize="off", use_hg=True, hg_weights="seeded-w8a8:1234", warmup_passes=0,
                                  lanes=2)
        proc._ensure_buffers(H, Wd)
        lib, ctx = proc._lib, proc._ctx
        nl = 1            # `value`: one frame at a time, as the headline; `value_two_lanes` beside it
        rgb48 = [torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev) for _ in range(2)]

        def step(i, lanes=nl):
            proc.enqueue_frame(i % lanes, dev_frames[i % len(dev_frames)].data_ptr(), H, Wd, rgb48[i % lanes].data_ptr())

        def timed(lanes):
            for i in range(warmup):
                step(i, lanes)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(steps):
                step(i, lanes)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        el = timed(nl)
        el2 = timed(2)
        # what ran, from the kernel tags of one profiled frame behind the timed region (never a constant string)
        proc.profile_enable(True)
        step(0, 1)
        torch.cuda.synchronize(dev)
        ran = proc.execution_summary()
        proc.profile_enable(False)
        proc.close()
        return {"metric": f"frames/sec, INT8-QAT HDRTVNet++ (HR: the shipped {recipe}-QAT checkpoint, predequantize off; HG: W8A8 stand-in), same frames; "
                          f"executed: {ran['text']}",
                "value": round(steps / el, 3), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
                "lanes": nl, "value_two_lanes": round(steps / el2, 3), "dtype": "i8+f16", "executed": {k: v for k, v in ran.items() if k != "text"}}
    except Exception as exc:  # noqa: BLE001  (an extra: never take the headline line down with it)
        return {"error": f"{type(exc).__name__}: {exc}"}


def fp32_extra(args, dev, steps=10, warmup=2, H=1080, Wd=1920):
    """The reference's fp32 preset (precision="fp32", csrc/fp32_ops.hip) beside the headline, at 1920x1080 (never `value`):
    pre + infer + post_rgb48 on fp32 tensors, frames resident in HBM."""
    import contextlib
    import torch
    from hdrtv_mi355x import lib as L
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X
    try:
        with contextlib.redirect_stdout(sys.stderr):
            proc = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"), device=str(dev), precision="fp32",
                                  use_hg=True, hg_weights="seeded:1234", warmup_passes=0)
        proc._ensure_buffers(H, Wd)
        lib, ctx = proc._lib, proc._ctx
        frames = [torch.from_numpy(W.synthetic_frame(H, Wd, seed=40 + i, kind="gradient" if i else "noise")).to(dev) for i in range(2)]
        rgb48 = torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev)

        def step(i):
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            proc._chk(lib.hdrtv_preprocess(ctx, st, frames[i % 2].data_ptr(), H, Wd, proc._gpu_input.data_ptr(), proc._gpu_cond.data_ptr()), "preprocess")
            proc._chk(lib.hdrtv_infer(ctx, st, proc._gpu_input.data_ptr(), proc._gpu_cond.data_ptr(), H, Wd,
                                      proc._gpu_out.data_ptr(), L.F32, proc._gpu_agcm.data_ptr()), "infer")
            proc._chk(lib.hdrtv_post_rgb48(ctx, st, proc._gpu_out.data_ptr(), L.F32, H, Wd, rgb48.data_ptr()), "post_rgb48")

        for i in range(warmup):
            step(i)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        launches, macs = C.c_int(), C.c_double()
        lib.hdrtv_infer_stats(ctx, C.byref(launches), C.byref(macs))
        proc.profile_enable(True)
        step(0)
        torch.cuda.synchronize(dev)
        prof = proc.profile_read()
        proc.profile_enable(False)
        proc.close()
        mm = sum(m for _, k, _, m, _ in prof if k == "conv_f32_mfma") / max(1.0, sum(m for _, k, _, m, _ in prof if k.startswith("conv_f32")))
        return {"metric": f"frames/sec, HDRTVNet++ precision=fp32 (planar fp32 tensors; {100 * mm:.0f} % of the conv MACs on v_mfma_f32_32x32x2_f32, the rest on "
                          f"vector-FMA kernels) {Wd}x{H}, frames resident in HBM",
                "value": round(steps / el, 3), "unit": "frames/s", "ms_per_step": round(el / steps * 1e3, 3), "steps": steps,
                "dtype": "f32", "launches_per_frame": launches.value,
                "tflops_end_to_end": round(2 * macs.value * steps / el / 1e12, 1), "fp32_peak_tflops": 157.3}
    except Exception as exc:  # noqa: BLE001  (an extra: never take the headline line down with it)
        return {"error": f"{type(exc).__name__}: {exc}"}


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as fresh child processes -- this
    process has not imported torch or touched a GPU, and it never replaces itself with another program -- wait for them and
    relay rank 0's JSON line.  Returns the launcher's exit code."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)          # stderr passes through
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
    for ln in child.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return child.returncode if child.returncode else (0 if lines else 1)


def rank_stub():
    """HDRTV_BENCH_RANK_STUB=1 (tests/test_bench_contract.py, CPU): the rank-side protocol without a GPU -- rendezvous over
    gloo, barrier, max-over-ranks of a wall time, rank 0 prints one JSON line carrying n_gpus = WORLD_SIZE."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"stub": True, "n_gpus": world, "max_over_ranks": float(t.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def dispatcher_host_fed(args, frames, device_index, use_hg, steps, warmup=5, n_workers=1):
    """The in-product form of the host-fed path, N = n_workers GPUs (`python bench.py --dispatcher --gpus N`) (hdrtv_mi355x/dispatch.py): this process only fills shared-memory slots
    (a 24.9 MB memcpy per 4K frame) and consumes RGB48 views in order; ONE worker process owns the GPU context, DMAs straight
    from / into the page-locked slots and keeps two frames in flight.  Comparable with value_pcie_inclusive; never `value`."""
    from hdrtv_mi355x.dispatch import FrameDispatcher
    H, Wd = args.height, args.width
    seen = {"n": 0, "sum": 0}

    def sink(i, view):
        seen["n"] += 1
        seen["sum"] += int(view[H // 2, Wd // 2, 1])            # touch the frame: the view is only valid during the call

    init = {"model_path": os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"), "use_hg": use_hg,
            "hg_weights": "seeded:1234" if use_hg else None, "frames_in_flight": args.dispatcher_depth, "lanes": args.dispatcher_lanes}
    try:
        # (bounded waits: an extra must never push a default run past the driver's limit)
        devices = [device_index] if n_workers == 1 else list(range(n_workers))
        if os.environ.get("HDRTV_BENCH_ONE_DEVICE"):
            devices = [0] * n_workers                    # rehearsal on a box with fewer GPUs than workers
        with FrameDispatcher(n_workers, H, Wd, sink, init_args=init, devices=devices, slots=args.dispatcher_slots, start_timeout=150.0) as d:
            # one producer thread per worker copies the frames in (submit_async); with one worker the caller's thread does
            put = d.submit if n_workers == 1 else d.submit_async
            if os.environ.get("HDRTV_BENCH_ZERO_COPY"):          # experiment: a decoder that writes in place (reserve / commit), slots filled once
                filled = set()
                def put(frame):
                    i, view = d.reserve()
                    if id(view) not in filled and len(filled) < 64:
                        np.copyto(view, frame)
                        filled.add(id(view))
                    d.commit()
            for i in range(warmup * n_workers):
                put(frames[i % len(frames)])
            d.flush(timeout=100)
            steps *= n_workers
            before = list(d.frames_per_worker)
            d.worker_stats(reset=True)
            t0 = time.perf_counter()
            for i in range(steps):
                put(frames[i % len(frames)])
            d.flush(timeout=100)
            el = time.perf_counter() - t0
            wstats = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if k != "since"} if st else None for st in d.worker_stats()]
            placement = d.placement
            # a worker's own rate: its frames over the time to ITS last frame (a straggling GPU finishes late and shows here)
            per_worker = [round((d.frames_per_worker[r] - before[r]) / max(d.last_done[r] - t0, 1e-9), 3) for r in range(n_workers)]
        return {"value": round(steps / el, 3), "unit": "frames/s", "frames": steps, "workers": n_workers, "slots": args.dispatcher_slots, "frames_in_flight": args.dispatcher_depth,
                "lanes": args.dispatcher_lanes,
                "ms_per_frame": round(el / steps * 1e3, 3), "per_worker_frames_per_s": per_worker, "worker_loop_s": wstats, "worker_exit_codes": d.exit_codes,
                "placement": [{k: (p[k] if k != "cpus" else len(p[k])) for k in ("device", "numa_node", "cpus", "pinned")} for p in placement],
                "what": "FrameDispatcher: parent memcpy into a pinned shared slot -> worker hipMemcpyAsync H2D -> pre + infer + post_rgb48 "
                        "-> hipMemcpyAsync D2H into a pinned shared slot -> hipEvent -> in-order sink"}
    except Exception as exc:  # noqa: BLE001  (an extra: never take the headline line down with it)
        return {"error": f"{type(exc).__name__}: {exc}"}


def dispatcher_host_sim(args, n_workers, target_fps, seconds=6.0, use_numa=False, device_ms=9.5):
    """The HOST side of the N-GPU dispatcher without the GPUs (`python bench.py --dispatcher-sim --gpus 8`; CPU only): the real
    FrameDispatcher -- producer thread per worker, worker-owned shared-memory slots, reorder thread, in-order sink -- over N
    stand-in workers (dispatch.host_sim_worker: reads the whole input slot, holds the "device" for device_ms per frame with two
    frames in flight, writes the whole RGB48 slot).  A pacing loop offers frames at target_fps; reported: the rate delivered in
    order to the sink, per worker, and the CPU seconds per delivered frame of every parent thread."""
    from hdrtv_mi355x.dispatch import FrameDispatcher, host_sim_worker
    from hdrtv_mi355x import weights as W
    H, Wd = args.height, args.width
    frames = [W.synthetic_frame(H, Wd, seed=1234 + i, kind="noise") for i in range(2)]
    order = []

    def sink(i, view):
        order.append(i)
        _ = int(view[H // 2, Wd // 2, 1])

    n_frames = int(target_fps * seconds)
    t_main0 = time.thread_time()
    with FrameDispatcher(n_workers, H, Wd, sink, make_worker=host_sim_worker, init_args={"device_ms": device_ms}, slots=3,
                         start_timeout=120.0, numa=(True if use_numa else None)) as d:
        for i in range(2 * n_workers):
            d.submit_async(frames[i % 2])
        d.flush(timeout=120)
        n_warm = len(order)
        before = list(d.frames_per_worker)
        t0 = time.perf_counter()
        for i in range(n_frames):
            due = t0 + i / target_fps
            dt = due - time.perf_counter()
            if dt > 0:
                time.sleep(dt)
            d.submit_async(frames[i % 2])
        t_offered = time.perf_counter() - t0
        d.flush(timeout=300)
        el = time.perf_counter() - t0
        per_worker = [round((d.frames_per_worker[r] - before[r]) / max(d.last_done[r] - t0, 1e-9), 2) for r in range(n_workers)]
        cpu = {"producer_ms_per_frame": [round(1e3 * c * n_workers / max(n_frames, 1), 3) for c in d.host_cpu_s["producers"]],
               "reorder_ms_per_frame": round(1e3 * d.host_cpu_s["reorder"] / max(len(order), 1), 4),
               "submit_thread_ms_per_frame": round(1e3 * (time.thread_time() - t_main0) / max(n_frames, 1), 4)}
        depth = d.max_reorder_depth
    in_order = order == list(range(len(order)))
    return {"target_frames_per_s": target_fps, "delivered_frames_per_s": round(n_frames / el, 2), "offered_over_s": round(t_offered, 2),
            "frames": n_frames, "workers": n_workers, "device_ms": device_ms, "size": f"{Wd}x{H}", "in_order": in_order and len(order) == n_warm + n_frames,
            "per_worker_frames_per_s": per_worker, "max_reorder_depth": depth, "parent_cpu": cpu, "worker_exit_codes": d.exit_codes,
            # host memory traffic per frame: source read 3 + slot write 3 + slot read 3 + RGB48 slot write 6 bytes per pixel
            "host_bytes_per_frame": H * Wd * 15, "cores": len(os.sched_getaffinity(0)), "numa_pinning": bool(use_numa)}


def main():
    args = parse()
    if args.dispatcher_sim:
        # CPU only: where the host side of an N-GPU node saturates
        runs = [dispatcher_host_sim(args, max(1, args.gpus), fps, use_numa=args.sim_numa) for fps in (args.sim_fps or [100.0 * max(1, args.gpus)])]
        print(json.dumps({"dispatcher_host_sim": runs}), flush=True)
        return None
    if args.dispatcher:
        # one parent, N worker processes (no torch.distributed): the in-product multi-GPU form
        from hdrtv_mi355x import weights as W
        frames = [W.synthetic_frame(args.height, args.width, seed=1234 + i, kind="noise" if i % 2 == 0 else "gradient") for i in range(4)]
        print(json.dumps({"dispatcher_host_fed": dispatcher_host_fed(args, frames, 0, not args.no_hg, max(40, args.steps), n_workers=max(1, args.gpus))}), flush=True)
        return None
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    if os.environ.get("HDRTV_BENCH_RANK_STUB"):
        return rank_stub()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # placement before anything touches the GPU: this rank's threads (and the pinned buffers it allocates below) on its GPU's
    # NUMA node (hdrtv_mi355x/numa.py reads sysfs, makes no HIP call); N = 1 leaves the affinity alone (the CPU baseline
    # wants the box's 16-core share)
    from hdrtv_mi355x import numa
    place = numa.pin_to_gpu_node(0 if os.environ.get("HDRTV_BENCH_ONE_DEVICE") else local_rank, apply=world > 1)
    if world > 1:
        print(f"[bench] rank {rank}: {numa.describe(place)}", file=sys.stderr, flush=True)
    import torch
    import torch.distributed as dist
    # rehearsal switches for a box with fewer GPUs than ranks (never set by the driver): HDRTV_BENCH_BACKEND=gloo and
    # HDRTV_BENCH_ONE_DEVICE=1 put every rank on cuda:0, so the N > 1 code path can be exercised on one GPU
    backend = os.environ.get("HDRTV_BENCH_BACKEND", "nccl")
    if os.environ.get("HDRTV_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from hdrtv_mi355x import lib as L
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.processor import HDRTVNetMI355X

    use_hg = not args.no_hg
    H, Wd = args.height, args.width
    want_two = not args.no_two_lanes and world == 1 and args.lanes == 1
    ctx_lanes = 2 if want_two else args.lanes           # the context holds the second workspace from the start; `value` uses args.lanes
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):       # stdout carries exactly one JSON line
        proc = HDRTVNetMI355X(os.path.join(REPO, "tests", "golden", f"hr_int8_{args.int8_recipe}_qat.hdrw" if args.int8 else "hr_weights.hdrw"),
                              device=f"cuda:{local_rank}", precision=f"int8-{args.int8_recipe}" if args.int8 else "auto",
                              predequantize="auto" if (args.int8_predequantize or not args.int8) else "off",
                              use_hg=use_hg, hg_weights=("seeded-w8a8:1234" if args.int8 else "seeded:1234") if use_hg else None,
                              warmup_passes=0, lanes=ctx_lanes)
    proc._ensure_buffers(H, Wd)
    lib, ctx = proc._lib, proc._ctx

    # synthetic frames, resident in HBM: BASELINE.md section 3 noise protocol + gradient/highlight pattern
    nfr = 4
    frames = [W.synthetic_frame(H, Wd, seed=1234 + rank * nfr + i, kind="noise" if i % 2 == 0 else "gradient")
              for i in range(nfr)]
    dev_frames = [torch.from_numpy(f).to(dev) for f in frames]
    rgb48 = torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev)
    out_dt = L.F32 if use_hg else L.F16

    def stream():
        return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def step(i, dst_ptr=None):
        fr = dev_frames[i % nfr]
        proc._chk(lib.hdrtv_preprocess(ctx, stream(), fr.data_ptr(), H, Wd, proc._gpu_input.data_ptr(),
                                       proc._gpu_cond.data_ptr()), "preprocess")
        proc._chk(lib.hdrtv_infer(ctx, stream(), proc._gpu_input.data_ptr(), proc._gpu_cond.data_ptr(), H, Wd,
                                  proc._gpu_out.data_ptr(), out_dt, proc._gpu_agcm.data_ptr()), "infer")
        proc._chk(lib.hdrtv_post_rgb48(ctx, stream(), proc._gpu_out.data_ptr(), out_dt, H, Wd,
                                       dst_ptr if dst_ptr is not None else rgb48.data_ptr()), "post_rgb48")

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- the path north_star names: frames resident in HBM -> pre + infer + RGB48 post -> the pinned host ring
    # (hipMemcpyAsync on a copy stream + hipEvent per slot; the consumer waits and releases one frame behind).
    # Frame i runs on lane i mod LANES (its own activation workspace and HIP stream, processor.enqueue_frame): with two lanes the
    # device starts frame i + 1's kernels in the tails of frame i's.  The ring keeps the order: slots are committed and consumed
    # in frame order whatever order the lanes finish in.
    proc._chk(lib.hdrtv_ring_create(ctx, 2 + ctx_lanes, H, Wd), "ring_create")
    dn_stream = torch.cuda.Stream(dev)
    pending = []

    def ring_step(i, lanes=args.lanes, ev=None):
        hp, dp = C.c_void_p(), C.c_void_p()
        slot = proc._chk(lib.hdrtv_ring_acquire(ctx, 250, C.byref(hp), C.byref(dp)), "ring_acquire")
        lane = i % lanes
        ls = proc.lane_stream(lane)
        if ev is not None:
            ev[0].record(ls)
        proc.enqueue_frame(lane, dev_frames[i % nfr].data_ptr(), H, Wd, dp.value, stream=ls)   # RGB48 into the slot's device buffer
        if ev is not None:
            ev[1].record(ls)
        done = torch.cuda.Event()
        done.record(ls)
        dn_stream.wait_event(done)
        proc._chk(lib.hdrtv_ring_commit(ctx, slot, C.c_void_p(dn_stream.cuda_stream)), "ring_commit")
        pending.append(slot)
        if len(pending) == 1 + lanes:                           # consumer side: wait + release `lanes` frames behind
            s0 = pending.pop(0)
            lib.hdrtv_ring_wait(ctx, s0)
            lib.hdrtv_ring_release(ctx, s0)

    def ring_drain():
        while pending:
            s0 = pending.pop(0)
            lib.hdrtv_ring_wait(ctx, s0)
            lib.hdrtv_ring_release(ctx, s0)

    for i in range(args.warmup):
        ring_step(i)
    ring_drain()
    torch.cuda.synchronize(dev)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ring_step(i, ev=evs[i])
    ring_drain()
    torch.cuda.synchronize(dev)
    elapsed_own = time.perf_counter() - t0          # this rank's own K frames (per_rank_frames_per_s)
    barrier()
    elapsed = time.perf_counter() - t0
    per_frame_ms = sorted(a.elapsed_time(b) for a, b in evs)
    p50 = per_frame_ms[len(per_frame_ms) // 2]
    p99 = per_frame_ms[min(len(per_frame_ms) - 1, int(len(per_frame_ms) * 0.99))]
    # 1 % low as main.py:599-604: mean of the lowest 1 % of the per-frame fps samples (at least one sample)
    # frame TIMES (main.py:586-604 samples the time from one presented frame to the next): the intervals between consecutive frames'
    # last kernels -- with one lane that is the frame's own duration, with two it is what a consumer of the ring sees
    def frame_intervals(pairs):
        return [max(pairs[j - 1][1].elapsed_time(pairs[j][1]), 1e-6) for j in range(1, len(pairs))]
    fps_samples = sorted(1000.0 / ms for ms in frame_intervals(evs))
    one_pct_low = float(np.mean(fps_samples[:max(1, len(fps_samples) // 100)]))
    per_rank_fps, per_rank_node = [round(args.steps / elapsed_own, 3)], [place["numa_node"]]
    if world > 1:
        red_dev = dev if backend == "nccl" else "cpu"
        mine = torch.tensor([args.steps / elapsed_own, float(place["numa_node"])], device=red_dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_fps, per_rank_node = [round(float(t[0].item()), 3) for t in every], [int(t[1].item()) for t in every]
        tt = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # per-frame statistics: the slowest rank's p50 / p99 and the lowest rank's 1 % low
        pp = torch.tensor([p50, p99, -one_pct_low], device=red_dev, dtype=torch.float64)
        dist.all_reduce(pp, op=dist.ReduceOp.MAX)
        p50, p99, one_pct_low = float(pp[0].item()), float(pp[1].item()), -float(pp[2].item())
    value = world * args.steps / elapsed
    # ---- latency tail over more frames than the timed region holds (never `value`): with the driver's K = 20, p99 is the slowest
    # of 20 frames and the 1 % low a single sample; the same ring-inclusive step for >= 300 more frames gives them a population
    tail_stats = None
    if args.steps < 300 and not args.no_latency_tail:
        n_tail = 300
        tev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_tail)]
        for i in range(n_tail):
            ring_step(i, ev=tev[i])
        ring_drain()
        torch.cuda.synchronize(dev)
        tms = sorted(a.elapsed_time(b) for a, b in tev)
        tfps = sorted(1000.0 / ms for ms in frame_intervals(tev))
        tail_stats = {"frames": n_tail, "rank": rank, "p50_ms": round(tms[n_tail // 2], 3), "p99_ms": round(tms[int(n_tail * 0.99)], 3),
                      "max_ms": round(tms[-1], 3), "one_percent_low_fps": round(float(np.mean(tfps[:max(1, n_tail // 100)])), 3),
                      "what": "the timed region's step repeated for 300 more frames behind it (per-frame HIP events): the population p99 / 1 % low need; "
                              "p50 / p99 / max are a frame's first kernel to its last, the 1 % low is over the intervals between consecutive frames' completions"}
    # ---- the same K ring steps with TWO frames in flight (frame i on lane i mod 2: own workspace and stream; the device starts a
    # frame's kernels in the tails of the other's), reported beside `value` with its per-frame latency, and a self-check: 60 frames on
    # two lanes against the bytes the same frames give one at a time.  Never `value` (lanes are opt-in: DESIGN.md 7)
    two_lanes = None
    if want_two:
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for i in range(4):
            ring_step(i, lanes=2)
        ring_drain()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for i in range(args.steps):
            ring_step(i, lanes=2, ev=ev2[i])
        ring_drain()
        torch.cuda.synchronize(dev)
        el2 = time.perf_counter() - t1
        ms2 = sorted(a.elapsed_time(b) for a, b in ev2)
        want_bytes = []
        chk = [torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev) for _ in range(2)]
        for k in range(nfr):
            proc.enqueue_frame(0, dev_frames[k].data_ptr(), H, Wd, chk[0].data_ptr())
            torch.cuda.synchronize(dev)
            want_bytes.append(chk[0].clone())
        n_chk, bad_chk, worst = 60, 0, 0
        for i in range(0, n_chk, 2):
            for l in range(2):
                proc.enqueue_frame(l, dev_frames[(i + l) % nfr].data_ptr(), H, Wd, chk[l].data_ptr())
            torch.cuda.synchronize(dev)
            for l in range(2):
                nd = int((chk[l] != want_bytes[(i + l) % nfr]).sum())
                bad_chk += nd > 0
                worst = max(worst, nd)
        del chk, want_bytes
        two_lanes = {"value": round(args.steps / el2, 3), "unit": "frames/s", "p50_ms": round(ms2[len(ms2) // 2], 3),
                     "p99_ms": round(ms2[min(len(ms2) - 1, int(len(ms2) * 0.99))], 3), "rank": rank,
                     "selfcheck": {"frames": n_chk, "frames_differing_from_one_lane": bad_chk, "most_values_differing": worst}}
    lib.hdrtv_ring_destroy(ctx)

    # ---- the same K steps with the RGB48 frame left in device memory (no ring): reported at N = 1, never `value`
    device_only = None
    if world == 1:
        torch.cuda.synchronize(dev)
        u16 = [torch.empty((H, Wd, 3), dtype=torch.uint16, device=dev) for _ in range(args.lanes)]
        t1 = time.perf_counter()
        for i in range(args.steps):
            proc.enqueue_frame(i % args.lanes, dev_frames[i % nfr].data_ptr(), H, Wd, u16[i % args.lanes].data_ptr())
        torch.cuda.synchronize(dev)
        device_only = args.steps / (time.perf_counter() - t1)
        del u16

    # ---- PCIe-inclusive variant (pinned H2D in, RGB48 out through the pinned host ring): reported, never `value`
    pcie = None
    if rank == 0 and world == 1:
        proc._chk(lib.hdrtv_ring_create(ctx, 2 + args.lanes, H, Wd), "ring_create")
        pin = [torch.from_numpy(f).pin_memory() for f in frames]
        torch.cuda.synchronize(dev)
        n2, nwarm = max(20, args.steps), 3
        pending = []
        # uploads run one frame ahead on their own stream and RGB48 leaves through the ring on a copy stream
        # (hipMemcpyAsync + hipEvent handoffs, as playback.PinnedPrefetch and the worker's feeder do)
        up_stream, dn_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        up_ev = [torch.cuda.Event() for _ in range(nfr)]
        done_ev = [None] * nfr

        def upload(i):
            with torch.cuda.stream(up_stream):
                if done_ev[i % nfr] is not None:
                    up_stream.wait_event(done_ev[i % nfr])          # the slot's previous frame has been unpacked
                dev_frames[i % nfr].copy_(pin[i % nfr], non_blocking=True)
                up_ev[i % nfr].record(up_stream)

        upload(0)
        t1 = None
        for i in range(n2 + nwarm):
            if i == nwarm:
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
            hp, dp = C.c_void_p(), C.c_void_p()
            slot = proc._chk(lib.hdrtv_ring_acquire(ctx, 250, C.byref(hp), C.byref(dp)), "ring_acquire")
            if i + 1 < n2 + nwarm:
                upload(i + 1)
            main = proc.lane_stream(i % args.lanes)
            main.wait_event(up_ev[i % nfr])
            proc.enqueue_frame(i % args.lanes, dev_frames[i % nfr].data_ptr(), H, Wd, dp.value, stream=main)   # RGB48 into the slot's device buffer
            done_ev[i % nfr] = torch.cuda.Event()
            done_ev[i % nfr].record(main)
            dn_stream.wait_event(done_ev[i % nfr])
            proc._chk(lib.hdrtv_ring_commit(ctx, slot, C.c_void_p(dn_stream.cuda_stream)), "ring_commit")
            pending.append(slot)
            if len(pending) == 1 + args.lanes:         # consumer side: wait + release `lanes` frames behind
                s0 = pending.pop(0)
                lib.hdrtv_ring_wait(ctx, s0)
                lib.hdrtv_ring_release(ctx, s0)
        for s0 in pending:
            lib.hdrtv_ring_wait(ctx, s0)
            lib.hdrtv_ring_release(ctx, s0)
        torch.cuda.synchronize(dev)
        pcie = n2 / (time.perf_counter() - t1)
        lib.hdrtv_ring_destroy(ctx)

    # ---- roofline of the dominant kernel: HIP events around every launch of hdrtv_infer
    roof, roof_next, layers = None, None, None
    if rank == 0:
        proc.profile_enable(True)
        agg = {}
        nprof = max(3, min(args.steps, 10))
        for i in range(nprof):
            step(i)
            torch.cuda.synchronize(dev)
            layers = proc.profile_read()
            for layer, kern, ms, macs, nbytes in layers:
                a = agg.setdefault(kern, [0.0, 0.0, 0.0, 0])
                a[0] += ms; a[1] += macs; a[2] += nbytes; a[3] += 1
        proc.profile_enable(False)
        infer_ms = sum(v[0] for v in agg.values()) / nprof
        # HBM bytes per launch from the PMC counters: collected by rocprofv3 in separate --pmc passes of this same
        # command (tools/r04_final.sh, tools/pmc_to_json.py) and committed
        pmc_file = os.path.join("profiles", "pmc_traffic_int8.json" if args.int8 else "pmc_traffic.json")
        try:
            pmc_doc = json.load(open(os.path.join(REPO, pmc_file)))
            pmc = pmc_doc["kernels"]
        except (OSError, KeyError, ValueError):
            pmc_doc, pmc = {}, {}
        # profile tag -> kernel name in the rocprofv3 counter CSV (template argument = store mode, common.h)
        pmc_key = {"conv_prw<nhwc>": "conv_prw_kernel<0, 16>", "conv_prw<ps>": "conv_prw_kernel<1, 16>", "conv_prw<pool>": "conv_prw_kernel<2, 16>",
                   "conv_prw<ps_dot3>": "conv_prw_kernel<4, 16>", "conv_prw8<nhwc>": "conv_prw_kernel<0, 8>", "conv_prw8<ps>": "conv_prw_kernel<1, 8>",
                   "conv_prw8<pool>": "conv_prw_kernel<2, 8>",
                   "conv_prw_i8<nhwc>": "conv_prw_i8_kernel<0, 16>", "conv_prw_i8<ps>": "conv_prw_i8_kernel<1, 16>",
                   "conv_prw_i8<pool>": "conv_prw_i8_kernel<2, 16>", "conv_prw8_i8<nhwc>": "conv_prw_i8_kernel<0, 8>",
                   "conv_prw8_i8<ps>": "conv_prw_i8_kernel<1, 8>", "conv_prw8_i8<pool>": "conv_prw_i8_kernel<2, 8>",
                   "conv_pglds<nhwc>": "conv_pglds_kernel<0>", "conv_pglds<ps>": "conv_pglds_kernel<1>",
                   "conv_pglds<pool>": "conv_pglds_kernel<2>", "conv_pglds<ps_dot3>": "conv_pglds_kernel<4>",
                   "conv_glds1": "conv_glds1p_kernel",
                   "le_rb_rows": "le_rb_rows_kernel<3, false>", "le_tail_rows": "le_tail_rows_kernel<3, false>",
                   "le_head_rows": "le_head_rows_kernel<3, false>", "le_rb_rows<fq>": "le_rb_rows_kernel<3, true>",
                   "le_tail_rows<fq>": "le_tail_rows_kernel<3, true>", "le_head_rows<fq>": "le_head_rows_kernel<3, true>",
                   "le_rb_rows<i8>": "le_rb_rows_i8_kernel<3>", "le_tail_rows<i8>": "le_tail_rows_i8_kernel<3>", "le_head_rows<i8>": "le_head_rows_i8_kernel<3>",
                   "conv_pglds_i8<nhwc>": "conv_pglds_i8_kernel<0, false>", "conv_pglds_i8<ps>": "conv_pglds_i8_kernel<1, false>",
                   "conv_pglds_i8<pool>": "conv_pglds_i8_kernel<2, false>", "conv_pglds_i8<nhwc,c64>": "conv_pglds_i8_kernel<0, true>",
                   "conv_pglds_i8<ps_dot3,c64>": "conv_pglds_i8_kernel<4, true>", "conv1x1_i8": "conv1x1_i8_kernel<false>",
                   "conv1x1_i8<f16>": "conv1x1_i8_kernel<true>", "conv32s<1,sft>": "conv32s_kernel<true, false, false, false, false, true>",
                   "conv32s<1,c3+sft>": "conv32s_kernel<true, false, false, false, true, true>",
                   "conv32s<1,sft-i8,i8>": "conv32s_kernel<true, true, true, false, false, false>",
                   "conv32s<1,sft,i8>": "conv32s_kernel<true, true, false, false, false, true>",
                   "conv32p<4,plain>": "conv32p_kernel<4, false, 8, false, false>", "conv32s<1,plain>": "conv32s_kernel<false, false, false, true, false, false>",
                   "conv3x3s2_preg<192>": "conv3x3s2_preg_kernel<12>", "conv3x3s2_preg<192>+tail": "conv3x3s2_preg_kernel<12>", "conv3x3s2_preg<64>": "conv3x3s2_preg_kernel<4>", "conv3x3s2_preg<64>+tail": "conv3x3s2_preg_kernel<4>"}

        def roof_of(kern, ms, macs, nbytes, n):
            """achieved = ALGORITHMIC flops (or bytes) per launch / average launch time, against the roof that bounds the kernel:
            its algorithmic intensity vs the machine balance (peak FLOP/s / 8 TB/s)."""
            avg_ms = ms / n
            tflops = 2.0 * macs / n / (avg_ms * 1e-3) / 1e12
            traffic, why = None, None
            key = pmc_key.get(kern, kern)
            if (H, Wd) == (2160, 3840) and use_hg:
                hit = [v for k, v in pmc.items() if k == key or k.endswith(key) or (len(key) > 12 and key in k)]
                if hit:
                    traffic, why = traffic_for(kern, hit[0]["hbm_bytes_per_launch"], pmc_doc, pmc_file)
            peak = MFMA_F16_DENSE_PEAK_TFLOPS * (2.0 if "_i8" in kern else 1.0)       # int8 MFMA: twice the K per instruction
            if 2.0 * macs / max(nbytes, 1.0) >= peak * 1e12 / (HBM_PEAK_GBS * 1e9):
                r = {"kernel": kern, "bound": "mfma", "achieved": round(tflops, 2), "peak": peak,
                     "unit": "TFLOP/s", "frac": round(tflops / peak, 4), "traffic": traffic}
            else:
                gbs = nbytes / n / (avg_ms * 1e-3) / 1e9
                r = {"kernel": kern, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic}
            if why:
                r["traffic_note"] = why
            r.update({"algorithmic_bytes_per_launch": round(nbytes / n), "launches_per_frame": n // nprof,
                      "avg_launch_ms": round(avg_ms, 4), "share_of_infer_time": round(ms / nprof / infer_ms, 3)})
            return r

        ranked = sorted(agg.items(), key=lambda kv: -kv[1][0])
        kern, (ms, macs, nbytes, n) = ranked[0]
        roof = roof_of(kern, ms, macs, nbytes, n)
        roof.update({
                "traffic_source": pmc_file + f" (rocprofv3 --pmc passes of this command on library build {pmc_doc.get('build_id', '?')}, corrected as MI355X_MICROARCH.md prescribes; not re-measured in this run)" if roof["traffic"] is not None else None,
                "flop_per_launch": 2.0 * macs / n, "infer_ms_profiled": round(infer_ms, 3)})
        # the same figures for the next kernels by time (never `roofline`: the dominant kernel is the one above)
        roof_next = [roof_of(k, *v) for k, v in ranked[1:7] if v[1] > 0 or v[2] > 0]
        if args.layers:
            for kname, (kms, kmacs, kb, kn) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                print(f"[kernel] {kname:28s} n/frame={kn // nprof:3d} ms/frame={kms / nprof:8.3f} "
                      f"TFLOP/s={2 * kmacs / max(kms, 1e-9) / 1e9:8.1f} GB/s={kb / max(kms, 1e-9) / 1e6:8.1f}", file=sys.stderr)
            for layer, kname, lms, lmacs, lb in layers:
                print(f"[layer] {layer:28s} {kname:26s} {lms:8.3f} ms  {2 * lmacs / max(lms, 1e-9) / 1e9:8.1f} TFLOP/s "
                      f"{lb / max(lms, 1e-9) / 1e6:8.1f} GB/s", file=sys.stderr)

    launches, macs_frame = proc.infer_stats()
    ran = proc.execution_summary(layers) if (rank == 0 and args.int8) else None
    if rank == 0:
        line = {
            "metric": f"frames/sec (HDRTVNet++ AGCM+LE{'+HG' if use_hg else ''} {'INT8-QAT (executed: ' + ran['text'] + ')' if args.int8 else 'fp16'} "
                      f"{args.width}x{args.height} + fused RGB48 post into the pinned host ring); p50 per-frame ms in p50_ms",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "p50_ms": round(p50, 3), "p99_ms": round(p99, 3),
            "one_percent_low_fps": round(one_pct_low, 3), "latency_tail": tail_stats,
            "lanes": args.lanes, "two_lanes": two_lanes,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i8+f16" if args.int8 else "f16",
            "world_size": dist.get_world_size() if world > 1 else 1, "backend": (backend if backend != "nccl" else "nccl (RCCL)") if world > 1 else None,
            "per_rank_frames_per_s": per_rank_fps, "per_rank_numa_node": per_rank_node,
            "data": "synthetic (seeded u8 noise + gradient/highlight frames; HR.pt weights, seeded HG weights)",
            "config": {"workload": ((f"configs[4]: INT8-QAT HDRTVNet++ {Wd}x{H}: HR = the reference's HR_original_int8_{args.int8_recipe}_qat checkpoint, "
                                     + ("int8 weights dequantised to fp16 (the reference's ROCm behaviour)" if args.int8_predequantize else
                                        ("all 128 layers W8A8 (predequantize off)" if args.int8_recipe == "full" else
                                         "29 W8A8 + 78 W8A16 + 21 fp16 layers (predequantize off)"))
                                     + "; HG head: W8A8 stand-in (18 quantised layers); executed: " + ran["text"])
                                    if args.int8 else
                                    f"configs[2]: full HDRTVNet++ fp16 {Wd}x{H} + fused RGB48 post, 1 frame per GPU per step")
                       if use_hg else f"DEBUG no-HG {Wd}x{H}",
                       "frames_per_step": world, "sharding": "frame i -> GPU i mod N, no collective",
                       "frames_in_flight_per_gpu": args.lanes,
                       "launches_per_frame": launches, "gmac_per_frame": round(macs_frame / 1e9, 1)},
            "tflops_end_to_end": round(2 * macs_frame * value / world / 1e12, 1),
            "value_device_only": round(device_only, 3) if device_only else None,
            "value_pcie_inclusive": round(pcie, 3) if pcie else None,
            "value_is": f"u8 frames resident in HBM -> pre_fused + infer + post_rgb48 -> pinned host RGB48 ring (hipMemcpyAsync + hipEvent), {args.lanes} frame(s) in "
                        "flight per GPU (p50_ms / p99_ms: one frame's first kernel to its last; `two_lanes`: the same K steps with frame i on lane i mod 2 "
                        "-- own workspace and stream --, an opt-in mode that is never `value`); "
                        "value_device_only leaves the RGB48 frame in HBM; value_pcie_inclusive also uploads each frame from pinned host memory",
            "roofline": roof,
            "roofline_next": roof_next,
        }
        if world == 1 and use_hg and not args.int8 and not args.no_int8_extra:
            line["config4_int8"] = int8_extra(args, dev, dev_frames)
            # the reference's default preset is the mixed recipe (gui_config.py: DEFAULT_PRECISION_KEY = "INT8 Mixed (QAT)")
            line["config4_int8_mixed"] = int8_extra(args, dev, dev_frames, recipe="mixed")
            line["preset_fp32_1080p"] = fp32_extra(args, dev)
        if world == 1 and not args.int8 and not args.no_dispatcher:
            line["dispatcher_host_fed"] = dispatcher_host_fed(args, frames, local_rank, use_hg, max(40, args.steps))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, use_hg)
        print(json.dumps(line), flush=True)
    proc.close()
    if world > 1:
        dist.barrier()                     # rank 0 is still profiling while the others are done: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
