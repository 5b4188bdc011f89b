"""FrameDispatcher -- one stream of frames over the N GPUs of a node (BASELINE.json configs[3]).

The reference drives a single device (``hdrtvnet_torch.py:1678-1690``); frames are independent, so N devices are fed
round-robin: frame *i* goes to worker *i mod N*, every worker is its own process (spawned before anything touches a
GPU) with a full weight replica, and there is no data-path collective.  The host restores presentation order by
frame index before the sink sees a frame -- the ``preserve_order`` behaviour of the reference's feeder queue
(``gui_pipeline_worker_feeders.py:639-643``: buffered playback presents in source order).

Data path per frame and GPU (3840x2160: 24.9 MB in, 49.8 MB out):

    parent: frame -> input slot (shared memory the worker has page-locked with hipHostRegister); a producer that can
            write in place uses ``reserve()`` / ``commit()`` and skips the memcpy
    worker: hipMemcpyAsync H2D straight from the slot on an upload stream -> hipEvent -> pre + infer + post_rgb48 on the
            compute stream -> hipEvent -> hipMemcpyAsync D2H straight into an output slot on a copy stream -> hipEvent
            (the hand-off of ``feeders.py:440-496``, with HIP events where the reference uses CUDA events)
    parent: reorder stage hands slot views to the sink in index order, then returns the slot to its worker

A worker keeps ``depth`` (2) frames in flight: frame *i + 1* is uploaded and queued behind frame *i* on the device before
the worker waits for frame *i*'s download event, so upload, compute and download overlap and the device never idles
between frames.  Slots are the hand-off unit in both directions (``slots`` per worker, default 3 like the reference's
``HDRTVNET_FEEDER_GPU_RGB48_RING_FRAMES``): ``submit`` blocks while worker *i mod N* has no free input slot, a worker
holds back while all its output slots are still with the sink.

NUMA (a two-socket node has four GPUs per socket): a worker first restricts itself to the cores of its GPU's NUMA node
(``numa.pin_to_gpu_node``, before anything touches the GPU), then CREATES and first-touches its own slot segment -- its pages
land on that node -- and only then builds its processor and page-locks the slots; the parent attaches to the segment by
name.  ``placement`` reports what every worker did.  On the parent, ``submit`` copies in the caller's thread;
``submit_async`` hands the frame to a producer thread per worker, so that the input copies of N workers run on N threads
(one Python thread copies ~10 GB/s: four 4K workers' worth).

The worker body comes from ``make_worker(rank, device_index, init_args)``.  It is either a plain function
``process(frame_u8[H,W,3], out_u16[H,W,3])`` (synchronous: tests substitute CPU stand-ins of this form) or an object with
``begin(frame, out) -> token`` / ``finish(token)`` (+ optional ``depth``, ``pin(buffer)``, ``close()``): the product's
``mi355x_worker`` below, one ``HDRTVNetMI355X`` context per process.
"""
from __future__ import annotations

import collections
import contextlib
import gc
import multiprocessing as mp
import os
import queue as _queue
import sys
import threading
import time
import traceback
import uuid
from multiprocessing import shared_memory

import numpy as np


class _Mi355xWorker:
    """One processor on ``cuda:<device_index>``; two frames in flight: an upload stream, a download stream and one compute
    stream (and activation workspace: a lane) per frame in flight."""

    depth = 2

    def __init__(self, rank, device_index, init_args):
        import ctypes as C

        import torch

        from . import lib as L
        from .processor import HDRTVNetMI355X
        self._C, self._torch, self._L = C, torch, L
        kw = dict(init_args)
        model = kw.pop("model_path")
        # frames in flight (uploads, compute, downloads together) and, of those, frames computing at once (lanes)
        self.depth = max(1, int(kw.pop("frames_in_flight", self.depth)))
        # compute lanes: 1 by default -- host-fed, the second lane's device-side gain (bench.py `one_lane` vs `value`) has not shown
        # through the upload / download hand-offs in this loop; init_args["lanes"] = 2 turns it on
        kw.setdefault("lanes", 1)
        torch.cuda.set_device(device_index)
        with contextlib.redirect_stdout(sys.stderr):   # a worker's banner must not land on the parent's stdout (bench.py's one JSON line)
            self.proc = HDRTVNetMI355X(model, device=f"cuda:{device_index}", warmup_passes=0, **kw)
        self.dev = self.proc.device
        hip = C.CDLL("libamdhip64.so")          # torch's copy: already loaded under this SONAME (lib.load imports torch first)
        hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
        hip.hipHostUnregister.argtypes = [C.c_void_p]
        hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        for f in (hip.hipHostRegister, hip.hipHostUnregister, hip.hipMemcpyAsync):
            f.restype = C.c_int
        self._hip = hip
        self._registered = None
        self._up, self._dn = torch.cuda.Stream(self.dev), torch.cuda.Stream(self.dev)
        self._hw = None
        self._n = 0

    def pin(self, buf):
        """Page-lock the shared-memory slots so that both copies are DMA transfers straight from / into them."""
        a = np.frombuffer(buf, dtype=np.uint8)
        ptr, n = a.ctypes.data, a.nbytes
        del a
        rc = self._hip.hipHostRegister(ptr, n, 0)
        if rc != 0:
            raise RuntimeError(f"hipHostRegister({n} bytes of shared slots) failed with {rc}: the dispatcher's copies would be "
                               "pageable transfers")
        self._registered = ptr

    def _buffers(self, h, w):
        torch = self._torch
        if self._hw != (h, w):
            self.proc._ensure_buffers(h, w)
            self._raw = [torch.empty((h, w, 3), dtype=torch.uint8, device=self.dev) for _ in range(self.depth)]
            self._u16 = [torch.empty((h, w, 3), dtype=torch.uint16, device=self.dev) for _ in range(self.depth)]
            self._up_ev = [torch.cuda.Event() for _ in range(self.depth)]
            self._comp_ev = [None] * self.depth
            self._dn_ev = [None] * self.depth
            self._hw = (h, w)

    def begin(self, frame, out):
        torch, p = self._torch, self.proc
        h, w = frame.shape[:2]
        self._buffers(h, w)
        k = self._n % self.depth
        # frame n runs on lane n mod lanes: its own workspace and compute stream, so that the device overlaps the tail of one
        # frame's kernels with the next frame's (processor.enqueue_frame); one lane = the single compute stream of before
        lane = self._n % p.lanes
        self._n += 1
        main = p.lane_stream(lane)
        if self._comp_ev[k] is not None:
            self._up.wait_event(self._comp_ev[k])              # the frame that last used raw[k] has been unpacked
        rc = self._hip.hipMemcpyAsync(self._raw[k].data_ptr(), frame.ctypes.data, frame.nbytes, 1, self._up.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"hipMemcpyAsync H2D failed with {rc}")
        self._up_ev[k].record(self._up)
        main.wait_event(self._up_ev[k])
        if self._dn_ev[k] is not None:
            main.wait_event(self._dn_ev[k])                    # u16[k]'s previous frame has left the device
        p.enqueue_frame(lane, self._raw[k].data_ptr(), h, w, self._u16[k].data_ptr(), stream=main)
        self._comp_ev[k] = torch.cuda.Event()
        self._comp_ev[k].record(main)
        self._dn.wait_event(self._comp_ev[k])
        rc = self._hip.hipMemcpyAsync(out.ctypes.data, self._u16[k].data_ptr(), out.nbytes, 2, self._dn.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"hipMemcpyAsync D2H failed with {rc}")
        self._dn_ev[k] = torch.cuda.Event()
        self._dn_ev[k].record(self._dn)
        return k

    def finish(self, token):
        self._dn_ev[token].synchronize()

    def close(self):
        self._torch.cuda.synchronize(self.dev)
        self.proc.close()
        if self._registered is not None:
            self._hip.hipHostUnregister(self._registered)
            self._registered = None


def mi355x_worker(rank, device_index, init_args):
    """The product's worker body (see the module docstring)."""
    return _Mi355xWorker(rank, device_index, init_args)


class _HostSimWorker:
    """CPU stand-in for one GPU worker, for measuring the HOST side of the dispatcher without a node of GPUs
    (``bench.py --dispatcher-sim``, tests): it does what the host sees of a real worker -- the whole input slot is read
    (the H2D copy's source traffic) and the whole output slot written (the D2H copy's destination traffic), with plain memcpys on
    the worker's core where a real worker's DMA engines would move them -- and a frame occupies the "device" for ``device_ms``
    (two frames in flight, as ``_Mi355xWorker``).  Pessimistic for the workers' CPU time, faithful for the parent's producers,
    the slot hand-off, the reorder stage and the memory traffic through the host."""

    depth = 2

    def __init__(self, rank, device_index, init_args):
        self.rank = rank
        self.device_ms = float(init_args.get("device_ms", 9.5))
        self._staged = None
        self._result = None
        self._busy_until = 0.0
        self.cpu_s = 0.0

    def begin(self, frame, out):
        t0 = time.thread_time()
        if self._staged is None or self._staged.shape != frame.shape:
            self._staged = np.empty_like(frame)
            self._result = np.empty(out.shape, out.dtype)
            self._result[...] = 257 * (self.rank + 1)
        np.copyto(self._staged, frame)                     # "H2D": the slot's 24.9 MB (4K) leave host memory
        self._result[0, 0, 0] = int(frame[0, 0, 0]) * 257  # (so that a test can tell the frames apart)
        self._result[0, 0, 1] = self.rank
        start = max(time.perf_counter(), self._busy_until)
        self._busy_until = start + self.device_ms * 1e-3   # the device works on one frame at a time
        self.cpu_s += time.thread_time() - t0
        return (out, self._busy_until, int(self._result[0, 0, 0]))

    def finish(self, token):
        out, t_done, tag = token
        dt = t_done - time.perf_counter()
        if dt > 0:
            time.sleep(dt)
        t0 = time.thread_time()
        np.copyto(out, self._result)                       # "D2H": 49.8 MB (4K) land in host memory
        out[0, 0, 0] = tag
        self.cpu_s += time.thread_time() - t0


def host_sim_worker(rank, device_index, init_args):
    return _HostSimWorker(rank, device_index, init_args)


class _SyncBody:
    """Adapter: a plain ``process(frame, out)`` function as a depth-1 begin / finish body."""

    depth = 1

    def __init__(self, fn):
        self._fn = fn
        for name in ("pin", "close"):
            if hasattr(fn, name):
                setattr(self, name, getattr(fn, name))

    def begin(self, frame, out):
        self._fn(frame, out)
        return None

    def finish(self, token):
        return None


def _worker_main(rank, device_index, make_worker, init_args, geom, task_q, done_q, use_numa, shm_name=None):
    shm, ins, outs, body = None, None, None, None
    code = 0
    try:
        h, w, slots = geom
        in_b, out_b = h * w * 3, h * w * 6
        # placement first: affinity, then the slots (created and first-touched HERE, on the GPU's node), then the GPU
        from . import numa
        info = numa.pin_to_gpu_node(device_index, apply=bool(use_numa)) if use_numa is not None else {"device": device_index, "numa_node": -1, "cpus": [], "pinned": False}
        # the NAME is the parent's (chosen before this process existed): whatever happens to this process, or to the message
        # below, the parent can unlink the segment
        shm = shared_memory.SharedMemory(name=shm_name, create=True, size=slots * (in_b + out_b))
        try:        # the parent unlinks the segment (it outlives this process's views); keep this process's tracker out of it
            from multiprocessing import resource_tracker
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:  # noqa: BLE001
            pass
        numa.first_touch(shm.buf)
        info = dict(info, pid=os.getpid(), slot_bytes=slots * (in_b + out_b))
        done_q.put(("shm", rank, shm.name, info))
        body = make_worker(rank, device_index, init_args)
        if not hasattr(body, "begin"):
            body = _SyncBody(body)
        if hasattr(body, "pin"):
            body.pin(shm.buf)
        ins = [np.ndarray((h, w, 3), np.uint8, shm.buf, offset=s * in_b) for s in range(slots)]
        outs = [np.ndarray((h, w, 3), np.uint16, shm.buf, offset=slots * in_b + s * out_b) for s in range(slots)]
        depth = max(1, int(getattr(body, "depth", 1)))
        free_out = list(range(slots))
        backlog = collections.deque()
        inflight = collections.deque()
        stopping = False
        # where this worker's wall time goes (FrameDispatcher.worker_stats): starved = blocked with nothing in flight and nothing
        # it could start (no frame offered, or every output slot still with the sink); finish = blocked on the oldest frame
        stats = {"frames": 0, "starved_s": 0.0, "begin_s": 0.0, "finish_s": 0.0, "since": time.perf_counter()}
        done_q.put(("ready", rank, None, None))
        while True:
            # messages: block only when there is nothing else to do
            block = not stopping and not inflight and not (backlog and free_out)
            while True:
                t_blk = time.perf_counter() if block else None
                try:
                    msg = task_q.get(block=block)
                except _queue.Empty:
                    break
                if t_blk is not None:
                    stats["starved_s"] += time.perf_counter() - t_blk
                block = False
                if msg[0] == "stop":
                    stopping = True
                elif msg[0] == "release":
                    free_out.append(msg[1])
                elif msg[0] == "stats":
                    now = time.perf_counter()
                    done_q.put(("stats", rank, None, dict(stats, wall_s=now - stats["since"], inflight=len(inflight), backlog=len(backlog))))
                    if msg[1:] and msg[1]:                 # ("stats", True): start a new interval
                        stats = {"frames": 0, "starved_s": 0.0, "begin_s": 0.0, "finish_s": 0.0, "since": now}
                else:
                    backlog.append(msg)
            if stopping:
                backlog.clear()
            # queue frames behind the ones in flight before waiting for any of them
            while backlog and free_out and len(inflight) < depth:
                _, idx, in_slot = backlog.popleft()
                out_slot = free_out.pop(0)
                t_b = time.perf_counter()
                inflight.append((body.begin(ins[in_slot], outs[out_slot]), idx, in_slot, out_slot))
                stats["begin_s"] += time.perf_counter() - t_b
            if inflight:
                token, idx, in_slot, out_slot = inflight.popleft()
                stats["depth_sum"] = stats.get("depth_sum", 0) + len(inflight) + 1      # frames in flight when the loop turns to wait
                t_b = time.perf_counter()
                body.finish(token)
                stats["finish_s"] += time.perf_counter() - t_b
                stats["frames"] += 1
                done_q.put(("frame", rank, idx, (in_slot, out_slot)))
            elif stopping:
                break
    except BaseException:  # noqa: BLE001  (reported to the parent, which raises it from submit / flush)
        done_q.put(("error", rank, None, traceback.format_exc()))
        code = 1
    finally:
        try:
            if body is not None and hasattr(body, "close"):
                body.close()
        except BaseException:  # noqa: BLE001
            code = code or 1
        # the numpy views export shm.buf: drop them before the mapping goes away (BufferError otherwise)
        ins = outs = body = None
        gc.collect()
        if shm is not None:
            try:
                shm.close()
            except BufferError:
                pass
    if code:
        raise SystemExit(code)


class FrameDispatcher:
    def __init__(self, n_workers, height, width, sink, make_worker=mi355x_worker, init_args=None, devices=None, slots=3,
                 start_timeout=600.0, numa=True):
        """``sink(index, rgb48_view)`` is called in index order from the reorder thread; the view is only valid during the
        call (the slot goes back to its worker afterwards).  ``numa``: workers pin themselves to their GPU's NUMA node
        (False: report only)."""
        if n_workers < 1 or slots < 2:
            raise ValueError("n_workers >= 1 and slots >= 2")
        self.n, self.h, self.w, self.slots = int(n_workers), int(height), int(width), int(slots)
        self._sink = sink
        ctx = mp.get_context("spawn")           # fresh interpreters: nothing GPU-related is inherited
        self._in_b, self._out_b = self.h * self.w * 3, self.h * self.w * 6
        self._shm = [None] * self.n             # created by the workers (first touch on their GPU's node), attached below
        # ... under names chosen HERE, so that close() can unlink a segment whose worker died between creating it and saying so
        self._shm_names = [f"hdrtv_{os.getpid()}_{uuid.uuid4().hex[:12]}_{r}" for r in range(self.n)]
        self.placement = [None] * self.n
        self._task = [ctx.Queue() for _ in range(self.n)]
        self._done = ctx.Queue()
        devices = list(devices) if devices is not None else list(range(self.n))
        self._stop = False
        self._procs = [ctx.Process(target=_worker_main, daemon=True,
                                   args=(r, devices[r], make_worker, dict(init_args or {}),
                                         (self.h, self.w, self.slots), self._task[r], self._done, bool(numa), self._shm_names[r]))
                       for r in range(self.n)]
        for p in self._procs:
            p.start()
        self._ins, self._outs = [None] * self.n, [None] * self.n
        self._producers, self._prod_q = None, None
        self._free_in = [_queue.Queue() for _ in range(self.n)]
        for r in range(self.n):
            for s in range(self.slots):
                self._free_in[r].put(s)
        self._next_submit = 0
        self._reserved = None
        self._next_emit = 0
        self._held = {}
        self._error = None
        self._emitted = threading.Condition()
        self.max_reorder_depth = 0
        self._stats = {}
        self.frames_per_worker = [0] * self.n                 # frames each worker has delivered (a straggler shows here)
        self.last_done = [0.0] * self.n                       # perf_counter() of each worker's last delivered frame
        self.host_cpu_s = {"producers": [0.0] * self.n, "reorder": 0.0}      # CPU seconds of the parent's own threads
        ready, t_end = 0, time.monotonic() + start_timeout
        while ready < self.n:
            try:
                kind, rank, name, payload = self._done.get(timeout=0.5)
            except _queue.Empty:
                dead = self._dead_worker()
                if dead or time.monotonic() > t_end:
                    self.close()
                    raise RuntimeError(dead or "dispatcher workers did not come up") from None
                continue
            if kind == "error":
                self.close()
                raise RuntimeError(f"dispatcher worker {rank} failed to start:\n{payload}")
            if kind == "shm":                   # the worker's slot segment: attach, build the views
                self._shm[rank] = shared_memory.SharedMemory(name=name)
                self.placement[rank] = payload
                buf = self._shm[rank].buf
                self._ins[rank] = [np.ndarray((self.h, self.w, 3), np.uint8, buf, offset=s * self._in_b) for s in range(self.slots)]
                self._outs[rank] = [np.ndarray((self.h, self.w, 3), np.uint16, buf, offset=self.slots * self._in_b + s * self._out_b)
                                    for s in range(self.slots)]
                continue
            ready += 1
        self._thread = threading.Thread(target=self._reorder, name="dispatch-reorder", daemon=True)
        self._thread.start()

    # ---------------------------------------------------------------- parent side
    def reserve(self):
        """Zero-copy submission, step 1: ``(index, view)`` of the input slot frame ``index`` will be read from -- a decoder
        writes the u8 BGR frame straight into ``view`` and then calls ``commit()``.  Blocks while worker ``index mod N`` has
        no free input slot.  One reservation at a time."""
        if self._reserved is not None:
            raise RuntimeError("reserve() called twice without commit()")
        self._raise_if_failed()
        i = self._next_submit
        r = i % self.n
        while True:
            try:
                s = self._free_in[r].get(timeout=0.25)
                break
            except _queue.Empty:
                self._raise_if_failed()
        self._reserved = (i, r, s)
        return i, self._ins[r][s]

    def commit(self):
        """Zero-copy submission, step 2: hand the reserved slot to its worker."""
        if self._reserved is None:
            raise RuntimeError("commit() without reserve()")
        i, r, s = self._reserved
        self._reserved = None
        self._task[r].put(("frame", i, s))
        self._next_submit = i + 1
        return i

    def submit(self, frame):
        """Frame ``i`` (the i-th call) -> worker ``i mod N``.  Blocks while that worker's input slots are all in flight."""
        _, view = self.reserve()
        try:
            np.copyto(view, frame)
        except BaseException:
            i, r, s = self._reserved
            self._reserved = None
            self._free_in[r].put(s)
            raise
        return self.commit()

    def submit_async(self, frame):
        """As ``submit``, but the copy into the slot runs on worker ``i mod N``'s producer thread: the call returns at once
        and ``frame`` must stay unchanged until the frame has been emitted (or ``flush`` has returned).  Do not mix with
        ``reserve`` / ``submit`` on one dispatcher."""
        if self._producers is None:
            self._prod_q = [_queue.Queue() for _ in range(self.n)]
            self._producers = [threading.Thread(target=self._produce, args=(r,), name=f"dispatch-producer-{r}", daemon=True) for r in range(self.n)]
            for t in self._producers:
                t.start()
        self._raise_if_failed()
        i = self._next_submit
        self._next_submit = i + 1
        self._prod_q[i % self.n].put((i, frame))
        return i

    def _produce(self, r):
        while not self._stop:
            self.host_cpu_s["producers"][r] = time.thread_time()
            try:
                i, frame = self._prod_q[r].get(timeout=0.1)
            except _queue.Empty:
                continue
            while not self._stop and self._error is None:
                try:
                    s = self._free_in[r].get(timeout=0.1)
                    break
                except _queue.Empty:
                    continue
            else:
                return
            np.copyto(self._ins[r][s], frame)
            self._task[r].put(("frame", i, s))

    def flush(self, timeout=600.0):
        """Wait until every submitted frame has been handed to the sink."""
        t_end = time.monotonic() + timeout
        with self._emitted:
            while self._next_emit < self._next_submit:
                self._raise_if_failed()
                if not self._emitted.wait(timeout=0.25) and time.monotonic() > t_end:
                    raise TimeoutError("dispatcher flush timed out")
        self._raise_if_failed()

    def worker_stats(self, reset=False, timeout=5.0):
        """Per worker, since start (or the last ``reset``): frames delivered and the seconds its loop spent ``starved`` (nothing in
        flight and nothing startable: no frame offered or no free output slot), in ``begin`` (enqueueing a frame) and blocked in
        ``finish`` (waiting for its oldest frame), of ``wall_s``.  A GPU-bound worker is mostly in ``finish``; a starved one says
        the host side is the limit."""
        with self._emitted:
            self._stats = {}
        for q in self._task:
            q.put(("stats", bool(reset)))
        t_end = time.monotonic() + timeout
        with self._emitted:
            while len(self._stats) < self.n and time.monotonic() < t_end:
                self._emitted.wait(timeout=0.1)
            return [self._stats.get(r) for r in range(self.n)]

    def _dead_worker(self):
        """A worker that is gone without having been asked to stop (GPU fault, abort, OOM kill, SIGSEGV: it posts nothing)."""
        if self._stop:
            return None
        for r, p in enumerate(getattr(self, "_procs", [])):
            if not p.is_alive() and p.exitcode is not None:
                return f"dispatcher worker {r} died with exit code {p.exitcode}"
        return None

    def _raise_if_failed(self):
        if self._error is None:
            dead = self._dead_worker()
            if dead:
                self._error = dead
        if self._error is not None:
            raise RuntimeError(f"dispatcher worker failed:\n{self._error}")

    def _reorder(self):
        while not self._stop:
            try:
                kind, rank, idx, payload = self._done.get(timeout=0.1)
            except _queue.Empty:
                if self._error is None:
                    dead = self._dead_worker()
                    if dead:
                        self._error = dead
                        with self._emitted:
                            self._emitted.notify_all()
                continue
            if kind == "error":
                self._error = payload
                with self._emitted:
                    self._emitted.notify_all()
                continue
            if kind == "stats":
                with self._emitted:
                    self._stats[rank] = payload
                    self._emitted.notify_all()
                continue
            if kind != "frame":
                continue
            self.frames_per_worker[rank] += 1
            self.last_done[rank] = time.perf_counter()
            self.host_cpu_s["reorder"] = time.thread_time()
            self._held[idx] = (rank, payload)
            self.max_reorder_depth = max(self.max_reorder_depth, len(self._held))
            while self._next_emit in self._held:            # release strictly in source order
                r, (in_slot, out_slot) = self._held.pop(self._next_emit)
                try:
                    self._sink(self._next_emit, self._outs[r][out_slot])
                except BaseException as exc:  # noqa: BLE001
                    self._error = f"sink raised: {exc!r}"
                self._task[r].put(("release", out_slot))
                self._free_in[r].put(in_slot)
                with self._emitted:
                    self._next_emit += 1
                    self._emitted.notify_all()

    def close(self):
        """Stops the workers (frames still queued are dropped) and releases the shared slots.  ``exit_codes`` afterwards
        holds every worker's exit code: 0 for a clean stop."""
        self._stop = True
        for q in getattr(self, "_task", []):
            try:
                q.put(("stop",))
            except Exception:  # noqa: BLE001
                pass
        for p in getattr(self, "_procs", []):
            p.join(timeout=20.0)
            if p.is_alive():
                p.terminate()
                p.join(timeout=5.0)
        self.exit_codes = [p.exitcode for p in getattr(self, "_procs", [])]
        t = getattr(self, "_thread", None)
        if t is not None:
            t.join(timeout=2.0)
        # the producer threads copy into the slots: they see _stop within 0.1 s; joined BEFORE the views go away (a copy in
        # flight would otherwise hold a buffer export, and SharedMemory.close() raises BufferError under it)
        for t in getattr(self, "_producers", None) or []:
            t.join(timeout=5.0)
        self._ins = self._outs = None
        gc.collect()
        shms = list(getattr(self, "_shm", []))
        for r, name in enumerate(getattr(self, "_shm_names", [])):
            s = shms[r] if r < len(shms) else None
            if s is None:
                # never attached (start-up failed, or the worker died before / while reporting): the segment may exist all the same
                try:
                    s = shared_memory.SharedMemory(name=name)
                except (FileNotFoundError, OSError, ValueError):
                    continue
            try:
                s.close()
            except Exception:  # noqa: BLE001
                pass
            try:
                s.unlink()
            except Exception:  # noqa: BLE001
                pass
        self._shm = []
        self._shm_names = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
