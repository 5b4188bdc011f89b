"""FrameDispatcher -- one stream of frames over the N GPUs of a node (BASELINE.json configs[3]).

The reference drives a single device (``hdrtvnet_torch.py:1678-1690``); frames are independent, so N devices are fed
round-robin: frame *i* goes to worker *i mod N*, every worker is its own process (spawned before anything touches a
GPU) with a full weight replica, and there is no data-path collective.  The host restores presentation order by
frame index before the sink sees a frame -- the ``preserve_order`` behaviour of the reference's feeder queue
(``gui_pipeline_worker_feeders.py:639-643``: buffered playback presents in source order).

Data path per frame and GPU (3840x2160: 24.9 MB in, 49.8 MB out):

    parent: memcpy frame -> input slot (shared memory, page-locked by the worker with hipHostRegister)
    worker: hipMemcpyAsync H2D -> pre + infer + post_rgb48 -> hipMemcpyAsync D2H into an output slot (same mapping) -> event
    parent: reorder stage hands slot views to the sink in index order, then returns the slot to its worker

Slots are the hand-off unit in both directions (``slots`` per worker, default 3 like the reference's
``HDRTVNET_FEEDER_GPU_RGB48_RING_FRAMES``): ``submit`` blocks while worker *i mod N* has no free input slot, a worker
blocks while all its output slots are still with the sink.  The worker body is a plain function
``make_worker(rank, device_index, init_args) -> process(frame_u8[H,W,3]) -> u16[H,W,3]``: the product's is
``mi355x_worker`` below (one ``HDRTVNetMI355X`` per process); tests substitute a CPU stand-in.
"""
from __future__ import annotations

import multiprocessing as mp
import queue as _queue
import threading
import time
import traceback
from multiprocessing import shared_memory

import numpy as np


def mi355x_worker(rank, device_index, init_args):
    """The product's worker body: one processor on ``cuda:<device_index>``, pinned shared slots, RGB48 out."""
    import ctypes as C

    import torch

    from . import lib as L
    from .processor import HDRTVNetMI355X
    kw = dict(init_args)
    model = kw.pop("model_path")
    torch.cuda.set_device(device_index)
    proc = HDRTVNetMI355X(model, device=f"cuda:{device_index}", warmup_passes=0, **kw)
    state = {"dev_u16": None}

    def pin(buf):          # page-lock the shared-memory slots so that both copies are DMA transfers
        rt = torch.cuda.cudart()
        a = np.frombuffer(buf, dtype=np.uint8)
        rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)

    def process(frame, out):
        h, w = frame.shape[:2]
        if state["dev_u16"] is None or tuple(state["dev_u16"].shape) != (h, w, 3):
            state["dev_u16"] = torch.empty((h, w, 3), dtype=torch.uint16, device=proc.device)
        t, c = proc.preprocess(frame)
        o, _ = proc.infer((t, c))
        st = C.c_void_p(torch.cuda.current_stream(proc.device).cuda_stream)
        proc._chk(proc._lib.hdrtv_post_rgb48(proc._ctx, st, o.data_ptr(), L.F32 if o.dtype == torch.float32 else L.F16, h, w,
                                             state["dev_u16"].data_ptr()), "hdrtv_post_rgb48")
        torch.from_numpy(out).copy_(state["dev_u16"], non_blocking=True)
        torch.cuda.current_stream(proc.device).synchronize()

    process.pin = pin
    process.close = proc.close
    return process


def _worker_main(rank, device_index, make_worker, init_args, shm_name, geom, task_q, done_q):
    shm = None
    try:
        h, w, slots = geom
        in_b, out_b = h * w * 3, h * w * 6
        shm = shared_memory.SharedMemory(name=shm_name)
        process = make_worker(rank, device_index, init_args)
        if hasattr(process, "pin"):
            process.pin(shm.buf)
        ins = [np.ndarray((h, w, 3), np.uint8, shm.buf, offset=s * in_b) for s in range(slots)]
        outs = [np.ndarray((h, w, 3), np.uint16, shm.buf, offset=slots * in_b + s * out_b) for s in range(slots)]
        free_out = list(range(slots))
        done_q.put(("ready", rank, None, None))
        backlog = []
        while True:
            msg = task_q.get() if not backlog or not free_out else (task_q.get_nowait() if not task_q.empty() else None)
            if msg is not None:
                if msg[0] == "stop":
                    break
                if msg[0] == "release":
                    free_out.append(msg[1])
                else:
                    backlog.append(msg)
            if backlog and free_out:
                _, idx, in_slot = backlog.pop(0)
                out_slot = free_out.pop(0)
                process(ins[in_slot], outs[out_slot])
                done_q.put(("frame", rank, idx, (in_slot, out_slot)))
        if hasattr(process, "close"):
            process.close()
    except BaseException:  # noqa: BLE001  (reported to the parent, which raises it from submit / flush)
        done_q.put(("error", rank, None, traceback.format_exc()))
    finally:
        if shm is not None:
            shm.close()


class FrameDispatcher:
    def __init__(self, n_workers, height, width, sink, make_worker=mi355x_worker, init_args=None, devices=None, slots=3,
                 start_timeout=600.0):
        """``sink(index, rgb48_view)`` is called in index order from the reorder thread; the view is only valid during the
        call (the slot goes back to its worker afterwards)."""
        if n_workers < 1 or slots < 2:
            raise ValueError("n_workers >= 1 and slots >= 2")
        self.n, self.h, self.w, self.slots = int(n_workers), int(height), int(width), int(slots)
        self._sink = sink
        ctx = mp.get_context("spawn")           # fresh interpreters: nothing GPU-related is inherited
        self._in_b, self._out_b = self.h * self.w * 3, self.h * self.w * 6
        self._shm = [shared_memory.SharedMemory(create=True, size=self.slots * (self._in_b + self._out_b)) for _ in range(self.n)]
        self._task = [ctx.Queue() for _ in range(self.n)]
        self._done = ctx.Queue()
        devices = list(devices) if devices is not None else list(range(self.n))
        self._procs = [ctx.Process(target=_worker_main, daemon=True,
                                   args=(r, devices[r], make_worker, dict(init_args or {}), self._shm[r].name,
                                         (self.h, self.w, self.slots), self._task[r], self._done)) for r in range(self.n)]
        for p in self._procs:
            p.start()
        self._ins = [[np.ndarray((self.h, self.w, 3), np.uint8, self._shm[r].buf, offset=s * self._in_b) for s in range(self.slots)]
                     for r in range(self.n)]
        self._outs = [[np.ndarray((self.h, self.w, 3), np.uint16, self._shm[r].buf, offset=self.slots * self._in_b + s * self._out_b)
                       for s in range(self.slots)] for r in range(self.n)]
        self._free_in = [_queue.Queue() for _ in range(self.n)]
        for r in range(self.n):
            for s in range(self.slots):
                self._free_in[r].put(s)
        self._next_submit = 0
        self._next_emit = 0
        self._held = {}
        self._error = None
        self._emitted = threading.Condition()
        self.max_reorder_depth = 0
        ready, t_end = 0, time.monotonic() + start_timeout
        while ready < self.n:
            try:
                kind, rank, _, payload = self._done.get(timeout=max(0.1, t_end - time.monotonic()))
            except _queue.Empty:
                self.close()
                raise RuntimeError("dispatcher workers did not come up") from None
            if kind == "error":
                self.close()
                raise RuntimeError(f"dispatcher worker {rank} failed to start:\n{payload}")
            ready += 1
        self._stop = False
        self._thread = threading.Thread(target=self._reorder, name="dispatch-reorder", daemon=True)
        self._thread.start()

    # ---------------------------------------------------------------- parent side
    def submit(self, frame):
        """Frame ``i`` (the i-th call) -> worker ``i mod N``.  Blocks while that worker's input slots are all in flight."""
        self._raise_if_failed()
        i = self._next_submit
        r = i % self.n
        while True:
            try:
                s = self._free_in[r].get(timeout=0.25)
                break
            except _queue.Empty:
                self._raise_if_failed()
        np.copyto(self._ins[r][s], frame)
        self._task[r].put(("frame", i, s))
        self._next_submit = i + 1
        return i

    def flush(self, timeout=600.0):
        """Wait until every submitted frame has been handed to the sink."""
        t_end = time.monotonic() + timeout
        with self._emitted:
            while self._next_emit < self._next_submit:
                self._raise_if_failed()
                if not self._emitted.wait(timeout=0.25) and time.monotonic() > t_end:
                    raise TimeoutError("dispatcher flush timed out")
        self._raise_if_failed()

    def _raise_if_failed(self):
        if self._error is not None:
            raise RuntimeError(f"dispatcher worker failed:\n{self._error}")

    def _reorder(self):
        while not self._stop:
            try:
                kind, rank, idx, payload = self._done.get(timeout=0.1)
            except _queue.Empty:
                continue
            if kind == "error":
                self._error = payload
                with self._emitted:
                    self._emitted.notify_all()
                continue
            if kind != "frame":
                continue
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
        self._stop = True
        for q in getattr(self, "_task", []):
            try:
                q.put(("stop",))
            except Exception:  # noqa: BLE001
                pass
        for p in getattr(self, "_procs", []):
            p.join(timeout=10.0)
            if p.is_alive():
                p.terminate()
        t = getattr(self, "_thread", None)
        if t is not None:
            t.join(timeout=2.0)
        self._ins = self._outs = None
        for s in getattr(self, "_shm", []):
            try:
                s.close()
                s.unlink()
            except Exception:  # noqa: BLE001
                pass
        self._shm = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
