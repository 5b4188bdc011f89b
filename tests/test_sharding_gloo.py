"""N > 1 path on CPU: two gloo ranks shard frames round-robin (frame i -> rank i mod N), process
their share with the ORACLE standing in for the per-GPU device work, and the host restores order
by frame index -- the same partitioning bench.py uses with one MI355X per rank (no data-path
collective; the only collectives are the barrier and the max-over-ranks of the wall time)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, n_frames, out_dir):
    for p in (REPO, os.path.join(REPO, "hdr-realtime-video-pipeline_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.worker import shard_frames
    from oracle import hdrtvnet_oracle as O
    O.set_threads(2)
    hr = W.load_pack(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"))
    mine = shard_frames(n_frames, rank, world)
    dist.barrier()
    for i in mine:
        frame = W.synthetic_frame(64, 96, seed=100 + i, kind="noise")
        np.save(os.path.join(out_dir, f"f{i}.npy"), O.process(hr, frame))
    t = torch.tensor([float(len(mine))], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)            # every frame processed exactly once
    assert int(t.item()) == n_frames
    tm = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX)            # bench.py's max-over-ranks timing
    assert tm.item() == float(world)
    dist.barrier()
    dist.destroy_process_group()


def test_round_robin_two_ranks(tmp_path):
    from hdrtv_mi355x import weights as W
    from hdrtv_mi355x.worker import shard_frames
    from oracle import hdrtvnet_oracle as O
    assert shard_frames(8, 0, 8) == [0] and shard_frames(5, 1, 2) == [1, 3] and shard_frames(3, 3, 4) == []
    assert sorted(sum((shard_frames(11, r, 4) for r in range(4)), [])) == list(range(11))
    n = 5
    mp.spawn(_rank_main, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    hr = W.load_pack(os.path.join(REPO, "tests", "golden", "hr_weights.hdrw"))
    for i in range(n):       # order restored by index; results identical to a single-rank run
        want = O.process(hr, W.synthetic_frame(64, 96, seed=100 + i, kind="noise"))
        assert np.array_equal(np.load(os.path.join(str(tmp_path), f"f{i}.npy")), want)


def _standin_worker(rank, device_index, init_args):
    """CPU stand-in for one GPU worker of the dispatcher: a deterministic u8 -> u16 map, slower on even ranks so that
    frames finish out of order."""
    import time

    def process(frame, out):
        time.sleep(init_args["delay"][rank % len(init_args["delay"])] * (1 + (int(frame[0, 0, 0]) % 3)))
        np.multiply(frame, 257, out=out, dtype=np.uint16, casting="unsafe")
        out[0, 0, 1] = rank

    return process


def _failing_worker(rank, device_index, init_args):
    def process(frame, out):
        if rank == 1:
            raise RuntimeError("device fell off the bus")
        out[...] = 0
    return process


def _dying_worker(rank, device_index, init_args):
    """A worker that is killed mid-stream without a Python exception (what a GPU fault, an abort or the OOM killer does)."""
    count = {"n": 0}

    def process(frame, out):
        count["n"] += 1
        if rank == 1 and count["n"] == 2:
            os._exit(3)
        out[...] = 1
    return process


class _PipelinedStandin:
    """begin / finish stand-in with two frames in flight: the work of a frame happens in finish(), so a correct worker
    loop must have begun frame i + 1 before it finishes frame i (recorded in the output for the test to see)."""
    depth = 2

    def __init__(self):
        self.begun = 0

    def begin(self, frame, out):
        self.begun += 1
        return (frame, out, self.begun)

    def finish(self, token):
        import time
        frame, out, seq = token
        time.sleep(0.01)                           # the "device" is slower than the producer: frames queue up behind it
        np.multiply(frame, 257, out=out, dtype=np.uint16, casting="unsafe")
        out[0, 0, 2] = self.begun - seq            # frames begun after this one at the time it is finished


def _pipelined_worker(rank, device_index, init_args):
    return _PipelinedStandin()


def test_dispatcher_round_robin_restores_order():
    """The product's multi-GPU dispatcher (hdrtv_mi355x/dispatch.py) with two stand-in workers: frame i is processed by
    worker i mod 2, completions arrive out of order, the sink sees indices 0, 1, 2, ... with the right contents."""
    from hdrtv_mi355x.dispatch import FrameDispatcher
    h, w, n = 24, 32, 23
    seen = []

    def sink(idx, view):
        seen.append((idx, int(view[0, 0, 1]), view.copy()))

    frames = [np.full((h, w, 3), (7 * i) % 251, np.uint8) for i in range(n)]
    with FrameDispatcher(2, h, w, sink, make_worker=_standin_worker, init_args={"delay": [0.03, 0.002]}, slots=3) as d:
        for f in frames:
            d.submit(f)
        d.flush(timeout=60)
        depth = d.max_reorder_depth
    assert d.exit_codes == [0, 0]                                       # a clean stop is exit code 0 (no BufferError at shm.close)
    assert [s[0] for s in seen] == list(range(n))                       # presentation order = source order
    assert [s[1] for s in seen] == [i % 2 for i in range(n)]            # frame i ran on worker i mod N
    for i, (_, _, got) in enumerate(seen):
        want = frames[i].astype(np.uint16) * 257
        want[0, 0, 1] = i % 2
        assert np.array_equal(got, want)
    assert depth >= 2                                                   # the reorder stage really held frames back


def test_dispatcher_reports_worker_failure():
    from hdrtv_mi355x.dispatch import FrameDispatcher
    import pytest
    d = FrameDispatcher(2, 8, 8, lambda i, v: None, make_worker=_failing_worker, init_args={}, slots=2)
    try:
        with pytest.raises(RuntimeError, match="fell off the bus"):
            for _ in range(8):
                d.submit(np.zeros((8, 8, 3), np.uint8))
            d.flush(timeout=30)
    finally:
        d.close()


def test_dispatcher_detects_a_dead_worker():
    """A worker process that disappears posts nothing: submit / flush must still fail (not spin), naming the exit code."""
    from hdrtv_mi355x.dispatch import FrameDispatcher
    import pytest
    import time
    d = FrameDispatcher(2, 8, 8, lambda i, v: None, make_worker=_dying_worker, init_args={}, slots=2)
    t0 = time.monotonic()
    try:
        with pytest.raises(RuntimeError, match="worker 1 died with exit code 3"):
            for _ in range(12):
                d.submit(np.zeros((8, 8, 3), np.uint8))
            d.flush(timeout=30)
        assert time.monotonic() - t0 < 20
    finally:
        d.close()
    assert d.exit_codes[1] == 3


def test_dispatcher_keeps_two_frames_in_flight_and_zero_copy_submit():
    """begin / finish bodies: the worker loop begins frame i + 1 before it waits for frame i whenever a frame is queued
    (the overlap the product's worker gets from its three streams), and reserve() / commit() feed slots in place."""
    from hdrtv_mi355x.dispatch import FrameDispatcher
    h, w, n = 8, 16, 20
    seen = {}
    with FrameDispatcher(1, h, w, lambda i, v: seen.__setitem__(i, v.copy()), make_worker=_pipelined_worker, init_args={}, slots=3) as d:
        for i in range(n):
            idx, view = d.reserve()
            assert idx == i
            view[...] = i + 1
            d.commit()
        d.flush(timeout=60)
    assert sorted(seen) == list(range(n))
    for i in range(n):
        assert seen[i][1, 1, 0] == (i + 1) * 257
    overlapped = sum(int(seen[i][0, 0, 2]) >= 1 for i in range(n))
    assert overlapped >= n - 4, overlapped          # the parent submits ahead: all but the first / last frames had a successor begun
