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


def _startup_failing_worker(rank, device_index, init_args):
    """make_worker itself fails on rank 1 -- AFTER _worker_main has created (and announced) its slot segment; rank 0 is slow to
    come up, so the parent is still collecting messages when the error arrives."""
    import time
    if rank == 1:
        raise RuntimeError("no such device")
    time.sleep(0.5)
    return lambda frame, out: None


def _dies_before_reporting(rank, device_index, init_args):
    os._exit(5)


def test_dispatcher_startup_failure_leaves_no_segment_behind(monkeypatch):
    """A failed start-up must not leak /dev/shm segments (224 MB per worker at 4K): the parent chose the names, so close()
    unlinks every segment -- attached or not, announced or not."""
    from multiprocessing import shared_memory
    import pytest
    from hdrtv_mi355x import dispatch
    for worker, msg in ((_startup_failing_worker, "no such device"), (_dies_before_reporting, "died with exit code 5")):
        names = []
        orig = dispatch.uuid.uuid4

        def tagged():
            u = orig()
            names.append(u.hex[:12])
            return u

        monkeypatch.setattr(dispatch.uuid, "uuid4", tagged)
        with pytest.raises(RuntimeError, match=msg):
            dispatch.FrameDispatcher(2, 8, 8, lambda i, v: None, make_worker=worker, init_args={}, slots=2, start_timeout=60)
        monkeypatch.setattr(dispatch.uuid, "uuid4", orig)
        assert len(names) == 2
        for r, tag in enumerate(names):
            with pytest.raises(FileNotFoundError):
                shared_memory.SharedMemory(name=f"hdrtv_{os.getpid()}_{tag}_{r}")


def test_dispatcher_close_joins_the_producer_threads():
    """close() with submit_async copies still queued: the producer threads are joined before the slot views are dropped (no
    BufferError from SharedMemory.close, no exception in a producer thread), workers exit 0."""
    from hdrtv_mi355x.dispatch import FrameDispatcher
    import threading
    errors = []
    old = threading.excepthook
    threading.excepthook = lambda a: errors.append(a)
    try:
        d = FrameDispatcher(2, 64, 64, lambda i, v: None, make_worker=_standin_worker, init_args={"delay": [0.02, 0.02]}, slots=2)
        frames = [np.full((64, 64, 3), i, np.uint8) for i in range(40)]
        for f in frames:
            d.submit_async(f)
        d.close()
        assert all(not t.is_alive() for t in d._producers)
    finally:
        threading.excepthook = old
    assert not errors, errors
    assert d.exit_codes == [0, 0]


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


# ------------------------------------------------------------------------------------------ NUMA placement (hdrtv_mi355x/numa.py)
def _fake_sysfs(root, gpus, nodes):
    """A sysfs tree with a two-socket topology: ``gpus`` = [(location_id, domain, numa_node)], ``nodes`` = {node: cpulist}."""
    top = os.path.join(root, "class/kfd/kfd/topology/nodes")
    os.makedirs(os.path.join(top, "0"))
    open(os.path.join(top, "0", "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (loc, dom, node) in enumerate(gpus):
        os.makedirs(os.path.join(top, str(i + 1)))
        open(os.path.join(top, str(i + 1), "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {loc}\ndomain {dom}\n")
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
        os.makedirs(os.path.join(root, "bus/pci/devices", bdf))
        open(os.path.join(root, "bus/pci/devices", bdf, "numa_node"), "w").write(f"{node}\n")
    for n, cl in nodes.items():
        os.makedirs(os.path.join(root, "devices/system/node", f"node{n}"))
        open(os.path.join(root, "devices/system/node", f"node{n}", "cpulist"), "w").write(cl + "\n")


def test_numa_placement_from_sysfs(tmp_path):
    """Eight GPUs on two sockets, as the KFD topology lists them: device i -> its PCI address -> its NUMA node -> that node's
    cores (intersected with the process's affinity).  No HIP call: the worker runs this before it touches the GPU."""
    from hdrtv_mi355x import numa
    root = str(tmp_path)
    have = sorted(os.sched_getaffinity(0))
    half = max(1, len(have) // 2)
    lo, hi = have[:half], have[half:] or have[:half]
    gpus = [((0x05 + 0x10 * i) << 8, 0, 0 if i < 4 else 1) for i in range(8)]
    _fake_sysfs(root, gpus, {0: ",".join(map(str, lo)), 1: ",".join(map(str, hi))})
    assert numa.parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    assert numa.gpu_pci_addresses(root)[:2] == ["0000:05:00.0", "0000:15:00.0"] and len(numa.gpu_pci_addresses(root)) == 8
    assert [numa.gpu_numa_node(i, root, env={}) for i in range(8)] == [0, 0, 0, 0, 1, 1, 1, 1]
    assert numa.gpu_numa_node(1, root, env={"HIP_VISIBLE_DEVICES": "6,2"}) == 0 and numa.gpu_numa_node(0, root, env={"HIP_VISIBLE_DEVICES": "6,2"}) == 1
    assert numa.gpu_numa_node(9, root, env={}) == -1
    i0 = numa.pin_to_gpu_node(1, root, env={}, apply=False)
    i1 = numa.pin_to_gpu_node(5, root, env={}, apply=False)
    assert i0["numa_node"] == 0 and i0["cpus"] == lo and i1["numa_node"] == 1 and i1["cpus"] == hi
    # a machine without the topology (this container, a single-node box): nothing is changed, nothing fails
    none = numa.pin_to_gpu_node(0, os.path.join(root, "nowhere"), env={})
    assert none["numa_node"] == -1 and not none["pinned"] and none["cpus"] == have
    assert "numa_node=1" in numa.describe(i1)


def _placement_worker(rank, device_index, init_args):
    """Stand-in that reports where it runs: by the time the body is built the worker has pinned itself and created its slots."""
    aff = sorted(os.sched_getaffinity(0))

    def process(frame, out):
        out[...] = 0
        out[0, 0, 0] = len(aff)
        out[0, 0, 1] = os.getpid() % 60000

    return process


def test_dispatcher_workers_own_their_slots_and_producers_scale():
    """The slot segments are created (and first-touched) by the WORKER processes, after they have pinned themselves -- not by
    the parent -- and ``submit_async`` copies on one producer thread per worker; order and contents as with ``submit``."""
    from hdrtv_mi355x.dispatch import FrameDispatcher
    h, w, n = 16, 24, 31
    seen = {}
    with FrameDispatcher(3, h, w, lambda i, v: seen.__setitem__(i, v.copy()), make_worker=_placement_worker, init_args={}, slots=2) as d:
        assert len(d.placement) == 3 and all(p is not None for p in d.placement)
        pids = [p["pid"] for p in d.placement]
        assert os.getpid() not in pids and len(set(pids)) == 3                  # three worker processes created three segments
        assert all(p["slot_bytes"] == 2 * (h * w * 3 + h * w * 6) for p in d.placement)
        assert all(p["numa_node"] == -1 and not p["pinned"] for p in d.placement)        # this container has no GPU topology
        frames = [np.full((h, w, 3), i % 200, np.uint8) for i in range(n)]
        for f in frames:
            d.submit_async(f)
        d.flush(timeout=60)
        assert sorted(t.name for t in d._producers) == ["dispatch-producer-0", "dispatch-producer-1", "dispatch-producer-2"]
    assert sorted(seen) == list(range(n))
    for i in range(n):
        assert int(seen[i][0, 0, 1]) == pids[i % 3] % 60000                     # frame i ran in worker i mod N's process
    assert d.exit_codes == [0, 0, 0]


def test_dispatcher_host_side_keeps_up_with_four_simulated_gpus():
    """The host side of the multi-GPU dispatcher with the GPUs simulated (bench.py --dispatcher-sim: the real FrameDispatcher over
    stand-in workers that read the whole input slot, hold a "device" for 9.5 ms per frame with two frames in flight and write the
    whole RGB48 slot): four workers at 1920x1080 offered 4 x 100 frames/s must deliver >= 85 % of it, strictly in order, evenly
    over the workers (no straggler), every worker exiting 0.  (3840x2160 over 8 workers in the 8-core build container: 398 of 400
    offered, saturating at ~400 frames/s with all 16 threads memcpy-bound -- profiles/r05_dispatcher_sim_8cores.json)"""
    import argparse
    import sys
    sys.path.insert(0, REPO) if REPO not in sys.path else None
    import bench
    args = argparse.Namespace(height=1080, width=1920)
    r = bench.dispatcher_host_sim(args, 4, 400.0, seconds=3.0)
    print(r)
    assert r["in_order"] and r["worker_exit_codes"] == [0, 0, 0, 0]
    assert r["delivered_frames_per_s"] >= 0.85 * 400.0
    pw = r["per_worker_frames_per_s"]
    assert len(pw) == 4 and min(pw) >= 0.85 * 100.0 and max(pw) - min(pw) <= 0.1 * max(pw)
    assert r["parent_cpu"]["reorder_ms_per_frame"] < 1.0 and r["parent_cpu"]["submit_thread_ms_per_frame"] < 1.0
