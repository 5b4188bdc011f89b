"""NUMA placement for one-process-per-GPU runs (BASELINE.json configs[3]; no counterpart in the reference, which drives
``cuda:0`` only: ``hdrtvnet_torch.py:1678-1690``).

On a two-socket MI355X node four GPUs hang off each socket.  A worker whose threads run on the other socket, or whose
page-locked frame slots were first touched there, moves every frame (24.9 MB in, 49.8 MB out at 3840x2160) across the
socket link.  So, BEFORE anything touches the GPU in the process:

    info = numa.pin_to_gpu_node(device_index)     # sched_setaffinity to the cores of the GPU's NUMA node

and the process then allocates and first-touches its own slots (Linux places a page on the node of the CPU that first
writes it).  Everything is read from sysfs -- the KFD topology lists the GPUs in HIP's enumeration order with their PCI
address -- so no HIP call is made and nothing needs the GPU; ``sysfs`` can point at a fake tree (tests).
"""
from __future__ import annotations

import os
import re


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    """``"0-3,8,10-11"`` -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in (text or "").replace("\n", "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def visible_devices(env=None):
    """The physical indices HIP's device numbers stand for (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES,
    integer lists only), or None when every device is visible."""
    env = os.environ if env is None else env
    for key in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(key)
        if v:
            try:
                return [int(x) for x in v.split(",") if x.strip() != ""]
            except ValueError:
                return None
    return None


def gpu_pci_addresses(sysfs="/sys"):
    """PCI addresses (``dddd:bb:dd.f``) of the GPUs in the KFD topology's order -- the order HIP enumerates them in."""
    root = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((int(n) for n in os.listdir(root) if n.isdigit()))
    except OSError:
        return []
    out = []
    for n in nodes:
        props = {}
        for line in (_read(os.path.join(root, str(n), "properties")) or "").splitlines():
            kv = line.split()
            if len(kv) == 2:
                props[kv[0]] = kv[1]
        if int(props.get("simd_count", "0")) <= 0:
            continue                                  # a CPU node
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        out.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}")
    return out


def gpu_numa_node(device_index, sysfs="/sys", env=None):
    """NUMA node of HIP device ``device_index`` (-1: unknown / single-node machine)."""
    vis = visible_devices(env)
    phys = vis[device_index] if vis is not None and 0 <= device_index < len(vis) else device_index
    gpus = gpu_pci_addresses(sysfs)
    if not 0 <= phys < len(gpus):
        return -1
    v = _read(os.path.join(sysfs, "bus/pci/devices", gpus[phys], "numa_node"))
    try:
        return int(v)
    except (TypeError, ValueError):
        return -1


def node_cpus(node, sysfs="/sys"):
    return parse_cpulist(_read(os.path.join(sysfs, "devices/system/node", f"node{node}", "cpulist")))


def pin_to_gpu_node(device_index, sysfs="/sys", env=None, apply=True):
    """Restrict this process to the cores of the GPU's NUMA node (intersected with its current affinity).  Call before the
    first GPU call and before allocating the buffers the GPU will DMA from / into.  Returns ``{"device", "numa_node",
    "cpus", "pinned"}``; a machine that reports no node for the GPU (or whose node has none of our cores) is left alone."""
    node = gpu_numa_node(device_index, sysfs, env)
    info = {"device": int(device_index), "numa_node": node, "cpus": sorted(os.sched_getaffinity(0)), "pinned": False}
    if node < 0:
        return info
    want = node_cpus(node, sysfs) & os.sched_getaffinity(0)
    if not want:
        return info
    if apply:
        os.sched_setaffinity(0, want)
    info["cpus"], info["pinned"] = sorted(want), bool(apply)
    return info


def first_touch(buf, page=4096):
    """Write one byte per page of a freshly mapped buffer so that its pages are placed on this thread's NUMA node."""
    mv = memoryview(buf).cast("B")
    for off in range(0, len(mv), page):
        mv[off] = 0
    if len(mv):
        mv[len(mv) - 1] = 0
    mv.release()


def describe(info):
    cpus = info.get("cpus") or []
    rng = re.sub(r"\s+", "", _ranges(cpus))
    return f"cuda:{info.get('device')} numa_node={info.get('numa_node')} cpus={rng} pinned={info.get('pinned')}"


def _ranges(cpus):
    out, start, prev = [], None, None
    for c in cpus:
        if start is None:
            start = prev = c
        elif c == prev + 1:
            prev = c
        else:
            out.append(f"{start}-{prev}" if prev > start else f"{start}")
            start = prev = c
    if start is not None:
        out.append(f"{start}-{prev}" if prev > start else f"{start}")
    return ",".join(out)
