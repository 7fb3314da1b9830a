"""Host cores per rank (the reference's analogue is N copies of the script on one host, OD/README.md:36).

The GPU work of the ranks of a node is independent -- scenes are sharded, there is no collective on the data path -- but the
ranks share the HOST: the streamed driver's packer, reader, merge and writer threads (``pack_threads`` / ``io_threads`` per
rank) and the pinned staging slabs the copy engines read.  Unbound, eight ranks x sixteen threads roam over every core of
both sockets, each rank's staging memory ends up on whichever NUMA node first touched it, and half of a rank's PCIe traffic
crosses the socket interconnect.  ``bind_rank`` gives every rank of a node its own slice of the cores the process may use,
on the NUMA node its GPU hangs off when the kernel's KFD topology says which that is (else an even split in NUMA order), and
is called BEFORE the GPU runtime starts: the runtime's helper threads and the first-touch placement of the pinned slabs
inherit the mask.  Everything here reads sysfs; nothing touches a GPU.
"""
from __future__ import annotations

import glob
import os
import re


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def numa_of_cpus():
    """cpu -> NUMA node from /sys/devices/system/node (every cpu on node 0 when that tree is absent)."""
    nodes = {}
    for path in glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"):
        node = int(re.search(r"node(\d+)/cpulist$", path).group(1))
        try:
            for c in _parse_cpulist(open(path).read()):
                nodes[c] = node
        except OSError:
            pass
    return nodes


def numa_of_gpus():
    """NUMA node of every GPU in the order of the KFD topology (= HIP's device order when no *_VISIBLE_DEVICES re-orders it), or
    [] when the topology cannot be read.  A GPU node of /sys/class/kfd/kfd/topology/nodes has simd_count > 0; its PCI address
    (domain, location_id) names the device whose numa_node the PCI tree knows."""
    gpus = []
    for path in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/[0-9]*/properties"),
                       key=lambda p: int(re.search(r"nodes/(\d+)/", p).group(1))):
        try:
            props = dict(ln.split(None, 1) for ln in open(path).read().splitlines() if " " in ln)
        except OSError:
            return []
        if int(props.get("simd_count", "0")) == 0:
            continue
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 0xFF:02x}:{(loc >> 3) & 0x1F:02x}.{loc & 7:x}"
        try:
            node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        except (OSError, ValueError):
            node = -1
        gpus.append(node)
    return gpus


def plan(local_rank, local_world, cpus=None, cpu_node=None, gpu_node=None):
    """The cores of rank ``local_rank`` of ``local_world`` ranks on this host: a sorted list.  ``cpus``: the cores the process
    may use (default: its affinity mask); ``cpu_node``: cpu -> NUMA node; ``gpu_node``: NUMA node per local rank's GPU (-1 /
    missing: unknown).  Ranks whose GPU sits on a node share that node's cores evenly; ranks without a known node (or whose
    node has no usable core) share what is left, in NUMA order.  Every rank gets at least one core; slices are disjoint
    whenever there are at least as many cores as ranks."""
    if not (0 <= local_rank < local_world):
        raise ValueError("local_rank outside [0, local_world)")
    cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
    cpu_node = numa_of_cpus() if cpu_node is None else cpu_node
    gpu_node = (numa_of_gpus() if gpu_node is None else list(gpu_node)) + [-1] * local_world
    by_node = {}
    for c in cpus:
        by_node.setdefault(cpu_node.get(c, 0), []).append(c)
    ranks_on = {}
    for r in range(local_world):
        node = gpu_node[r] if gpu_node[r] in by_node else -1
        ranks_on.setdefault(node, []).append(r)
    slices = {}
    taken = set()
    for node, ranks in ranks_on.items():
        if node < 0:
            continue
        pool = by_node[node]
        for i, r in enumerate(ranks):
            lo, hi = i * len(pool) // len(ranks), (i + 1) * len(pool) // len(ranks)
            slices[r] = pool[lo:hi] or [pool[min(lo, len(pool) - 1)]]
            taken.update(slices[r])
    rest = [c for node in sorted(by_node) for c in by_node[node] if c not in taken] or \
           [c for node in sorted(by_node) for c in by_node[node]]
    loose = ranks_on.get(-1, [])
    for i, r in enumerate(loose):
        lo, hi = i * len(rest) // len(loose), (i + 1) * len(rest) // len(loose)
        slices[r] = rest[lo:hi] or [rest[min(lo, len(rest) - 1)]]
    return sorted(slices[local_rank])


def bind_rank(local_rank, local_world):
    """Restrict this process (and every thread it starts from now on) to its slice of the host's cores.  Call before the GPU
    runtime is initialised.  Returns {"cores": n, "first": c0, "last": c1, "numa_nodes": [...]}; with one rank per host nothing
    is changed."""
    before = sorted(os.sched_getaffinity(0))
    if local_world <= 1:
        return {"cores": len(before), "bound": False}
    mine = plan(local_rank, local_world, before)
    os.sched_setaffinity(0, mine)
    nodes = numa_of_cpus()
    return {"cores": len(mine), "first": mine[0], "last": mine[-1], "numa_nodes": sorted({nodes.get(c, 0) for c in mine}),
            "bound": True, "cores_of_the_host_share": len(before)}
