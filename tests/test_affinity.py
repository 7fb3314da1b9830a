"""CPU: the per-rank core plan of a multi-GPU node (pcl-augmentation_amd/affinity.py) -- pure arithmetic on what sysfs says,
injected here: two sockets, eight GPUs, with and without a known GPU -> NUMA map; odd shares; fewer cores than ranks."""
import os

import pytest


def test_plan_splits_the_numa_nodes_among_the_ranks_on_them(pkg):
    A = pkg.affinity
    cpus = list(range(128))
    cpu_node = {c: (0 if c < 64 else 1) for c in cpus}
    gpu_node = [0, 0, 0, 0, 1, 1, 1, 1]
    got = [A.plan(r, 8, cpus, cpu_node, gpu_node) for r in range(8)]
    assert all(len(g) == 16 for g in got)
    assert sorted(c for g in got for c in g) == cpus                       # disjoint, nothing left out
    for r, g in enumerate(got):
        assert {cpu_node[c] for c in g} == {gpu_node[r]}                   # every rank on its GPU's node
    # the same cores whatever order the ranks ask in (every rank computes the whole plan from sysfs alone)
    assert got[5] == A.plan(5, 8, list(reversed(cpus)), cpu_node, gpu_node)


def test_plan_without_a_gpu_map_and_with_odd_shares(pkg):
    A = pkg.affinity
    cpus = [c for c in range(40) if c not in (3, 17)]                       # a cgroup's odd mask, two nodes
    cpu_node = {c: c // 20 for c in range(40)}
    got = [A.plan(r, 3, cpus, cpu_node, []) for r in range(3)]
    assert sorted(c for g in got for c in g) == cpus and all(len(g) in (12, 13) for g in got)
    # one GPU's node unknown, the others known: the known ones keep to their nodes, the loose rank gets what is left
    got = [A.plan(r, 3, cpus, cpu_node, [0, -1, 0]) for r in range(3)]
    assert {cpu_node[c] for c in got[0]} == {0} and {cpu_node[c] for c in got[2]} == {0} and {cpu_node[c] for c in got[1]} == {1}
    assert not set(got[0]) & set(got[2])
    # fewer cores than ranks: everybody still gets a core
    assert all(len(A.plan(r, 4, [7, 9], {7: 0, 9: 0}, [])) >= 1 for r in range(4))
    with pytest.raises(ValueError):
        A.plan(4, 4, cpus, cpu_node, [])


def test_bind_rank_restricts_this_process_and_one_rank_changes_nothing(pkg):
    A = pkg.affinity
    before = sorted(os.sched_getaffinity(0))
    try:
        assert A.bind_rank(0, 1)["bound"] is False and sorted(os.sched_getaffinity(0)) == before
        if len(before) >= 2:
            info = A.bind_rank(1, 2)
            now = sorted(os.sched_getaffinity(0))
            assert info["bound"] and info["cores"] == len(now) and set(now) < set(before)
            assert now == A.plan(1, 2, before)
    finally:
        os.sched_setaffinity(0, before)
