"""bench.py runs the objects beside the headline (placement_search, e2e, placed, e2e_files) in a child process: whatever
happens there -- a flagged frame, a fault of the runtime -- the headline line is printed and the failure recorded."""
import importlib.util
import json
import os
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _args(**kw):
    base = dict(placement=0, e2e=0, placed=0, e2e_files=0, no_cpu_baseline=True)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_a_failing_child_is_recorded_per_object(monkeypatch):
    bench = _bench()
    seen = {}

    def run(cmd, **kw):
        seen["cmd"] = cmd
        return types.SimpleNamespace(returncode=-6, stdout="", stderr="x\nMemory access fault by GPU node-2\nAborted\n")

    monkeypatch.setattr(bench.subprocess, "run", run)
    out = bench.extra_legs_in_child(_args(), True)
    assert set(out) == {"placement_search", "e2e", "placed", "e2e_files"}
    assert all("code -6" in v["error"] and "Memory access fault" in v["error"] for v in out.values())
    assert "--legs-child" in seen["cmd"] and "--legs-all" in seen["cmd"] and "--no-cpu-baseline" in seen["cmd"]


def test_the_childs_line_is_merged_and_only_what_was_asked_for_is_expected(monkeypatch):
    bench = _bench()
    line = {"e2e": {"frames_per_s": 1.0}}
    monkeypatch.setattr(bench.subprocess, "run",
                        lambda cmd, **kw: types.SimpleNamespace(returncode=0, stdout="noise\n" + json.dumps(line) + "\n", stderr=""))
    assert bench.extra_legs_in_child(_args(e2e=512), False) == line

    def boom(cmd, **kw):
        raise bench.subprocess.TimeoutExpired(cmd, 900)

    monkeypatch.setattr(bench.subprocess, "run", boom)
    out = bench.extra_legs_in_child(_args(e2e=512), False)
    assert list(out) == ["e2e"] and "TimeoutExpired" in out["e2e"]["error"]
