"""StreamedAugmenter.run hands a lane back to the submitting thread only after ``consume`` has returned: the lane's
output slabs, counters and pinned input belong to the drain thread until then (a lane resubmitted while
``write_files`` still reads it would have its buffers overwritten by the next batch's downloads).  Stub lanes, no GPU."""
import importlib
import threading
import time

streaming = importlib.import_module("pcl-augmentation_amd.streaming")


class _StubLane:
    def __init__(self):
        self.busy = False
        self.tag = None


def _stub_augmenter(n_lanes):
    aug = object.__new__(streaming.StreamedAugmenter)
    aug.lanes = [_StubLane() for _ in range(n_lanes)]
    aug.consuming = set()
    import collections
    aug.times = collections.defaultdict(float)
    aug.lock = threading.Lock()
    aug.violations = []
    aug.order = []

    def submit(lane, scenes, inserts, min_points, tag=None):
        with aug.lock:
            if lane in aug.consuming or aug.lanes[lane].busy:
                aug.violations.append((tag, lane))
        aug.lanes[lane].busy, aug.lanes[lane].tag = True, tag
        return lane

    def wait(lane):                           # (the waiter thread of `run`: the lane's device work)
        time.sleep(0.001)

    def finish_collect(lane):                 # (the drain thread: status, merge, the results' views)
        ln = aug.lanes[lane]
        with aug.lock:
            aug.consuming.add(lane)
        ln.busy = False                       # as the real one does, BEFORE consume runs
        return ln.tag, lane, None

    aug.submit, aug.wait, aug.finish_collect = submit, wait, finish_collect
    return aug


def test_lane_is_not_resubmitted_while_it_is_consumed():
    aug = _stub_augmenter(3)

    def consume(tag, lane, _):
        time.sleep(0.003)                     # write_files reading the lane's buffers
        aug.order.append(tag)
        with aug.lock:
            aug.consuming.discard(lane)

    aug.run(((None, None, None, i) for i in range(60)), consume)
    assert aug.order == list(range(60))
    assert aug.violations == []


def test_error_in_consume_surfaces_and_does_not_deadlock():
    aug = _stub_augmenter(2)

    def consume(tag, lane, _):
        with aug.lock:
            aug.consuming.discard(lane)
        if tag == 3:
            raise RuntimeError("boom")

    try:
        aug.run(((None, None, None, i) for i in range(20)), consume)
    except RuntimeError as e:
        assert "boom" in str(e)
    else:
        raise AssertionError("the error of consume was swallowed")
