"""The throughput structure of the detection hot path on one GPU: three stages on three streams and three host threads,
every stage ONE batched call per step of B frames (todhip_orb_batch_device | the matcher's single DB pass for the B x Q
descriptors | todhip_verify_batch_device). Step s of the matcher overlaps ORB of steps s+1.. and the verifier of step
s-1 (a frame costs ~0.07 ms of full-chip matrix work plus ~1.3 ms of latency-bound ORB/verifier launches,
so the batch -- not the frame -- has to be the unit of every launch).

The stages are callables, so the same loop serves bench.py's headline (stages fed from resident synthetic inputs), its
data-chained run (each stage consumes the previous one's device buffers) and the multi-rank run (the matcher stage is
tod_amd/sharded.py's ShardedMatcher)."""
import threading
import time
from concurrent.futures import ThreadPoolExecutor


class StagePipeline:
    """orb(i) -> int            blocking, on the ORB thread (None: no ORB stage); returns keypoints found
       match(i, n_steps) -> s   issues step i's matcher work on the calling thread; s = torch stream on which its
                                outputs become complete
       verify(i) -> int         blocking, on the verifier thread (None: no verifier); called after `done event` of
                                match(i) has been made a dependency of the verifier's stream via wait_for(i, event)
       depth                    ring depth D of the matcher's output buffers: ORB runs up to D steps ahead of the
                                matcher, the matcher up to D - 1 steps ahead of the verifier (step i waits for verify(i - D))
       match_needs_next_orb     the matcher of step i issues work that reads ORB(i + 1)'s outputs (the overlapped
                                multi-rank step gathers the next step's descriptors one step early)"""

    def __init__(self, torch, orb=None, match=None, verify=None, wait_for=None, depth=3, match_needs_next_orb=False,
                 verify_workers=1, orb_workers=1):
        self.torch, self.orb, self.match, self.verify, self.wait_for = torch, orb, match, verify, wait_for
        self.D, self.next_orb = depth, match_needs_next_orb
        # orb_workers > 1: consecutive steps' ORB calls overlap the same way (the callable picks its context by i % orb_workers)
        self.opool = ThreadPoolExecutor(orb_workers) if orb else None
        self.olocks = [threading.Lock() for _ in range(max(orb_workers, 1))]
        # verify_workers > 1: consecutive steps' verifier calls overlap (each worker has its own context and stream; the
        # callables pick theirs by i % verify_workers) -- the verifier is latency bound, two batches in flight fill its gaps
        self.vpool = ThreadPoolExecutor(verify_workers) if verify else None
        self.vlocks = [threading.Lock() for _ in range(max(verify_workers, 1))]   # one call in flight per context
        self.stat_lock = threading.Lock()                  # stage_s is updated from the ORB thread, the verifier workers and the caller
        self.stage_s = {"orb": 0.0, "match_issue": 0.0, "verify": 0.0}
        self.n_kp = self.n_poses = self.n_steps = 0

    def reset_stats(self):
        with self.stat_lock:
            for key in self.stage_s:
                self.stage_s[key] = 0.0

    def _orb_task(self, i):
        with self.olocks[i % len(self.olocks)]:
            t = time.perf_counter()
            n = self.orb(i)
            with self.stat_lock:
                self.stage_s["orb"] += time.perf_counter() - t
        return n

    def _verify_task(self, i, ev):
        with self.vlocks[i % len(self.vlocks)]:
            self.wait_for(i, ev)                          # device-side edge: this step's matcher outputs
            ev.synchronize()                              # host side too, so that the stage time is the verifier's own
            t = time.perf_counter()
            n = self.verify(i)
            with self.stat_lock:
                self.stage_s["verify"] += time.perf_counter() - t
        return n

    def run(self, n_steps):
        D = self.D
        ofut, vfut = {}, {}
        if self.orb:
            for j in range(min(D, n_steps)):
                ofut[j] = self.opool.submit(self._orb_task, j)

        def orb_done(j):                                  # wait for ORB batch j, keep ORB D batches ahead
            if self.orb and j in ofut:
                self.n_kp += ofut.pop(j).result()
                if j + D < n_steps:
                    ofut[j + D] = self.opool.submit(self._orb_task, j + D)

        for i in range(n_steps):
            if self.verify and i - D in vfut:
                self.n_poses += vfut.pop(i - D).result()  # buffer set i % D is free again
            orb_done(i)
            if self.next_orb:
                orb_done(i + 1)
            t = time.perf_counter()
            out_stream = self.match(i, n_steps)
            with self.stat_lock:
                self.stage_s["match_issue"] += time.perf_counter() - t
            if self.verify:
                ev = self.torch.cuda.Event()
                ev.record(out_stream)                     # the matcher outputs of this step are complete after this
                vfut[i] = self.vpool.submit(self._verify_task, i, ev)
        for i in sorted(vfut):
            self.n_poses += vfut[i].result()
        self.n_steps += n_steps

    def close(self):
        for p in (self.opool, self.vpool):
            if p:
                p.shutdown()
