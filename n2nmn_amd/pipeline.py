"""The serving pipeline bench.py times and tests/test_gpu_bench_config.py checks: S workers, each a
host thread + HIP stream + forked context (shared weights, n2nmn_ctx_fork) + two super-buckets of
KCAP slots that it alternates between, so a pass does not find its inputs in the caches the previous
pass left.

The reference serves one batch of 64 per `sess.partial_run` pair (exp_clevr/eval_clevr.py:103-135);
questions are independent, so here the batches of several clients share every launch
of a pass (n2nmn_amd/superbucket.py) and S passes are in flight on S streams.  Nothing synchronises
inside a pass: a timed block releases the workers and waits for them.
"""
from __future__ import annotations

import threading
from typing import List, Optional, Sequence, Tuple

from .engine import _torch
from .nmn3_assembler import Assembler
from .spec import Dims
from .superbucket import SuperBucket


class PassPipeline:
    def __init__(self, dims: Dims, assembler: Assembler, weights, streams: int = 2, kcap: int = 16,
                 device: int = 0, host_assemble: bool = False, mode: Optional[str] = None):
        """dims.N = questions per client batch.  mode: recurrent-step tile ('latency', 'throughput',
        'throughput_ksplit'); default 'throughput' when several batches share a pass."""
        torch = _torch()
        self.dims, self.S, self.KCAP = dims, max(1, int(streams)), max(1, int(kcap))
        self.host_assemble = host_assemble
        first = SuperBucket(dims, assembler, self.KCAP, device=device)
        first.load_weights(weights)
        self.engine = first.engine
        self.device = self.engine.device
        self.mode = mode or ('throughput' if (self.S > 1 or self.KCAP > 1) else 'latency')
        engines = [self.engine] + [self.engine.fork() for _ in range(self.S - 1)]
        self.workers = []
        for si, e in enumerate(engines):
            e.set_mode(self.mode)
            bk = [first if si == 0 else SuperBucket(dims, assembler, self.KCAP, device=device, engine=e),
                  SuperBucket(dims, assembler, self.KCAP, device=device, engine=e)]
            self.workers.append(dict(engine=e, buckets=bk, next=0, todo=[],
                                     stream=torch.cuda.Stream(device=self.device) if self.S > 1 else None))
        self._threads: List[threading.Thread] = []
        self._start = threading.Barrier(self.S + 1)
        self._done = threading.Barrier(self.S + 1)
        self._state = {'stop': False, 'gt': False, 'err': None}
        self.eos_retire = False        # inference option N2NMN_S2S_EOS_RETIRE of the passes (SuperBucket.run)
        if self.S > 1:
            for wk in self.workers:
                t = threading.Thread(target=self._worker_main, args=(wk,), daemon=True)
                t.start()
                self._threads.append(t)

    # ---- filling ------------------------------------------------------------------------------
    def bucket(self, worker: int, j: int) -> SuperBucket:
        return self.workers[worker]['buckets'][j]

    def fill_all(self, make_inputs, make_layout):
        """slot k of bucket j of worker si <- make_inputs(i), make_layout(i), i = (2 si + j) KCAP + k"""
        for si, wk in enumerate(self.workers):
            for j, b in enumerate(wk['buckets']):
                for k in range(self.KCAP):
                    i = (si * 2 + j) * self.KCAP + k
                    b.fill(k, make_inputs(i), make_layout(i))

    # ---- running ------------------------------------------------------------------------------
    def _run_worker(self, wk, widths: Sequence[int], gt: bool):
        """passes of the given widths (slots) on one worker, alternating its two buckets"""
        for n in widths:
            b = wk['buckets'][wk['next'] % 2]
            wk['next'] += 1
            b.run(use_gt_layout=gt, n_slots=n, host_assemble=self.host_assemble,
                  eos_retire=self.eos_retire and not self.host_assemble)

    def _worker_main(self, wk):
        torch = _torch()
        torch.cuda.set_device(self.device)
        while True:
            self._start.wait()
            if self._state['stop']:
                return
            try:
                with torch.cuda.stream(wk['stream']):
                    self._run_worker(wk, wk['todo'], self._state['gt'])
                    wk['stream'].synchronize()
            except Exception as ex:           # surface worker failures instead of hanging
                self._state['err'] = ex
            self._done.wait()

    def run(self, widths_per_worker: Sequence[Sequence[int]], gt: bool):
        """worker si runs passes of widths_per_worker[si] slots, the workers concurrently; returns
        when every worker's stream has drained"""
        if self.S == 1:
            self._run_worker(self.workers[0], widths_per_worker[0], gt)
            return
        for wk, widths in zip(self.workers, widths_per_worker):
            wk['todo'] = list(widths)
        self._state['gt'] = gt
        self._start.wait()
        self._done.wait()
        if self._state['err'] is not None:
            err, self._state['err'] = self._state['err'], None
            raise err

    @staticmethod
    def split(count: int, K: int, kcap: int) -> List[int]:
        """`count` client batches as passes of (nearly) equal width: about K slots each, never more
        than a bucket holds"""
        if count <= 0:
            return []
        n_pass = max(1, int(round(count / K)))
        while -(-count // n_pass) > kcap:
            n_pass += 1
        out, done = [], 0
        for pi in range(n_pass):
            n = (count - done + (n_pass - pi) - 1) // (n_pass - pi)
            out.append(n)
            done += n
        return out

    def plan(self, count: int, K: int) -> List[List[int]]:
        """exactly `count` client batches, split as evenly as possible over the workers"""
        return [self.split(count // self.S + (1 if i < count % self.S else 0), K, self.KCAP)
                for i in range(self.S)]

    def close(self):
        if self._threads:
            self._state['stop'] = True
            self._start.wait()
            for t in self._threads:
                t.join(5.0)
            self._threads = []
