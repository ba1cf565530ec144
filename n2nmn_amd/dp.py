"""Data-parallel protocol of the forward path: one process per GPU, questions sharded by rank, no
data-path collective.  The process group is used only to bracket timed regions
(barrier) and to take the max elapsed time over ranks.  Backend 'nccl' (= RCCL) on GPUs, 'gloo' in
the CPU tests."""
from __future__ import annotations

import os
import time
from typing import Callable, Optional


class DataParallel:
    def __init__(self, backend: Optional[str] = None, device=None):
        self.rank = int(os.environ.get('RANK', '0'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.device = device
        self._dist = None
        if self.world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ or \
                os.environ.get('N2NMN_FORCE_PROCESS_GROUP') == '1':
            # under torch.distributed.run a 1-rank job still builds its (RCCL) group, so the
            # single-GPU run exercises the same collective calls as the 8-GPU run
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            kw = {}
            if backend == 'nccl' and device is not None:
                kw['device_id'] = device
            dist.init_process_group(backend or 'nccl', rank=self.rank, world_size=self.world, **kw)
            self._dist = dist

    # every rank streams its own questions: rank r, local batch i -> global batch seed
    def batch_seed(self, i: int) -> int:
        return self.rank * 1000 + i

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self._dist is None:
            return value
        import torch
        t = torch.tensor([value], dtype=torch.float64,
                         device=self.device if self.device is not None else 'cpu')
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def min_over_ranks(self, value: float) -> float:
        return -self.max_over_ranks(-value)

    def group_size(self) -> int:
        """ranks the process group actually holds (1 without a group): what a multi-GPU line reports as
        n_gpus, instead of the number it was asked for"""
        return int(self._dist.get_world_size()) if self._dist is not None else 1

    def timed(self, run: Callable[[], None], sync: Callable[[], None]) -> float:
        """barrier + sync, run, sync + barrier; returns the max elapsed seconds over ranks (the fastest
        rank's time of the same region is kept in `last_min_elapsed`)."""
        sync()
        self.barrier()
        t0 = time.perf_counter()
        run()
        sync()
        elapsed = time.perf_counter() - t0
        self.barrier()
        self.last_min_elapsed = self.min_over_ranks(elapsed)
        return self.max_over_ranks(elapsed)

    def throughput(self, units_per_rank: int, elapsed: float) -> float:
        """whole-job units/sec: every rank processed `units_per_rank` in `elapsed` (max) seconds"""
        return self.world * units_per_rank / elapsed

    def close(self):
        if self._dist is not None:
            self._dist.barrier()
            self._dist.destroy_process_group()
            self._dist = None
