"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI through
``torch.distributed`` (backend "nccl" is RCCL on ROCm; "gloo" for the CPU rehearsal tests).

The reference gets DDP implicitly from Lightning (``fs2/cli/train.py:33-41``).  Here
the gradient already lives in one contiguous buffer laid out in forward order, so the backward
pass -- which completes it from the end towards the beginning -- hands finished slices ("buckets":
PostNet+mel head, each decoder layer, variance adaptor, each encoder layer (+ embedding)) to an asynchronous sum-all-reduce
as soon as their last kernel has been enqueued; the collective runs on RCCL's stream underneath the
remaining backward kernels.  The 1/world_size factor is folded into the clip coefficient of the
fused optimizer, so no extra pass over the gradient is needed.  BatchNorm statistics stay per rank
(the reference does not use SyncBatchNorm).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, store, process_group=None, world_size: Optional[int] = None, force: bool = False):
        """``force``: issue the collectives even with one rank (exercises the RCCL call path on a one-GPU box)."""
        self.store, self.group, self.force = store, process_group, force
        self.world = world_size if world_size is not None else dist.get_world_size(process_group)
        self.ranges = store.bucket_ranges()
        self.by_id = store.bucket_map()  # bucket id -> (start, end), None for an id without parameters
        self._work, self._done = [], set()

    def bucket_ready(self, bucket: int):
        """Called by ``FastSpeech2.backward`` right after the last gradient kernel of ``bucket``."""
        if bucket not in self.by_id:
            raise KeyError(f"GradSync: unknown gradient bucket {bucket} (the store has {sorted(self.by_id)})")
        if bucket in self._done:
            raise RuntimeError(f"GradSync: bucket {bucket} handed over twice in one step")
        self._done.add(bucket)
        if (self.world == 1 and not self.force) or self.by_id[bucket] is None:
            return
        s, e = self.by_id[bucket]
        self._work.append(dist.all_reduce(self.store.grad[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Makes the current stream wait for every outstanding bucket (call before the optimizer).  Raises when a
        bucket that holds parameters was never handed over this step: its gradient would silently stay local."""
        missing = [b for b, r in self.by_id.items() if r is not None and b not in self._done]
        if missing:
            raise RuntimeError(f"GradSync: buckets {missing} were not exchanged in this step")
        for w in self._work:
            w.wait()
        self._work.clear()
        self._done.clear()

    def broadcast_parameters(self, src: int = 0):
        """Rank ``src``'s weights and BatchNorm buffers to every rank (start of training)."""
        if self.world == 1 and not self.force:
            return
        dist.broadcast(self.store.flat, src, group=self.group)
        for b in self.store.buffers.values():
            dist.broadcast(b, src, group=self.group)

    def broadcast_buffers(self, src: int = 0):
        """torch DDP's ``broadcast_buffers=True`` (Lightning's default strategy, fs2/cli/train.py:33-41; SURVEY.md 2.4
        (2)): rank 0's BatchNorm running statistics and step counters reach every rank at the start of each forward.  A
        TRAINING forward never reads them (batch statistics), so the only places a rank's own running statistics could
        be observed are an evaluation forward and a checkpoint -- this is called in front of both (``validation_step``
        's first batch, ``on_save_checkpoint``, the native trainer's validate / save), which gives DDP's observable
        state, rank 0's buffers on every rank, for two dozen tiny broadcasts per validation instead of per step."""
        if self.world == 1 and not self.force:
            return
        for b in self.store.buffers.values():  # (the num_batches_tracked entries are views of store.bn_counters)
            dist.broadcast(b, src, group=self.group)

    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world


def share_tile_table(src: int = 0, process_group=None) -> int:
    """Every rank adopts rank ``src``'s tuned GEMM tiles (``hip.tile_table``): the same tiles mean the same summation
    order on every rank, so ranks fed identical data produce identical bits (debugging) and no rank runs a tile that
    lost the timing race elsewhere.  Returns the number of signatures in the table."""
    from . import hip as H
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return sum(1 for k in H.tile_table() if not k.startswith("__"))  # (not the format stamp / the grouped-launch tiles)
    box = [(H.tile_table(), H.tile_timings()) if dist.get_rank(process_group) == src else None]
    dist.broadcast_object_list(box, src=src, group=process_group)
    if dist.get_rank(process_group) != src:
        H.load_tile_table(box[0][0])
        H.load_tile_timings(box[0][1])  # (the in-step refinement's candidate lists: the same trials on every rank)
    return sum(1 for k in box[0][0] if not k.startswith("__"))


def host_budget(local_world: int) -> dict:
    """What one rank may use of the host when ``local_world`` ranks share it.  A rank is one Python
    thread that enqueues launches plus RCCL's proxy thread; torch's intra-op pool defaults to EVERY core, and eight ranks
    each spinning up a 128-thread pool for a stray host-side op (collation, ``.cpu()`` reductions, checkpoint
    conversion) starve one another's enqueue threads.  ``threads`` = cores // ranks, at least 1; ``workers`` = the most
    DataLoader worker processes a rank should start (cores // ranks - 2 for the enqueue and proxy threads, at least 0)."""
    import os
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        cores = os.cpu_count() or 1
    share = max(1, cores // max(1, int(local_world)))
    return {"cores": cores, "threads": share, "workers": max(0, share - 2)}


def apply_host_budget(local_world: int) -> dict:
    """``torch.set_num_threads`` to this rank's share (call once per rank, before the first step)."""
    b = host_budget(local_world)
    torch.set_num_threads(b["threads"])
    return b
