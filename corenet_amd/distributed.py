"""Data parallelism for the CoReNet hot path: one process per GPU,
torch.distributed over RCCL/xGMI (backend "nccl" IS RCCL on ROCm), gloo on CPU
for tests.  Replaces the reference's DistributedDataParallel wrapping
(pipeline.py:199-200,224-230) and env-var rank discovery (distributed.py:96-138).

The engine keeps every gradient in ONE contiguous fp32 slab, so the gradient
exchange is a single (optionally chunked) all-reduce of 144.6 MB instead of
DDP's per-bucket copies; BatchRenorm buffers are broadcast from rank 0 before
each forward exactly like DDP's broadcast_buffers=True (the r/d clamps read the
running statistics, batch_renorm.py:46-49)."""
from __future__ import annotations

import os
from typing import Optional

import torch as t
import torch.distributed as dist
import torch.utils.data


def init_from_env(backend: Optional[str] = None) -> tuple:
  """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun, dist_launch.py:51-105)."""
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if world > 1 and not dist.is_initialized():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
      # CRN_DIST_BACKEND=gloo: dry runs of the multi-rank path on a box with fewer GPUs than ranks
      backend = os.environ.get("CRN_DIST_BACKEND") or ("nccl" if t.cuda.is_available() else "gloo")
    if backend == "nccl":
      t.cuda.set_device(local)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
  return rank, local, world


class GradientSync:
  """Gradient exchange of the data-parallel step: all-reduce(sum) over RCCL, the division by world_size is
  folded into the Adam kernel's grad_scale.

  Two ways to drive it.  `sync(grads)` reduces the whole flat slab after backward in `chunks` pieces.
  With `overlap` (default, env CRN_OVERLAP_ALLREDUCE=0 turns it off) `CoreNet.train_step` passes `push` to
  `Plan.backward` as the bucket hook: every finished range of the slab (engine.GRAD_BUCKET_LABELS, reverse
  layer order like DDP's buckets, pipeline.py:199) is reduced on RCCL's own stream while the rest of
  backward still runs, and `wait` joins them before Adam."""

  def __init__(self, world_size: int, chunks: int = 4, group=None, overlap: Optional[bool] = None,
               force: bool = False):
    self.world, self.chunks, self.group = world_size, max(1, chunks), group
    if overlap is None:
      overlap = os.environ.get("CRN_OVERLAP_ALLREDUCE", "1") != "0"
    self.overlap = overlap
    self.force = force          # exchange even at world_size 1 (tests: exercises the stream wiring on one GPU)
    self._works = []
    self.pushed = []            # (offset-free) element counts of the buckets of the last step, for tests

  def _active(self) -> bool:
    return self.world > 1 or (self.force and dist.is_initialized())

  def __call__(self, grads: t.Tensor):
    if not self._active():
      return
    n = grads.numel()
    step = (n + self.chunks - 1) // self.chunks
    step = (step + 1023) // 1024 * 1024
    works = []
    for o in range(0, n, step):
      works.append(dist.all_reduce(grads[o:o + step], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
    for w in works:
      w.wait()

  def push(self, grads: t.Tensor):
    """Bucket hook: start the all-reduce of one finished slice of the slab (ordered after the work already
    queued on the current stream) and return immediately."""
    self.pushed.append(grads.numel())
    if self._active():
      self._works.append(dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

  def wait(self):
    """The current stream waits for every pushed bucket."""
    for w in self._works:
      w.wait()
    self._works.clear()


def broadcast_buffers(store, src: int = 0, group=None):
  """DDP broadcast_buffers=True semantics for the BatchRenorm buffers."""
  if dist.is_initialized() and dist.get_world_size(group) > 1:
    dist.broadcast(store.buffers, src, group=group)
    dist.broadcast(store.nbt, src, group=group)


def reduce_confusion_matrix(cm: t.Tensor, dst: int = 0, group=None):
  """evaluation_results.py:256-257."""
  if dist.is_initialized() and dist.get_world_size(group) > 1:
    dist.reduce(cm, dst, op=dist.ReduceOp.SUM, group=group)
  return cm


class DistributedSampler(t.utils.data.Sampler):
  """Rank's share of a dataset (reference distributed.py:203-230): a fixed-seed (0x1234) permutation of the
  dataset, zero-padded to a multiple of the world size when `pad_data`, cut into contiguous per-rank slices."""

  def __init__(self, dataset, global_rank: int, global_world_size: int, pad_data: bool):
    n = len(dataset)
    total_size = (n + global_world_size - 1) // global_world_size * global_world_size if pad_data else n
    g = t.Generator()
    g.manual_seed(0x1234)
    indices = t.randperm(n, generator=g)
    indices = t.constant_pad_nd(indices, [0, total_size - indices.shape[0]])
    start = global_rank * total_size // global_world_size
    end = (global_rank + 1) * total_size // global_world_size
    self.indices = indices[start:end]

  def __iter__(self):
    return iter(self.indices)

  def __len__(self):
    return self.indices.shape[0]

