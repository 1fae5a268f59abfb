"""Data parallelism for the CoReNet hot path: one process per GPU,
torch.distributed over RCCL/xGMI (backend "nccl" IS RCCL on ROCm), gloo on CPU
for tests.  Replaces the reference's DistributedDataParallel wrapping
(pipeline.py:199-200,224-230) and env-var rank discovery (distributed.py:96-138).

The engine keeps every gradient in ONE contiguous fp32 slab, so the gradient
exchange is a single (optionally chunked) all-reduce of 144.6 MB instead of
DDP's per-bucket copies -- cut into seven ranges that are reduced while backward
still runs (GradientSync.push / wait), each followed by its own Adam launch on an
optimizer stream.  DDP's broadcast_buffers=True (the r/d clamps read the running
statistics, batch_renorm.py:46-49) needs no collective of its own: rank 0's
BatchRenorm buffers ride on the staging tail of the first gradient bucket
(GradientSync.attach); `broadcast_buffers` below is the explicit form, used once
at start-up / after loading a checkpoint."""
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


class NativeComm:
  """RCCL communicator owned by libcorenet_hip.so (crn_comm_*, include/corenet_hip.h): rank 0 draws the unique id,
  the torch.distributed store that init_from_env already set up hands the 128 bytes to the other ranks, every rank
  creates its communicator on its own GPU.  all_reduce(x) enqueues an in-place sum on the CURRENT library stream
  (corenet_amd._lib.stream(): inside the engine's bucket hook that is the side stream, right behind the un-pack)."""

  def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None):
    import ctypes as C
    from corenet_amd import _lib
    self._lib = _lib
    L = _lib.lib()
    if rank is None:
      rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
      world = dist.get_world_size(group) if dist.is_initialized() else 1
    buf = (C.c_char * 128)()
    err = None
    if rank == 0:
      try:
        L.crn_comm_unique_id(C.cast(buf, C.c_void_p))
      except Exception as e:                                     # (librccl missing, no device ...): the other ranks are
        if world == 1:                                           # waiting in the broadcast below -- tell them instead
          raise                                                  # of leaving them there
        err = repr(e)
    if world > 1:
      box = [(err, bytes(buf))]
      # `src` of broadcast_object_list is a GLOBAL rank: rank 0 of a sub-group is not global rank 0
      src = dist.get_global_rank(group, 0) if group is not None else 0
      dist.broadcast_object_list(box, src=src, group=group)      # through the store / the existing process group
      err, raw = box[0]
      if err is not None:
        raise RuntimeError(f"NativeComm: rank 0 of the group could not create the RCCL unique id: {err}")
      buf = (C.c_char * 128).from_buffer_copy(raw)
    comm = C.c_void_p()
    L.crn_comm_init(C.cast(buf, C.c_void_p), rank, world, C.byref(comm))
    self.comm, self.rank, self.world = comm, rank, world
    ver = C.c_int(0)
    L.crn_comm_info(self.comm, C.byref(ver), None, None)
    self.version = ver.value

  def all_reduce(self, x: t.Tensor, stream: Optional[int] = None):
    assert x.is_cuda and x.dtype == t.float32 and x.is_contiguous()
    self._lib.lib().crn_allreduce_f32(self.comm, x.data_ptr(), x.numel(),
                                      self._lib.stream() if stream is None else stream)

  def close(self):
    if self.comm is not None:
      self._lib.lib().crn_comm_destroy(self.comm)
      self.comm = None


class _StreamWork:
  """What dist.all_reduce(async_op=True) returns, for a collective this module enqueued on its own communication
  stream: wait() makes the CURRENT stream wait for it (no host wait)."""

  def __init__(self, done: "t.cuda.Event"):
    self.done = done

  def wait(self):
    t.cuda.current_stream().wait_event(self.done)


class GradientSync:
  """Gradient exchange of the data-parallel step: all-reduce(sum) over RCCL, the division by world_size is
  folded into the Adam kernel's grad_scale.

  Two ways to drive it.  `sync(grads)` reduces the whole flat slab after backward in `chunks` pieces.
  With `overlap` (default, env CRN_OVERLAP_ALLREDUCE=0 turns it off) `CoreNet.train_step` passes `push` to
  `Plan.backward` as the bucket hook: every finished range of the slab (engine.GRAD_BUCKET_LABELS, reverse
  layer order like DDP's buckets, pipeline.py:199) is reduced while the rest of backward still runs, and `wait`
  joins them before Adam.

  Transport: torch.distributed (RCCL on GPUs, gloo in the CPU tests; the default until the native one has run on
  two or more real GPUs), or -- `native=True` / env CRN_NATIVE_RCCL=1 -- the library's own RCCL communicator
  (NativeComm) on a communication stream of its own: the stream that pushed the bucket (the engine's side stream, which
  carries the weight gradients) records an event and goes on, later weight-gradient kernels do not queue behind the
  collective.

  BatchRenorm buffers (`attach(engine)`): DDP broadcasts rank 0's buffers before every forward (broadcast_buffers=True,
  pipeline.py:199-200; the r/d clamps read them, batch_renorm.py:46-49) -- a blocking collective in front of a 8 ms
  step.  Here the 0.22 MB ride on the first gradient bucket of the step BEFORE: the slab has a staging tail that rank 0
  fills with its buffers (already stepped by this step's forward) and every other rank with zeros, the bucket's
  sum-all-reduce covers the tail, and `wait` copies it back into the buffers on every rank: the same values the
  broadcast would deliver, no extra collective.  (num_batches_tracked advances identically on every rank.)"""

  def __init__(self, world_size: int, chunks: int = 4, group=None, overlap: Optional[bool] = None,
               force: bool = False, native: Optional[bool] = None):
    self.world, self.chunks, self.group = world_size, max(1, chunks), group
    if overlap is None:
      overlap = os.environ.get("CRN_OVERLAP_ALLREDUCE", "1") != "0"
    self.overlap = overlap
    self.force = force          # exchange even at world_size 1 (tests: exercises the stream wiring on one GPU)
    self._works = []
    self.pushed = []            # (offset-free) element counts of the buckets of the last step, for tests
    if native is None:
      native = os.environ.get("CRN_NATIVE_RCCL", "0") == "1"
    self.native: Optional[NativeComm] = None
    if native and t.cuda.is_available() and (world_size > 1 or force):
      self.native = NativeComm(group=group) if dist.is_initialized() else NativeComm(0, 1)
    self.engine = None
    self.needs_buffer_broadcast = False
    self._staged = False
    self.probe = False          # bench.py: HIP-event pairs around the wait for every bucket ...
    self.bucket_log = []        # ... one list of (start, end) per step, buckets in push order
    # per-bucket optimizer step behind the exchange (CoreNet.train_step sets it): called with the bucket's gradient slice
    # on `opt_stream` once the bucket's all-reduce has finished there, so Adam runs under the rest of backward on every
    # rank like it does on one GPU, instead of after the last bucket
    self.after_bucket = None
    self._opt_stream = None
    self._opt_done = None
    self._comm_stream = None    # native transport: the collectives' own stream
    self._stepped = False       # optimizer steps are queued on the optimizer stream: wait() joins it
    self._probe_evs = []

  def attach(self, engine):
    """Carry the BatchRenorm buffers of `engine` on the first gradient bucket (see the class docstring).  The piggy-back
    synchronises the ranks at the END of a step; the FIRST step after attach() starts from whatever each rank holds -- so the
    exchange owes one explicit broadcast of rank 0's buffers, which CoreNet.train_step pays before that step's forward
    (`needs_buffer_broadcast`; DDP broadcast_buffers=True semantics, pipeline.py:199-200).

    COLLECTIVE, like wrapping a module in DistributedDataParallel: every rank of the group calls attach() (the flag it sets
    gates a collective in train_step, so it must be set on all ranks or on none).  Nothing rank-local ever sets that flag:
    a `load_state_dict()` / `mark_weights_dirty()` on ONE rank after attach() does not ask for a broadcast (it used
    to, and the other ranks then went straight to their bucket all-reduces -- mismatched collectives, a hang).  What happens
    instead is what the piggy-back does every step: rank 0's buffers, as stepped by its next forward, reach every rank with
    the first bucket of that step (a reload on rank 0 wins one step late; a reload on another rank is overwritten, as under
    DDP).  Ranks that all reload after attach() and want DDP's "before the next forward" exactly call
    `request_buffer_broadcast()` -- on every rank."""
    self.engine = engine
    engine.exchange = self
    self.needs_buffer_broadcast = True
    return self

  def request_buffer_broadcast(self):
    """COLLECTIVE (call on every rank of the group, or on none): the next train_step broadcasts rank 0's BatchRenorm buffers
    before its forward, as after attach()."""
    self.needs_buffer_broadcast = True

  def broadcast_buffers_once(self):
    self.needs_buffer_broadcast = False
    if self.engine is None or not self._active() or not dist.is_initialized():
      return
    src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
    broadcast_buffers(self.engine.store, src, group=self.group)

  def _active(self) -> bool:
    return self.world > 1 or (self.force and (dist.is_initialized() or self.native is not None))

  def _rank(self) -> int:
    if self.native is not None:
      return self.native.rank
    return dist.get_rank(self.group) if dist.is_initialized() else 0

  def _with_buffers(self, grads: t.Tensor) -> t.Tensor:
    """The slice `grads` extended by the staging tail when it is the top of the gradient slab."""
    if self.engine is None or not self._active():
      return grads
    st = self.engine.store
    np_ = st.grads.numel()
    es = st.gslab.element_size()                         # (float64 on the CPU contract emulator)
    lo = (grads.data_ptr() - st.gslab.data_ptr()) // es
    if lo + grads.numel() != np_ or self._staged:        # not the top-of-slab bucket, or staged already this step
      return grads
    if self._rank() == 0:
      st.buf_stage.copy_(st.buffers)
    else:
      st.buf_stage.zero_()
    self._staged = True
    return st.gslab[lo:]

  def _reduce(self, x: t.Tensor):
    if self.native is not None:
      if self._comm_stream is None:
        self._comm_stream = t.cuda.Stream(device=x.device)
      ready, done = t.cuda.Event(), t.cuda.Event()
      ready.record()                                     # the bucket is final on the current stream
      self._comm_stream.wait_event(ready)
      self.native.all_reduce(x, stream=self._comm_stream.cuda_stream)
      done.record(self._comm_stream)
      return _StreamWork(done)
    return dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

  def __call__(self, grads: t.Tensor):
    if not self._active():
      return
    grads = self._with_buffers(grads)
    n = grads.numel()
    step = (n + self.chunks - 1) // self.chunks
    step = (step + 1023) // 1024 * 1024
    works = [self._reduce(grads[o:o + step]) for o in range(0, n, step)]
    for w in works:
      if w is not None:
        w.wait()
    self._unstage()

  def opt_stream(self):
    """The stream of the per-bucket optimizer steps (None on the CPU)."""
    if self._opt_stream is None and t.cuda.is_available() and self.engine is not None and self.engine.device.type == "cuda":
      self._opt_stream = t.cuda.Stream(device=self.engine.device)
      self._opt_done = t.cuda.Event()
    return self._opt_stream

  def push(self, grads: t.Tensor):
    """Bucket hook: start the all-reduce of one finished slice of the slab (ordered after the work already
    queued on the current stream) and return immediately.  With `after_bucket` set, the bucket's optimizer step is
    queued behind the all-reduce on the optimizer stream."""
    self.pushed.append(grads.numel())
    if not self._active():
      if self.after_bucket is not None:
        self.after_bucket(grads)
      return
    w = self._reduce(self._with_buffers(grads))
    ost = self.opt_stream() if self.after_bucket is not None else None
    if ost is None:
      self._works.append(w)
      if self.after_bucket is not None:       # (no streams: CPU dry run -- reduce, then step, in order)
        if w is not None:
          w.wait()
          self._works.pop()
        self.after_bucket(grads)
      return
    from corenet_amd import _lib
    cur = t.cuda.current_stream()
    ev = t.cuda.Event()
    ev.record(cur)
    with t.cuda.stream(ost), _lib.pinned_stream(ost):
      ost.wait_event(ev)
      a = b = None
      if self.probe:
        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        a.record(ost)
      if w is not None:
        w.wait()                              # the optimizer stream (not the host's current one) waits for the collective
      if self.probe:
        b.record(ost)
        self._probe_evs.append((a, b))
      self.after_bucket(grads)
    self._stepped = True

  def wait(self):
    """The current stream waits for every pushed bucket (and for the optimizer steps queued behind them)."""
    if self._stepped:
      self._opt_done.record(self._opt_stream)
      t.cuda.current_stream().wait_event(self._opt_done)
      self._stepped = False
      if self._probe_evs:
        self.bucket_log.append(list(self._probe_evs))
        self._probe_evs = []
      self._works.clear()
      self._unstage()
      return
    evs = []
    for w in self._works:
      if self.probe and t.cuda.is_available():
        a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        a.record()
        if w is not None:
          w.wait()
        b.record()
        evs.append((a, b))
      elif w is not None:
        w.wait()
    self._works.clear()
    if evs:
      self.bucket_log.append(evs)
    self._unstage()

  def exposed_ms_per_bucket(self):
    """Mean time the main stream spent waiting for each bucket (push order) over the probed steps; synchronises."""
    if not self.bucket_log:
      return []
    t.cuda.synchronize()
    n = min(len(e) for e in self.bucket_log)
    return [sum(e[i][0].elapsed_time(e[i][1]) for e in self.bucket_log) / len(self.bucket_log) for i in range(n)]

  def describe(self) -> dict:
    """What the N > 1 bench line reports about the exchange."""
    be = "native RCCL (crn_allreduce_f32 on its own stream)" if self.native is not None else (
        f"torch.distributed/{dist.get_backend(self.group)}" if dist.is_initialized() else "none")
    ver = None
    if self.native is not None:
      ver = self.native.version
    elif dist.is_initialized() and dist.get_backend(self.group) == "nccl":
      ver = ".".join(str(v) for v in t.cuda.nccl.version())
    return {"transport": be, "ranks": self.world, "version": ver, "overlap": bool(self.overlap),
            "buffers_on_first_bucket": self.engine is not None,
            "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO")}

  def _unstage(self):
    if self._staged:
      st = self.engine.store
      st.buffers.copy_(st.buf_stage)       # (every transport: wait() has joined the collectives on this stream)
      self._staged = False


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

