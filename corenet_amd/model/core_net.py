"""Drop-in for `corenet.model.core_net.CoreNet` (core_net.py:25-61) on MI355X.

Same constructor (`CoreNet(config)` with `config.decoder.{resolution,
num_output_channels,last_upscale_factor,latent_channels,skip_fraction}`), same
`forward(image uint8[B,3,H,W], voxel_projection_matrix f32[B,4,4],
voxel_sample_locations f32[B,3]) -> logits f32[B,C,D,H,W]`, same
`state_dict()` keys and shapes (so the published checkpoints and
`model.encoder.load_state_dict(resnet50_checkpoint)` of state.py:69 load), works
under autograd (`loss.backward()` fills `.grad` of every parameter) and can be
wrapped by DistributedDataParallel.  All arithmetic runs in libcorenet_hip.so.
"""
from __future__ import annotations

import contextlib
import os
import sys
import dataclasses
import math
from typing import Any, Dict, Optional, Tuple

import torch as t
from torch import nn

from corenet_amd import _lib
from corenet_amd.model.engine import Engine, IMAGE_HW, check_image_hw


@dataclasses.dataclass(frozen=True)
class DecoderConfig:
  """Mirror of corenet.configuration.DecoderConfig (configuration.py:278-294)."""
  resolution: Tuple[int, int, int]
  num_output_channels: int
  last_upscale_factor: int = 2
  latent_channels: int = 64
  skip_fraction: float = 0.75


@dataclasses.dataclass(frozen=True)
class CoreNetConfig:
  """Mirror of corenet.configuration.CoreNetConfig (configuration.py:298-299)."""
  decoder: DecoderConfig

  def to_dict(self) -> Dict[str, Any]:
    return {"decoder": dataclasses.asdict(self.decoder)}

  @classmethod
  def from_dict(cls, d) -> "CoreNetConfig":
    dd = dict(d["decoder"])
    dd["resolution"] = tuple(dd["resolution"])
    return cls(decoder=DecoderConfig(**dd))


_EVAL_GRAPH = os.environ.get("CRN_EVAL_GRAPH", "1") != "0"      # inference forwards replay a captured HIP graph
_LIVE_GRAPHS = []      # captured steps; released before interpreter teardown (graph destruction after the HIP
                       # runtime has started to unload faults)


def _release_graphs():
  while _LIVE_GRAPHS:
    try:
      _LIVE_GRAPHS.pop().reset()
    except Exception:
      pass
  try:                                # graphs of models the garbage collector dropped (Engine.orphan_graph_resources)
    from corenet_amd.model.engine import Engine
    Engine.drain_orphans()
  except Exception:
    pass


import atexit  # noqa: E402
atexit.register(_release_graphs)


class _Tree(nn.Module):
  """Nested container so that parameters get the reference's dotted names."""

  def add(self, path, tensor: t.Tensor, kind: str):
    head, *rest = path
    if rest:
      if head not in self._modules:
        self.add_module(head, _Tree())
      self._modules[head].add(rest, tensor, kind)
    elif kind == "param":
      self.register_parameter(head, nn.Parameter(tensor))
    else:
      self.register_buffer(head, tensor)


class _CoreNetFn(t.autograd.Function):
  """Whole-network autograd node: forward/backward are the engine's HIP plans."""

  @staticmethod
  def forward(ctx, model, image, v2s, offset, *params):
    plan = model.engine.plan(image.shape[0], image.shape[2:])
    logits = plan.forward(image, v2s, offset, training=model.training)
    ctx.model, ctx.plan, ctx.generation = model, plan, plan.generation
    return logits.clone()      # the plan's buffer is reused by the next forward

  @staticmethod
  def backward(ctx, glogits):
    model, plan = ctx.model, ctx.plan
    with model._on_device():
      return _CoreNetFn._backward(ctx, model, plan, glogits)

  @staticmethod
  def _backward(ctx, model, plan, glogits):
    if plan.generation != ctx.generation:
      # the saved activations and BatchRenorm statistics of a forward live in the (per batch size) plan: a later
      # forward with the same batch size has overwritten them.  The reference's autograd keeps one set per graph;
      # raising beats returning gradients of the wrong activations.
      raise RuntimeError("corenet_amd.CoreNet: backward() of a forward pass whose saved activations were "
                         "overwritten by a later forward with the same batch size; run backward before the "
                         "next forward (gradient accumulation: forward, backward, forward, backward)")
    plan.glogits.copy_(glogits)          # plan-owned buffer: no per-step view cache entries, no pinned tensors
    plan.backward(plan.glogits)
    # autograd keeps (or accumulates into) what is returned here, so it must not alias the engine's gradient slab,
    # which the next backward overwrites: ONE copy of the slab, handed out as per-parameter views
    store = model.engine.store
    grads = store.grads.clone()
    return (None, None, None, None) + tuple(store.view_of(grads, k) for k in model._param_keys)


class CoreNet(nn.Module):
  """Image to 3D reconstruction with CoReNet (MI355X-native)."""

  def __init__(self, config, device: Optional[str] = None, backend=None, decoder_math: Optional[str] = None):
    """decoder_math: "bf16x3" (default on the GPU; also env CRN_DECODER_MATH) -- the big decoder convolutions and the
    encoder's 3x3 layers on the split-bf16 MFMA engines (engine.BF16X3_LAUNCHES), ~3e-6 relative error per layer,
    inside the 1e-3 logits tolerance on every reference fixture -- or "fp32": every convolution on the fp32 MFMA
    engine, the reference's own arithmetic, at half the speed."""
    super().__init__()
    self.config = config
    dc = config.decoder
    if device is None:
      device = f"cuda:{t.cuda.current_device()}" if t.cuda.is_available() else "cuda"
    self.engine = Engine(dc.num_output_channels, resolution=tuple(dc.resolution),
                         latent_channels=dc.latent_channels, skip_fraction=dc.skip_fraction,
                         last_upscale_factor=dc.last_upscale_factor, device=device, backend=backend,
                         decoder_math=decoder_math)
    self._packed_version = None      # (params, buffers) version counters at the last weight pack of an inference forward
    self._tree_root = _Tree()
    self._param_keys = []
    for key, shape, kind in self.engine.specs:
      self._tree_root.add(key.split("."), self.engine.store.view(key), kind)
      if kind == "param":
        self._param_keys.append(key)
    # expose encoder / decoder exactly like the reference module tree
    self.encoder = self._tree_root._modules["encoder"]
    self.decoder = self._tree_root._modules["decoder"]
    del self._tree_root
    self.reset_parameters()
    self.register_load_state_dict_post_hook(lambda m, k: m._mark_dirty())
    self.encoder.register_load_state_dict_post_hook(lambda m, k: self._mark_dirty())

  def _mark_dirty(self):
    self.engine.weights_dirty = True
    self._packed_version = None              # whatever wrote the slabs: the next inference forward packs again
    # (data parallel: a rank-local reload never raises the exchange's broadcast flag -- that flag gates a collective and
    # is only set by the collective calls GradientSync.attach() / request_buffer_broadcast(); see attach())

  def close(self):
    """Gives back what the inference / training graphs of this model hold outside torch's allocator: the captured graphs with their
    private pools and the capture stream's split-K scratch (a `hipFree`, i.e. a device-wide wait).  EXPLICIT: call it (or use the
    model as a context manager) when a process builds and drops many models; the model stays usable (it captures again).  The
    garbage collector never does this (`__del__` used to -- a device synchronisation on whatever thread the collector
    runs, possibly at interpreter shutdown): a model that is simply dropped parks its graphs and capture stream on a list that the
    next graph capture of any model in the process releases (Engine.orphan_graph_resources / drain_orphans)."""
    eng = getattr(self, "engine", None)
    if eng is None:
      return
    live = {id(g) for p in eng.plans.values() for g in list(p.graphs.values()) + [p.eval_graph] if g is not None}
    _LIVE_GRAPHS[:] = [g for g in _LIVE_GRAPHS if id(g) not in live]
    eng.release_graph_resources()

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()
    return False

  def __del__(self):
    # references only: no graph reset, no hipFree from the collector (see close()); at interpreter shutdown nothing at all
    # (the atexit hook above has reset the live graphs already)
    try:
      eng = getattr(self, "engine", None)
      if eng is not None and not sys.is_finalizing():
        live = {id(g) for p in eng.plans.values() for g in list(p.graphs.values()) + [p.eval_graph] if g is not None}
        _LIVE_GRAPHS[:] = [g for g in _LIVE_GRAPHS if id(g) not in live]
        eng.orphan_graph_resources()
    except Exception:
      pass

  def train(self, mode: bool = True):
    self._packed_version = None              # mode switches re-derive the packed weights once (cheap, and never stale)
    return super().train(mode)

  def _on_device(self):
    """Kernels go to the current stream of the CURRENT device: make that the model's device for the call, and pin
    that stream's handle for the library calls of the block (_lib.pinned_stream)."""
    d = self.engine.device
    if d.type != "cuda":
      return contextlib.nullcontext()
    stack = contextlib.ExitStack()
    stack.enter_context(t.cuda.device(d))
    stack.enter_context(_lib.pinned_stream())
    return stack

  def reset_parameters(self, seed: int = 0):
    """resnet50.py:40-47 (kaiming-normal convs, BN gamma=1 beta=0) and torch's
    default initialisers for the decoder layers, drawn on the host."""
    g = t.Generator().manual_seed(seed)
    s = self.engine.store
    with t.no_grad():
      for key, shape, kind in self.engine.specs:
        v = s.view(key)
        if key.endswith("running_var"):
          v.fill_(1.0)
        elif kind != "param":
          v.zero_()
        elif len(shape) >= 2:
          fan_in = shape[1] * int(math.prod(shape[2:])) if len(shape) > 2 else shape[1]
          if key.startswith("encoder."):
            w = t.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
          else:
            bound = 1.0 / math.sqrt(fan_in)
            w = (t.rand(shape, generator=g) * 2 - 1) * bound
          v.copy_(w)
        elif ".bn." in key or ".b1." in key or ".b2." in key:
          v.fill_(1.0 if key.endswith("weight") else 0.0)
        else:
          v.zero_()
    self._mark_dirty()

  # nn.Module plumbing -------------------------------------------------------
  def _apply(self, fn, recurse=True):
    # Parameters are views into the engine's flat device slabs; moving them
    # would silently detach them from the kernels.
    probe = fn(t.zeros(1, device=self.engine.device))
    same = probe.device.type == self.engine.device.type and (
        probe.device.index is None or self.engine.device.index is None or probe.device.index == self.engine.device.index)
    if not same or probe.dtype != t.float32:
      raise RuntimeError("corenet_amd.CoreNet lives on its construction device in fp32; "
                         "construct it with CoreNet(config, device=...) instead of .to()/.half()")
    return self

  def _check_image(self, image: t.Tensor):
    """Argument checks of resnet50.py:198-199.  The reference's encoder is fully convolutional (resnet50.py:176-186) and so is this
    one: a plan (buffers + launch sequences) is built per (batch size, H, W) on first use.  One restriction is this engine's own: H
    and W must be multiples of 4 (engine.check_image_hw); anything else raises before a kernel runs."""
    assert image.dtype == t.uint8 and image.dim() == 4 and image.shape[1] == 3
    check_image_hw(image.shape[2:])

  def forward(self, image: t.Tensor, voxel_projection_matrix: t.Tensor,
              voxel_sample_locations: t.Tensor) -> t.Tensor:
    # same argument checks as resnet50.py:198-199 / ray_traced_skip_connection.py:81-89
    self._check_image(image)
    B = image.shape[0]
    assert voxel_projection_matrix.shape == (B, 4, 4)
    assert voxel_sample_locations.shape == (B, 3)
    if not image.is_cuda and getattr(self.engine.be, "name", "") != "emu":   # "emu": tests' contract emulator
      raise ValueError("Only CUDA(HIP) tensors are supported by corenet_amd.CoreNet")
    eng = self.engine
    inference = not self.training and not t.is_grad_enabled()
    if inference:
      # Inference with the weights left alone (evaluation loops, super-resolution, serving): the packed / operand forms of
      # the weights are re-derived only when the parameter or buffer slab was written since the last pack -- torch's version
      # counters see every in-place torch write to a parameter (optimizers, load_state_dict, `p.add_()`); writes through
      # `.data` or raw pointers need `mark_weights_dirty()`.  (A training-mode forward re-derives them every call, as before:
      # an external optimizer steps between two of them.)
      # `.data` / raw-pointer / custom-kernel writes bypass the counters and need `mark_weights_dirty()`; CRN_EVAL_WEIGHT_CHECK=N
      # compares a device checksum of both slabs every N-th inference call (a host sync: a debugging aid, off by default)
      ver = (eng.store.params._version, eng.store.buffers._version)
      if ver != self._packed_version or self._weights_moved():
        eng.weights_dirty = True
      self._pending_version = ver              # becomes _packed_version once the forward below has packed
    else:
      eng.weights_dirty = True           # parameters may have been stepped by an external optimizer
      self._packed_version = None
    image = image.contiguous()
    v2s = voxel_projection_matrix.to(t.float32).contiguous()
    off = voxel_sample_locations.to(t.float32).contiguous()
    with self._on_device():
      if t.is_grad_enabled() and self.training:
        params = [self.get_parameter(k) for k in self._param_keys]
        return _CoreNetFn.apply(self, image, v2s, off, *params)
      plan = eng.plan(B, image.shape[2:])
      if inference and image.is_cuda and _EVAL_GRAPH and plan.trace is None and plan.probes is None:
        out = self._forward_eval_graph(plan, image, v2s, off)
      else:
        out = plan.forward(image, v2s, off, training=self.training).clone()
      if inference and not eng.weights_dirty:
        self._packed_version = self._pending_version      # the packed forms now belong to this version of the slabs
      return out

  def _weights_moved(self) -> bool:
    n = int(os.environ.get("CRN_EVAL_WEIGHT_CHECK", "0") or 0)
    if n <= 0:
      return False
    self._eval_calls = getattr(self, "_eval_calls", 0) + 1
    if self._eval_calls % n:
      return False
    st = self.engine.store
    sig = (float(st.params.double().sum()), float(st.params.double().abs().sum()), float(st.buffers.double().sum()))
    moved = getattr(self, "_weight_sig", None) is not None and sig != self._weight_sig
    self._weight_sig = sig
    return moved

  def mark_weights_dirty(self):
    """Parameters or buffers were written behind torch's back (`.data`, raw pointers): the next forward re-derives the packed
    weights."""
    self._mark_dirty()

  def _forward_eval_graph(self, plan, image, v2s, off) -> t.Tensor:
    """The eval-mode forward of a batch size as a captured HIP graph (single stream, ~200 launches): launch by launch the
    host needs 1.0-1.1 ms to enqueue it and the GPU 2.45 ms (B = 4) / 1.42 ms (B = 1) including the weight pack; replayed
    (weights untouched) 2.29 / 1.25 ms with 0.06 ms of host time, same logits bit for bit (round 4, tools/eval_enqueue.py).
    The first call of a batch size runs launch by launch (sizes workspaces, sets kernel attributes), the second captures."""
    eng = self.engine
    if eng.weights_dirty or plan.eval_graph is None:
      # launch by launch: packs the weights if they changed (a graph captured earlier stays valid: it reads the packed forms)
      out = plan.forward(image, v2s, off, training=False).clone()
      plan.eval_eager += 1
      if plan.eval_graph is None and plan.eval_eager >= 2 and not eng.weights_dirty:
        plan.in_image.copy_(image); plan.in_v2s.copy_(v2s); plan.in_off.copy_(off)
        g = t.cuda.CUDAGraph()
        cap = eng.capture_stream()
        cap.wait_stream(t.cuda.current_stream())
        with t.cuda.graph(g, stream=cap), _lib.pinned_stream(cap):
          plan.forward(plan.in_image, plan.in_v2s, plan.in_off, training=False)
        t.cuda.current_stream().wait_stream(cap)
        plan.eval_graph = g
        _LIVE_GRAPHS.append(g)
      return out
    plan.in_image.copy_(image); plan.in_v2s.copy_(v2s); plan.in_off.copy_(off)
    plan.eval_graph.replay()
    plan.generation += 1                         # (a training forward's saved activations are gone, as after any forward)
    return plan.logits.clone()

  # multi-offset inference (super_resolution.py:114-129) ------------------------------------
  def multi_offset_pmf(self, image: t.Tensor, voxel_projection_matrix: t.Tensor, grid_offsets: t.Tensor,
                       resolution_multiplier: int = 0) -> t.Tensor:
    """softmax(model(image, v2s, offset)) for every offset in grid_offsets [n, B, 3].
    The reference re-runs the whole network per offset (super_resolution.py:123-125); in eval mode the
    encoder does not depend on the offset, so it runs ONCE here and only the skip compression, the
    ray sampling and the 3D decoder run per offset -- identical results, n-1 encoder passes saved.
    resolution_multiplier m > 0 (with n == m^3 in the reference's offset order): returns the interleaved
    [B, C, m*D, m*H, m*W] grid directly; 0: returns [n, B, C, D, H, W] like MultiOffsetInferenceFn."""
    assert not self.training, "multi-offset inference is an eval-mode path (running BatchRenorm statistics)"
    self._check_image(image)
    B = image.shape[0]
    assert voxel_projection_matrix.shape == (B, 4, 4)
    assert grid_offsets.dim() == 3 and grid_offsets.shape[1:] == (B, 3)
    if not image.is_cuda:
      raise ValueError("Only CUDA(HIP) tensors are supported by corenet_amd.CoreNet")
    n = grid_offsets.shape[0]
    m = int(resolution_multiplier)
    if m > 0 and n != m ** 3:
      raise ValueError("resolution_multiplier**3 offsets expected")
    ver = (self.engine.store.params._version, self.engine.store.buffers._version)     # (as in forward(): inference keeps its packs)
    if ver != self._packed_version:
      self.engine.weights_dirty = True
    plan = self.engine.plan(B, image.shape[2:])
    v2s = voxel_projection_matrix.to(t.float32).contiguous()
    offs = grid_offsets.to(t.float32).contiguous()
    C = self.engine.num_classes
    D, H, W = plan.logits.shape[2:]
    with t.no_grad():
      plan.forward_encoder(image.contiguous(), training=False)      # packs encoder AND decoder weights when they are dirty
      if not self.engine.weights_dirty:
        self._packed_version = ver
      stash = t.empty((n, B, C, D, H, W), dtype=plan.logits.dtype, device=plan.logits.device)
      for i in range(n):
        stash[i].copy_(plan.forward_decoder(v2s, offs[i], training=False))
      be = self.engine.be
      if m > 0:
        out = t.empty((B, C, m * D, m * H, m * W), dtype=stash.dtype, device=stash.device)
        be.softmax_superres(stash, m, B, C, D, H, W, out)
        return out
      out = t.empty_like(stash)
      be.softmax_superres(stash, 1, n * B, C, D, H, W, out)      # plain channel softmax
      return out

  # fused training step (bench.py / the train hot loop pipeline.py:215-240) ------
  def train_step(self, image: t.Tensor, voxel_projection_matrix: t.Tensor,
                 voxel_sample_locations: t.Tensor, grid: t.Tensor, loss: str = "iou_fgbg",
                 lr: float = 4e-4, adam_eps: float = 1e-4, world_size: int = 1,
                 all_reduce=None, graph: Optional[bool] = None) -> t.Tensor:
    """forward -> loss -> backward -> (gradient all-reduce) -> Adam, without
    autograd bookkeeping.  Returns the loss as a 1-element device tensor (no sync).

    graph (default: off; env CRN_GRAPH=1 or graph=True turns it on): without a gradient exchange the whole step
    is static per batch size, so it can be captured once into a HIP graph (second call) and replayed afterwards.
    The inputs are copied into plan-owned buffers first, and Adam reads its scalars (lr, eps, bias corrections of
    this step count) from a small device buffer refreshed by an ordinary launch before every replay.
    Measured on MI355X / ROCm 7.0 (tools/cpu_enqueue.py, profiles/r02_graph_vs_eager.txt): hipGraphLaunch still
    walks the ~650 kernel nodes on the host (3.6 ms per step vs 6.5 ms launch by launch through ctypes) and the
    replay loses the overlap of the weight-gradient stream with the data-gradient chain (15.2 vs 13.5 ms per
    step), so launch by launch stays the default while the GPU needs more than the host's 6.5 ms per step."""
    from corenet_amd.model.engine import LOSS_KINDS
    eng = self.engine
    self._check_image(image)
    B, C = image.shape[0], eng.num_classes
    if tuple(grid.shape) != (B,) + tuple(eng.resolution):
      raise ValueError(f"grid of shape {tuple(grid.shape)}, expected {(B,) + tuple(eng.resolution)}")
    with self._on_device():
      plan = eng.plan(B, image.shape[2:])
      if all_reduce is not None and getattr(all_reduce, "needs_buffer_broadcast", False):
        all_reduce.broadcast_buffers_once()      # first step after attach() / request_buffer_broadcast(): DDP's broadcast_buffers
      if graph is None:
        graph = os.environ.get("CRN_GRAPH", "0") == "1"
      if all_reduce is not None and hasattr(all_reduce, "_active") and not all_reduce._active():
        # an exchange object with nothing to exchange (world_size 1 without `force`): the step WITHOUT an exchange, whose
        # per-bucket Adam runs on the side stream behind the scalars written there -- not the bucket hooks of the
        # exchange, whose inactive branch would step on the side stream while the scalars (and, on the first step, the
        # zeroed moments) are still in flight on the optimizer stream
        if hasattr(all_reduce, "pushed"):
          all_reduce.pushed.clear()
        all_reduce = None
      graphable = (all_reduce is None and eng.device.type == "cuda" and plan.probes is None and plan.trace is None)
      if graph and graphable:
        return self._train_step_graph(plan, image, voxel_projection_matrix, voxel_sample_locations, grid, loss, lr,
                                      adam_eps, 1.0 / world_size)
      plan.forward(image.contiguous(), voxel_projection_matrix, voxel_sample_locations, training=True)
      if grid.dtype == t.int32 and grid.device == plan.gt.device and grid.is_contiguous() and grid.shape == plan.gt.shape:
        gt = grid                                   # (the loss kernels read it in place: no 33 MB copy per step)
      else:
        plan.gt.copy_(grid); gt = plan.gt           # dtype / device conversion
      eng.be.loss_fwd_bwd(LOSS_KINDS[loss], plan.logits, gt, B, C, plan.logits[0, 0].numel(), plan.loss,
                          plan.glogits, 1.0)
      if all_reduce is not None and getattr(all_reduce, "overlap", False):
        all_reduce.pushed.clear()
        step_buckets = (hasattr(all_reduce, "opt_stream") and plan.side is not None and plan.trace is None
                        and os.environ.get("CRN_BUCKET_ADAM", "1") != "0" and all_reduce.opt_stream() is not None)
        if step_buckets:
          # Adam per bucket behind its all-reduce, on the exchange's optimizer stream: the update runs under the rest of
          # backward on every rank, as it does without an exchange (adam_bucket_hook)
          ost = all_reduce.opt_stream()
          ost.wait_stream(t.cuda.current_stream())                # (first step: the moments are allocated on this stream)
          with t.cuda.stream(ost), _lib.pinned_stream(ost):
            eng.adam_step_graphable(lr, adam_eps, grad_scale=1.0 / world_size, launch=False)
          all_reduce.after_bucket = eng.adam_bucket_hook()
        plan.backward(plan.glogits, grad_hook=all_reduce.push)   # buckets are reduced while backward runs
        plan._probe("grad_exchange_wait", all_reduce.wait)       # what is left of the exchange once backward is done
        if step_buckets:
          all_reduce.after_bucket = None
          eng.weights_dirty = True
          return plan.loss
      elif all_reduce is None and plan.side is not None and plan.trace is None:
        # no exchange: every finished bucket of the grad slab is un-packed AND stepped on the side stream
        # (the step's Adam scalars are written on the stream that runs the bucket updates: one launch off the main chain)
        with t.cuda.stream(plan.side), _lib.pinned_stream(plan.side):
          eng.adam_step_graphable(lr, adam_eps, grad_scale=1.0 / world_size, launch=False)
        plan.backward(plan.glogits, grad_hook=eng.adam_bucket_hook())
        eng.weights_dirty = True
        return plan.loss
      else:
        plan.backward(plan.glogits)
        if all_reduce is not None:
          all_reduce(eng.store.grads)
      eng.adam_step(lr, adam_eps, grad_scale=1.0 / world_size)
      return plan.loss

  def _step_body(self, plan, loss: str):
    """The static part of a fused step on the plan's own input buffers (what the graph captures)."""
    from corenet_amd.model.engine import LOSS_KINDS
    eng = self.engine
    eng.weights_dirty = True                   # the step starts by packing the (just stepped) parameters
    plan.forward(plan.in_image, plan.in_v2s, plan.in_off, training=True)
    eng.be.loss_fwd_bwd(LOSS_KINDS[loss], plan.logits, plan.gt, plan.B, eng.num_classes, plan.logits[0, 0].numel(),
                        plan.loss, plan.glogits, 1.0)
    plan.backward(plan.glogits)
    eng.adam_update_from_hyper()

  def _train_step_graph(self, plan, image, v2s, offset, grid, loss, lr, adam_eps, grad_scale) -> t.Tensor:
    eng = self.engine
    plan.in_image.copy_(image); plan.in_v2s.copy_(v2s); plan.in_off.copy_(offset); plan.gt.copy_(grid)
    eng.adam_step_graphable(lr, adam_eps, grad_scale=grad_scale, launch=False)
    g = plan.graphs.get(loss)
    if g is None and plan.eager_steps < 1:
      # first step: launch by launch (sizes the workspaces, sets kernel attributes, allocates the split-K scratch:
      # nothing of that may happen inside a capture)
      self._step_body(plan, loss)
      plan.eager_steps += 1
      return plan.loss
    if g is None:
      g = t.cuda.CUDAGraph()
      cap = eng.capture_stream()
      cap.wait_stream(t.cuda.current_stream())
      with t.cuda.graph(g, stream=cap), _lib.pinned_stream(cap):      # (the library calls follow the capture stream)
        self._step_body(plan, loss)
      t.cuda.current_stream().wait_stream(cap)
      plan.graphs[loss] = g
      _LIVE_GRAPHS.append(g)
    g.replay()
    eng.weights_dirty = True                   # for whatever runs next outside the graph (eval forward, autograd path)
    return plan.loss
