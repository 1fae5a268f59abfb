"""Drop-in for `corenet.state` checkpoint (de)serialisation (state.py:30-97).

A checkpoint written by the reference (`encode_state`: torch.save of global_step / model_state / model_config /
optimizer_state / extra_metadata, state.py:74-83) loads into the MI355X model, and a checkpoint written here
loads into the reference: same keys, same tensor shapes, same torch.optim.Adam state-dict layout.

The optimizer is `FusedAdam`: torch.optim.Adam's arithmetic (amsgrad=False, weight_decay=0) executed by ONE HIP
launch over the engine's flat parameter slab (`crn_adam_step`), with torch.optim.Adam's state_dict format.
"""
import dataclasses
import io
from typing import Any, Dict

import torch as t

from corenet_amd.model import core_net


class FusedAdam:
  """torch.optim.Adam look-alike bound to a corenet_amd CoreNet (state.py:65: Adam(model.parameters(),
  lr=initial_learning_rate, eps=adam_epsilon)).  Moments live in the engine's flat slabs."""

  def __init__(self, model: core_net.CoreNet, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
    self.model = model
    self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False,
                              params=list(range(len(model._param_keys))))]
    self.gather_copies = 0

  # -- torch.optim.Optimizer surface used by the reference's train loop (pipeline.py:224-230) --
  def _shared_grad_slab(self, params):
    """The ONE tensor all `.grad`s are slices of, laid out like the engine's slab -- or None.  After loss.backward()
    the gradients are per-parameter views of one slab-shaped clone (core_net._CoreNetFn._backward); AccumulateGrad
    stores `new_grad.detach()`, which drops `._base`, so the views are recognised by their storage: same
    untyped storage, contiguous, and storage offsets that differ from the slab offsets by one constant."""
    store = self.model.engine.store
    g0 = params[0].grad
    if g0 is None or g0.dtype != store.grads.dtype or g0.device != store.grads.device:
      return None
    st = g0.untyped_storage()
    c0 = g0.storage_offset() - store.offset(self.model._param_keys[0])
    n = store.grads.numel()
    if c0 < 0 or (c0 + n) * g0.element_size() > st.nbytes():
      return None
    sp = st.data_ptr()
    for p, k in zip(params, self.model._param_keys):
      g = p.grad
      if (g is None or g.dtype != g0.dtype or not g.is_contiguous() or g.untyped_storage().data_ptr() != sp
          or g.storage_offset() - store.offset(k) != c0):
        return None
    return t.empty(0, dtype=g0.dtype, device=g0.device).set_(st, c0, (n,))

  # -- torch.optim.Optimizer surface used by the reference's train loop (pipeline.py:224-230) --
  def zero_grad(self, set_to_none: bool = False):
    """torch 1.7 semantics by default (zero in place, pipeline.py:225); like torch.optim, the gradients are the
    parameters' `.grad` tensors -- the engine's slab is only the staging area of step()."""
    params = [self.model.get_parameter(k) for k in self.model._param_keys]
    flat = None if set_to_none else self._shared_grad_slab(params)
    if flat is not None:
      flat.zero_()                                  # one launch instead of ~270
    else:
      for p in params:
        if p.grad is not None:
          if set_to_none:
            p.grad = None
          else:
            p.grad.detach_()
            p.grad.requires_grad_(False)
            p.grad.zero_()
    self.model.engine.store.grads.zero_()

  def _gather_grads(self):
    """`.grad` of the model's parameters -> the engine's flat gradient slab.  After loss.backward() the
    gradients are views of one slab-shaped tensor (_shared_grad_slab): one copy; anything else (gradients
    replaced by the user, DistributedDataParallel buckets with gradient_as_bucket_view, ...) is
    copied per parameter.  Parameters without a gradient are skipped by torch.optim.Adam; the fused kernel
    steps the whole slab, so they get a zero gradient (their moments decay like Adam's would not -- the
    reference never trains with frozen parameters)."""
    store = self.model.engine.store
    slab = store.grads
    params = [self.model.get_parameter(k) for k in self.model._param_keys]
    if all(p.grad is None for p in params):
      return                                        # the fused path (CoreNet.train_step) wrote the slab itself
    flat = self._shared_grad_slab(params)
    if flat is not None:
      if flat.data_ptr() != slab.data_ptr():
        slab.copy_(flat)
      self.gather_copies = 1                        # (observable by tests)
      return
    self.gather_copies = 0
    for p, k in zip(params, self.model._param_keys):
      v = store.view(k, grad=True)
      if p.grad is None:
        v.zero_()
      elif p.grad.data_ptr() != v.data_ptr():
        v.copy_(p.grad); self.gather_copies += 1

  def step(self, grad_scale: float = 1.0):
    g = self.param_groups[0]
    self._gather_grads()
    self.model.engine.adam_step(g["lr"], g["eps"], betas=g["betas"], grad_scale=grad_scale)

  def _moments(self):
    eng = self.model.engine
    if eng.adam_m is None:
      eng.adam_m = t.zeros_like(eng.store.params)
      eng.adam_v = t.zeros_like(eng.store.params)
    return eng.adam_m, eng.adam_v

  def _slices(self):
    s = self.model.engine.store
    for i, key in enumerate(self.model._param_keys):
      o, shape = s.off[key]
      n = 1
      for d in shape: n *= d
      yield i, o, n, shape

  def state_dict(self) -> Dict[str, Any]:
    eng = self.model.engine
    state = {}
    if eng.adam_t > 0:
      m, v = self._moments()
      for i, o, n, shape in self._slices():
        state[i] = {"step": t.tensor(float(eng.adam_t)), "exp_avg": m[o:o + n].view(shape).clone(),
                    "exp_avg_sq": v[o:o + n].view(shape).clone()}
    return {"state": state, "param_groups": [dict(g) for g in self.param_groups]}

  def load_state_dict(self, sd: Dict[str, Any]):
    eng = self.model.engine
    groups = sd["param_groups"]
    if len(groups) != 1 or len(groups[0]["params"]) != len(self.model._param_keys):
      raise ValueError("optimizer state does not match the model's parameter list")
    g = dict(groups[0]); g["betas"] = tuple(g["betas"]); g["params"] = list(range(len(self.model._param_keys)))
    self.param_groups = [g]
    m, v = self._moments()
    m.zero_(); v.zero_()
    steps = set()
    ids = list(groups[0]["params"])            # ids in the saved state map to positions (torch semantics)
    for i, o, n, shape in self._slices():
      st = sd["state"].get(ids[i])
      if st is None:
        continue
      if tuple(st["exp_avg"].shape) != tuple(shape):
        raise ValueError(f"optimizer state of parameter {self.model._param_keys[i]} has the wrong shape")
      m[o:o + n].copy_(st["exp_avg"].reshape(-1)); v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
      steps.add(int(st["step"]))
    if len(steps) > 1:
      raise ValueError("per-parameter Adam step counts differ; the fused optimizer keeps one")
    eng.adam_t = steps.pop() if steps else 0


@dataclasses.dataclass
class State:
  """state.py:30-35."""
  global_step: int
  model: core_net.CoreNet
  optimizer: Any
  extra_metadata: Any


@dataclasses.dataclass
class SavedState:
  """state.py:38-44."""
  global_step: int
  model_state: Dict[str, Any]
  model_config: Dict[str, Any]
  optimizer_state: Dict[str, Any]
  extra_metadata: Any


def encode_state(state: State) -> bytes:
  """state.py:74-83."""
  saved = SavedState(global_step=state.global_step, model_state=state.model.state_dict(),
                     model_config=state.model.config.to_dict(), optimizer_state=state.optimizer.state_dict(),
                     extra_metadata=state.extra_metadata)
  d = {k.name: getattr(saved, k.name) for k in dataclasses.fields(saved)}
  buf = io.BytesIO()
  t.save(d, buf)
  return buf.getvalue()


def decode_state(raw_state: bytes, device: str, backend=None) -> State:
  """state.py:86-97.  The model is built on `device` directly (its parameters are views of the engine's device
  slabs, so there is no .to(device) step)."""
  d = t.load(io.BytesIO(raw_state), map_location="cpu", weights_only=False)
  saved = SavedState(**d)
  model = core_net.CoreNet(core_net.CoreNetConfig.from_dict(saved.model_config), device=device, backend=backend)
  model.load_state_dict(saved.model_state)
  optimizer = FusedAdam(model)
  optimizer.load_state_dict(saved.optimizer_state)
  return State(global_step=saved.global_step, model=model, optimizer=optimizer, extra_metadata=saved.extra_metadata)
