"""CoReNet forward/backward engine: the host-side plan that strings the HIP
kernels together for core_net.CoreNet (core_net.py:36-43), i.e.
preprocess -> ResNet-50 encoder (resnet50.py:176-186) -> decoder with
ray-traced skips (reconstruction_decoder.py:119-152), and the matching
backward pass.

Python here only allocates buffers (torch), builds views / packed-weight index
tables, and issues C-ABI calls through a backend object; every arithmetic
operation on tensors happens inside libcorenet_hip.so.

Dataflow choices (MI355X-first):
  * all parameters live in ONE flat fp32 slab (one Adam launch, one RCCL
    all-reduce over one contiguous gradient slab), exposed to PyTorch under the
    reference's state_dict names/shapes;
  * conv outputs are stored pre-normalisation; BatchRenorm-apply + ReLU are fused
    into the consumer conv's LDS staging, so normalised activations never touch HBM;
  * decoder stages write straight into the concat buffer of the next stage and
    the ray-sample kernel fills the skip channels in place (no torch.cat copies);
  * strided / transposed convolutions are views (views.py), never materialised.
"""
from __future__ import annotations

import os

import dataclasses
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch as t

from corenet_amd import _lib
from corenet_amd import views as V
from corenet_amd.backend import Transform
from corenet_amd.model import conv_geometry as G

IMAGE_HW = (256, 256)   # the default input size (all configs/models/*.json5); plans are built per (batch, H, W)


def check_image_hw(hw):
  """The encoder is fully convolutional (resnet50.py:176-186); this engine's stem reads a 2 x 2 space-to-depth view of the image and
  its max-pool kernel pools whole 2 x 2 cells, so H and W must be multiples of 4 (the reference takes any size)."""
  H, W = int(hw[0]), int(hw[1])
  if H < 32 or W < 32 or H % 4 or W % 4 or H * W > 4096 * 4096:
    raise ValueError(f"corenet_amd.CoreNet: image batch of {H}x{W} pixels; this engine takes H and W that are multiples of 4 and >= 32 "
                     f"(256x256 in every shipped config)")
  return H, W
BN_EPS = 1e-3        # resnet50.py:63, reconstruction_decoder.py:44
BN_MOMENTUM = 0.01   # batch_renorm.py:19

ENC_STAGES = (("stage2", "abc", (64, 64, 256), 1), ("stage3", "abcd", (128, 128, 512), 2),
              ("stage4", "abcdef", (256, 256, 1024), 2), ("stage5", "abc", (512, 512, 2048), 2))
# Gradient buckets of the overlapped exchange, in the order backward finishes them; a label is the prefix of
# the first (lowest-offset) parameter of the bucket, the last bucket runs down to offset 0.  pipeline.py:199
# gets the same effect from DDP's reverse-order 25 MB buckets.
GRAD_BUCKET_LABELS = ("decoder.stage_3.", "decoder.stage_0.", "encoder.stage5.c.", "encoder.stage5.b.",
                      "encoder.stage5.a.", "encoder.stage4.a.", "encoder.stage2.a.", "")
# (the last bucket is the stem alone: what follows the step's last weight gradient -- un-pack, exchange, Adam -- is the exposed tail of
# the step, so encoder stages 2-3 are handed over when their last block is done, under the stem's own backward)
# Layers (and directions) that run on the split-bf16 MFMA engine when Engine(decoder_math="bf16x3"): the
# Conv3d k5 / ConvTranspose3d k7 of decoder stages 3-6 (reconstruction_decoder.py:64-95): forward, data gradient
# and weight gradient.  Decoder stages 0-2 and the stem stay on the fp32 MFMA engine, the encoder's 3x3 layers go to
# the encoder engine (Engine._build_operands); measured per layer: profiles/r02_layer_times_bf16x3.txt.
BF16X3_LAUNCHES = frozenset(
    [(f"decoder.stage_{k}.{l}.", d) for k in (3, 4, 5, 6) for l in ("c1", "t1") for d in ("fwd", "dgrad")] +
    [(f"decoder.stage_{k}.{l}.", "wgrad") for k in (3, 4, 5, 6) for l in ("c1", "t1")])
# MEASUREMENT ONLY: bits 1 / 2 / 4 = the forward weight packs / the data-gradient weight pack / the gradient
# un-packs are NOT launched -- the kernels then read stale packed weights and Adam steps on stale gradients, so every result is WRONG; it
# prices what the step would gain without these copies (master weights kept in the kernels' layout).  Never set outside a timing run.
_TIMING_ONLY_SKIP_COPIES = int(os.environ.get("CRN_TIMING_ONLY_SKIP_COPIES", "0"))
LOSS_KINDS = {"iou_fgbg": 0, "xent_times_iou_agnostic": 1, "iou_agnostic": 2, "xent": 3,
              "xent_times_iou_fgbg": 4}


def param_specs(num_classes: int, latent: int = 64, skip_fraction: float = 0.75):
  """(key, shape, kind) in the reference's state_dict order (core_net.CoreNet)."""
  specs = []
  def conv(p, co, ci, *k):
    specs.append((p + "weight", (co, ci) + tuple(k), "param"))
    specs.append((p + "bias", (co,), "param"))
  def bn(p, c):
    specs.append((p + "weight", (c,), "param"))
    specs.append((p + "bias", (c,), "param"))
    specs.append((p + "running_mean", (c,), "buffer"))
    specs.append((p + "running_var", (c,), "buffer"))
    specs.append((p + "num_batches_tracked", (), "nbt"))
  e = "encoder."
  conv(e + "stage1.conv.", 64, 3, 7, 7)
  bn(e + "stage1_part2.bn.", 64)
  cin = 64
  for name, blocks, f, _ in ENC_STAGES:
    for bl in blocks:
      p = f"{e}{name}.{bl}."
      conv(p + "op_a.conv.", f[0], cin, 1, 1); bn(p + "op_a.bn.", f[0])
      conv(p + "op_b.conv.", f[1], f[0], 3, 3); bn(p + "op_b.bn.", f[1])
      conv(p + "op_c.conv.", f[2], f[1], 1, 1); bn(p + "op_c.bn.", f[2])
      if bl == "a":
        conv(p + "shortcut.conv.", f[2], cin, 1, 1); bn(p + "shortcut.bn.", f[2])
      cin = f[2]
  d = "decoder."
  specs.append((d + "stage_0.weight", (latent, 2048), "param"))
  specs.append((d + "stage_0.bias", (latent,), "param"))
  bn(d + "stage_1.b1.", latent + 3)
  specs.append((d + "stage_1.t1.weight", (latent + 3, 256, 4, 4, 4), "param"))
  specs.append((d + "stage_1.t1.bias", (256,), "param"))
  chans = {2: (256, 128, 3, 3), 3: (128, 64, 5, 7), 4: (64, 32, 5, 7), 5: (32, 16, 5, 7),
           6: (16, num_classes, 5, 7)}
  src_c = {2: 2048, 3: 1024, 4: 512, 5: 256}
  cin = 256
  for st in range(2, 7):
    cmid, cout, k1, k2 = chans[st]
    p = f"{d}stage_{st}."
    bn(p + "b1.", cin)
    conv(p + "c1.", cmid, cin, k1, k1, k1)
    bn(p + "b2.", cmid)
    specs.append((p + "t1.weight", (cmid, cout, k2, k2, k2), "param"))
    specs.append((p + "t1.bias", (cout,), "param"))
    if st in src_c:
      sk = round(cout * skip_fraction)
      conv(f"{d}rt_skip_{st}.compress_channels.", sk, src_c[st] + 3, 1, 1)
      cin = cout + sk
  return specs


class ParamStore:
  """Flat fp32 parameter / gradient / buffer slabs with reference-named views."""

  def __init__(self, specs, device, dtype=t.float32):
    self.specs = specs
    self.device = device
    self.off: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
    np_, nb, nn = 0, 0, 0
    for key, shape, kind in specs:
      n = int(np.prod(shape)) if len(shape) else 1
      n4 = (n + 3) // 4 * 4        # keep every tensor 16-B aligned inside the slab
      if kind == "param":
        self.off[key] = (np_, shape); np_ += n4
      elif kind == "buffer":
        self.off[key] = (nb, shape); nb += n4
      else:
        self.off[key] = (nn, shape); nn += 1
    self.kind = {k: kd for k, _, kd in specs}
    self.params = t.zeros(np_, dtype=dtype, device=device)
    # gradient slab + a staging tail of the size of the buffer slab: the data-parallel exchange carries rank 0's
    # BatchRenorm buffers on the first (top-of-slab) gradient bucket (distributed.GradientSync)
    self.gslab = t.zeros(np_ + nb, dtype=dtype, device=device)
    self.grads = self.gslab[:np_]
    self.buf_stage = self.gslab[np_:]
    self.buffers = t.zeros(nb, dtype=dtype, device=device)
    self.nbt = t.zeros(nn, dtype=t.int64, device=device)

  def slab(self, key):
    return {"param": self.params, "buffer": self.buffers, "nbt": self.nbt}[self.kind[key]]

  def view(self, key, grad=False) -> t.Tensor:
    o, shape = self.off[key]
    n = int(np.prod(shape)) if len(shape) else 1
    slab = self.grads if grad else self.slab(key)
    return slab[o:o + n].view(shape)

  def offset(self, key) -> int:
    return self.off[key][0]

  def view_of(self, slab: t.Tensor, key) -> t.Tensor:
    """The parameter `key`'s slice of any tensor laid out like the parameter slab."""
    o, shape = self.off[key]
    n = int(np.prod(shape)) if len(shape) else 1
    return slab[o:o + n].view(shape)


@dataclasses.dataclass
class BN:
  """One BatchRenorm instance bound to the slabs + its per-step work buffers."""
  C: int
  gamma: t.Tensor
  beta: t.Tensor
  rmean: t.Tensor
  rvar: t.Tensor
  nbt: t.Tensor
  dgamma: t.Tensor
  dbeta: t.Tensor
  scale: t.Tensor
  shift: t.Tensor
  saved: t.Tensor


@dataclasses.dataclass
class Conv:
  """One reference conv layer bound to its packed weights."""
  name: str
  fwd: G.Geom
  dgrad: Optional[G.Geom]
  wf: t.Tensor              # packed forward weight      (slice of eng.packed)
  wd: Optional[t.Tensor]    # packed data-gradient weight
  bias: t.Tensor            # packed bias [npad]
  gwf: t.Tensor             # packed weight gradient     (slice of eng.gpacked)
  dbias: t.Tensor           # reference-layout bias grad (slice of the grad slab)
  n_ref: int                # reference output channels (bias length)
  wop_f: Optional[t.Tensor] = None   # forward / data-gradient weights pre-split to bf16 hi + lo and pre-arranged
  wop_d: Optional[t.Tensor] = None   # (slices of eng.wop): MFMA operand blocks for the encoder engine ("e2d"),
  wop_kind: str = ""                 # slab images for the decoder's bf16x3 engine ("slab"); None: not used
  ct_f: Optional[t.Tensor] = None    # step-ordered weight images of the parity-walk kernels (csrc/convt_par.hip; decoder stage_6.t1
  ct_d: Optional[t.Tensor] = None    # with more than 8 classes): forward / data gradient
  ct_kind: str = ""                  # "par": the parity walk (> 8 classes, also the weight gradient); "res": resident weights (2 classes)


_ORPHANS = []      # (backend, graphs, capture stream) of engines dropped by the garbage collector: Engine.orphan_graph_resources


class Engine:
  """Builds and runs the forward/backward plan for a fixed per-GPU batch size."""

  def __init__(self, num_classes: int, resolution=(128, 128, 128), latent_channels: int = 64,
               skip_fraction: float = 0.75, last_upscale_factor: int = 2, device="cuda",
               backend=None, dtype=t.float32, decoder_math: Optional[str] = None):
    if tuple(resolution) != (128, 128, 128) or last_upscale_factor != 2:
      # the reference decoder only runs at 128^3 with factor 2
      raise ValueError("the CoReNet decoder is only defined for resolution 128^3, "
                       "last_upscale_factor 2 (reconstruction_decoder.py:38-54,93-95)")
    if backend is None:
      from corenet_amd.backend import default_backend
      backend = default_backend()
    self.be = backend
    # "bf16x3" (default on the GPU): BF16X3_LAUNCHES and the encoder's 3x3 layers multiply operands split into two bf16
    #   terms -- three bf16 MFMAs per product, fp32 accumulation, ~2^-16 relative per product.  Signed off as the product
    #   default in round 3: eval logits 1e-4 against the oracle on every voxel at the bench batch for
    #   C=2 and C=14, train logits within north_star's 1e-3 on every reference fixture, every parameter gradient of the
    #   well-conditioned fixture element by element (tests/test_model_gpu.py) -- and it is what bench.py measures.
    # "fp32": v_mfma_f32_16x16x4_f32 everywhere, the reference's own arithmetic (it is fp32 end to end); half the speed.
    default_math = "bf16x3" if getattr(backend, "name", "") == "hip" else "fp32"
    self.decoder_math = decoder_math or os.environ.get("CRN_DECODER_MATH", default_math)
    if self.decoder_math not in ("fp32", "bf16x3"):
      raise ValueError(f"decoder_math must be 'fp32' or 'bf16x3', not {self.decoder_math!r}")
    # in bf16x3 mode the encoder's stride-1 1x1 / 3x3 convs (forward and data gradient) run on the encoder
    # engine (csrc/conv_e2d.hip: the same split-bf16 products, weights pre-arranged as MFMA operand blocks)
    self.encoder_e2d = self.decoder_math == "bf16x3" and os.environ.get("CRN_E2D", "1") != "0"
    self.wgrad_2d = os.environ.get("CRN_WG2D", "1") != "0"
    self.defer_reduce = os.environ.get("CRN_DEFER_REDUCE", "1") != "0"
    # decoder data gradients leave the two sums of the following BatchRenorm backward (crn_conv_fwd_bf3_slabs_bnbwd)
    self.bn_bwd_fuse = os.environ.get("CRN_BN_BWD_FUSE", "1") != "0"
    # ... except for these layers (measured per layer, profiles/r05_bn_bwd_fuse.txt; CRN_BN_BWD_FUSE_SKIP overrides, comma separated)
    self.bn_bwd_fuse_skip = frozenset(x for x in os.environ.get("CRN_BN_BWD_FUSE_SKIP", "").split(",") if x)
    self.fuse_tail = os.environ.get("CRN_FUSE_TAIL", "1") != "0"
    # fp32 is the only dtype of the HIP kernels; float64 exists so that the CPU
    # contract emulator (tests/) can check the host wiring far below fp32 noise.
    assert dtype == t.float32 or getattr(backend, "name", "") == "emu"
    self.dtype = dtype
    self.device = t.device(device)
    if self.device.type == "cuda" and self.device.index is None:      # "cuda" -> the current device, indexed
      self.device = t.device("cuda", t.cuda.current_device() if t.cuda.is_available() else 0)
    self.num_classes = num_classes
    self.latent = latent_channels
    self.resolution = tuple(resolution)
    self.specs = param_specs(num_classes, latent_channels, skip_fraction)
    self.store = ParamStore(self.specs, self.device, dtype)
    self.skip_ch = {2: round(128 * skip_fraction), 3: round(64 * skip_fraction),
                    4: round(32 * skip_fraction), 5: round(16 * skip_fraction)}
    self._build_layers()
    self.plans: Dict[int, "Plan"] = {}
    self.adam_m: Optional[t.Tensor] = None
    self.adam_v: Optional[t.Tensor] = None
    self.adam_t = 0
    self.adam_hyper: Optional[t.Tensor] = None
    self.dgrad_dirty = True
    self.weights_dirty = True
    self.exchange = None           # distributed.GradientSync.attach(): the data-parallel exchange that carries this engine's buffers
    self._side = False             # created on first use (side_stream)
    self._capture = None           # created on first use (capture_stream)
    self._pack_ev = self._dgrad_packed = self._dec_packed = self._enc_late_packed = None
    self._enc_late_pending = False     # the side stream still owes the main stream: encoder stage 3-5 weights,
    self._dec_pack_pending = False     # ... the decoder's forward weights,
    self._dgrad_pack_pending = False   # ... the data-gradient weights;
    self._gpacked_zeroed = False       # the packed gradient slab was zeroed (on the side stream) for the next backward

  def orphan_graph_resources(self):
    """What `CoreNet.__del__` does instead of release_graph_resources(): hand the engine's graphs and capture stream to a
    module-level list WITHOUT touching the HIP runtime (no graph reset, no hipFree from the garbage collector: ADVICE r5).
    The next capture_stream() of any engine -- an ordinary call on the thread that drives the GPU -- releases them."""
    graphs = [g for p in self.plans.values() for g in list(p.graphs.values()) + [p.eval_graph] if g is not None]
    for p in self.plans.values():
      p.graphs, p.eval_graph, p.eval_eager = {}, None, 0
    if graphs or self._capture is not None:
      _ORPHANS.append((self.be, graphs, self._capture))
    self._capture = None

  @staticmethod
  def drain_orphans():
    while _ORPHANS:
      be, graphs, stream = _ORPHANS.pop()
      for g in graphs:
        try:
          g.reset()
        except Exception:
          pass
      if stream is not None:
        be.splitk_release(stream)

  def capture_stream(self):
    """The ONE stream every HIP graph of this engine is captured on (fused training steps, inference forwards of every batch
    size): a capture stream owns a split-K scratch slot (>= 16 MB, 64 per process) and a BatchRenorm workspace, so one per
    captured graph would leak both.  Replays run on the caller's stream; graphs of one engine are replayed one
    after the other."""
    Engine.drain_orphans()       # scratch slots / graphs of models the garbage collector dropped since the last capture
    if self._capture is None:
      self._capture = t.cuda.Stream(device=self.device)
    # (every call, i.e. before every capture:) as large as the largest split-K scratch any stream of this device has needed so far
    # -- the eager runs that precede a capture have sized the main and the side stream's scratch for exactly these launches.  A
    # capture stream's slot is pinned: if a later capture (a larger batch size) needs more, the outgrown buffer is kept for the
    # graphs captured before (crn_splitk_reserve) and everything goes back in release_graph_resources().  (Round 4 reserved the
    # library's maximum, 256 MB, for every engine that ever replayed an inference graph: ADVICE r4.)
    self.be.splitk_reserve(self._capture, floats=0)
    return self._capture

  def release_graph_resources(self):
    """Resets every graph captured for this engine's plans and gives the capture stream's scratch back (CoreNet.close())."""
    for p in self.plans.values():
      for g in list(p.graphs.values()) + ([p.eval_graph] if p.eval_graph is not None else []):
        try:
          g.reset()
        except Exception:
          pass
      p.graphs, p.eval_graph, p.eval_eager = {}, None, 0
    if self._capture is not None:
      self.be.splitk_release(self._capture)
      self._capture = None

  def side_stream(self):
    """The engine's second HIP stream (weight gradients, weight packs, gradient buckets) or None (CPU emulator,
    CRN_SIDE_STREAM=0); one per engine, shared by the plans of all batch sizes."""
    if self._side is False:
      use = self.device.type == "cuda" and os.environ.get("CRN_SIDE_STREAM", "1") != "0"
      self._side = t.cuda.Stream(device=self.device) if use else None
      if use:
        self._pack_ev, self._dgrad_packed = t.cuda.Event(), t.cuda.Event()
        self._dec_packed, self._enc_late_packed = t.cuda.Event(), t.cuda.Event()
    return self._side

  # ------------------------------------------------------------------ layers
  def _bn(self, prefix: str) -> BN:
    s = self.store
    c = s.off[prefix + "weight"][1][0]
    dev = self.device
    # scale / shift live in two slabs shared by all instances: eval mode fills them in one launch (_bn_slabs)
    o = self._bn_slab_off[prefix]
    return BN(c, s.view(prefix + "weight"), s.view(prefix + "bias"), s.view(prefix + "running_mean"),
              s.view(prefix + "running_var"), s.view(prefix + "num_batches_tracked"),
              s.view(prefix + "weight", grad=True), s.view(prefix + "bias", grad=True),
              self.bn_scale[o:o + c], self.bn_shift[o:o + c],
              t.zeros(4 * c, device=dev, dtype=self.dtype))

  def _bn_slabs(self):
    """Offsets of every BatchRenorm instance in the shared scale / shift slabs (16-byte aligned) and the table of
    crn_batch_renorm_eval_affine: one row per channel."""
    s = self.store
    self._bn_slab_off, rows, n = {}, [], 0
    for key, shape, kind in self.specs:
      if not key.endswith("running_mean"):
        continue
      p = key[:-len("running_mean")]
      c = shape[0]
      self._bn_slab_off[p] = n
      ch = np.arange(c)
      rows.append(np.stack([s.offset(p + "weight") + ch, s.offset(p + "bias") + ch, s.offset(p + "running_mean") + ch,
                            s.offset(p + "running_var") + ch, n + ch], 1))
      n += (c + 3) // 4 * 4
    self.bn_scale = t.zeros(n, device=self.device, dtype=self.dtype)
    self.bn_shift = t.zeros(n, device=self.device, dtype=self.dtype)
    self.bn_eval_table = t.as_tensor(np.concatenate(rows).astype(np.int32), device=self.device)

  def bn_eval_affine(self):
    """Eval-mode scale / shift of every BatchRenorm from the running statistics (batch_renorm.py:59): 1 launch."""
    self.be.bn_eval_affine(self.store.params, self.store.buffers, self.bn_eval_table, BN_EPS, self.bn_scale,
                           self.bn_shift)

  def _build_layers(self):
    s = self.store
    self.convs: Dict[str, Conv] = {}
    self.bns: Dict[str, BN] = {}
    self._bn_slabs()
    reg: List[Tuple[str, G.Geom, Optional[G.Geom], int, int]] = []   # name, fwd, dgrad, repeat, nref

    def add(name, fwd, dgrad, repeat=1):
      nref = s.off[name + "bias"][1][0]
      reg.append((name, fwd, dgrad, repeat, nref))

    for key, shape, kind in self.specs:
      if kind != "param" or not key.endswith("weight") or len(shape) < 4:
        if key.endswith("running_mean"):
          p = key[:-len("running_mean")]
          self.bns[p] = self._bn(p)
        continue
      name = key[:-len("weight")]
      if name == "encoder.stage1.conv.":
        add(name, G.stem_fwd(shape, 3), None)
      elif name == "decoder.stage_1.t1.":
        add(name, G.convt_1to4_fwd(shape), G.convt_1to4_dgrad(shape), repeat=64)
      elif ".t1." in name:
        p = 1 if shape[2] == 3 else 3
        add(name, G.convt_fwd(shape, p), G.convt_dgrad(shape, p), repeat=8)
      else:
        p = shape[-1] // 2
        add(name, G.conv_fwd(shape, p), G.conv_dgrad(shape, p))
    # one gather builds every packed weight / bias from the flat parameter slab
    idx_parts, sizes = [], []
    for name, fwd, dgrad, repeat, nref in reg:
      wo, bo = s.offset(name + "weight"), s.offset(name + "bias")
      def shift(ix, o):
        return np.where(ix >= 0, ix.astype(np.int64) + o, -1)
      parts = [shift(fwd.index, wo), shift(G.bias_index(nref, repeat, fwd.npad, parity_major=(repeat == 8)), bo)]
      if dgrad is not None:
        # a data-gradient layout that IS the reference layout (the 1x1 convolutions: [Cout][Cin], no padding -- 12 M of the 30 M
        # weights of the plain convolutions) is not packed at all: the kernels read the parameter slab (an empty part here)
        alias = (os.environ.get("CRN_DGRAD_ALIAS", "1") != "0" and len(dgrad.index) == int(np.prod(s.off[name + "weight"][1]))
                 and np.array_equal(dgrad.index, np.arange(len(dgrad.index))))
        parts.append(np.zeros(0, np.int64) if alias else shift(dgrad.index, wo))
      idx_parts.append(parts)
    flat_len = sum(len(p) for parts in idx_parts for p in parts)
    assert max(int(p.max()) for parts in idx_parts for p in parts if len(p)) < 2 ** 31
    self.packed = t.zeros(flat_len, dtype=self.dtype, device=self.device)
    gsize = sum(len(parts[0]) for parts in idx_parts)
    self.gpacked = t.zeros(gsize, dtype=self.dtype, device=self.device)
    # 8x8-tile descriptors of the pack (reference -> packed) and un-pack (packed grad -> reference grad)
    # copies: crn_copy_tiles_f32 moves both sides in full 32-byte sectors and reads ~0.5 B of index per element
    # instead of 4 (conv_geometry.tile_index)
    pack_parts, pack_is_bwd, pack_names, unpack_parts = [], [], [], []
    po, go = 0, 0
    for (name, fwd, dgrad, repeat, nref), parts in zip(reg, idx_parts):
      pack_parts.append((po, parts[0], fwd.npad, 0)); po += len(parts[0]); pack_is_bwd.append(False); pack_names.append(name)
      pack_parts.append((po, parts[1], len(parts[1]), 0)); po += len(parts[1]); pack_is_bwd.append(False); pack_names.append(name)
      if dgrad is not None and len(parts[2]):
        pack_parts.append((po, parts[2], dgrad.npad, dgrad.taps if dgrad.taps > 1 else 0)); po += len(parts[2])
        pack_is_bwd.append(True); pack_names.append(name)
      unpack_parts.append((go, parts[0], fwd.npad, 0)); go += len(parts[0])
    dev = self._tiles_dev
    # two packs: what forward reads (forward weights + biases) and what only backward reads (data-gradient
    # weights) -- the second one runs on the side stream under the forward pass, and never in eval mode
    self.pack_tiles = dev([pp for pp, bwd in zip(pack_parts, pack_is_bwd) if not bwd])
    # ... and the forward pack once more in two pieces: the encoder's weights are needed at once, the decoder's
    # ~1.4 ms later, so in training the decoder piece is packed on the side stream under the encoder
    sel = lambda pred: [pp for pp, bwd, nm in zip(pack_parts, pack_is_bwd, pack_names) if not bwd and pred(nm)]
    early = lambda nm: nm.startswith("encoder.stage1") or nm.startswith("encoder.stage2")
    self.pack_tiles_enc = dev(sel(lambda nm: nm.startswith("encoder.")))
    self.pack_tiles_enc_early = dev(sel(early))                                   # stem + stage2: 0.2 M
    self.pack_tiles_enc_late = dev(sel(lambda nm: nm.startswith("encoder.") and not early(nm)))  # 23 M
    self.pack_tiles_dec = dev(sel(lambda nm: not nm.startswith("encoder.")))
    self.pack_tiles_bwd = dev([pp for pp, bwd in zip(pack_parts, pack_is_bwd) if bwd])
    self.unpack_tiles = dev(unpack_parts)
    # gradient buckets for the overlapped exchange (corenet_amd/distributed.py): contiguous ranges of the
    # grad slab in the order backward completes them, each with the tile descriptors of its own convs
    los = [min(o for k, (o, _) in s.off.items() if s.kind[k] == "param" and k.startswith(lb))
           for lb in GRAD_BUCKET_LABELS[:-1]] + [0]
    self.grad_buckets: List[Tuple[str, int, int]] = []
    self._bucket_tiles = []
    hi = s.params.numel()
    for lb, lo in zip(GRAD_BUCKET_LABELS, los):
      assert lo < hi
      self.grad_buckets.append((lb, lo, hi))
      sel = [up for (name, *_), up in zip(reg, unpack_parts) if lo <= s.offset(name + "weight") < hi]
      self._bucket_tiles.append(sel)
      hi = lo
    self._bucket_dev = None
    po, go = 0, 0
    for (name, fwd, dgrad, repeat, nref), parts in zip(reg, idx_parts):
      nwf, nb = len(parts[0]), len(parts[1])
      wf = self.packed[po:po + nwf]; po += nwf
      bias = self.packed[po:po + nb]; po += nb
      wd = None
      if dgrad is not None:
        if len(parts[2]):
          wd = self.packed[po:po + len(parts[2])]; po += len(parts[2])
        else:
          wd = s.view(name + "weight").view(-1)          # (the reference layout is the data-gradient layout: see above)
      gwf = self.gpacked[go:go + nwf]; go += nwf
      self.convs[name] = Conv(name, fwd, dgrad, wf, wd, bias, gwf,
                              s.view(name + "bias", grad=True), nref)
    self._build_operands(reg, idx_parts)

  def _build_ct_images(self):
    """Weight images of the parity-walk kernels of decoder stage_6.t1 (ConvTranspose3d 16 -> C, k 7, stride 2: csrc/convt_par.hip),
    gathered straight from the flat parameter slab after every weight pack (crn_bf3_gather_image).  Used with more than 8 classes
    (m7 / m9): the walk multiplies 16 columns per parity, with the 2 classes of h7 the generic engine's one 16-column block for
    all 8 parities executes fewer MFMAs (CRN_CT_PAR=0 turns the kernels off, CRN_CT_PAR_MIN sets the class threshold)."""
    self.ct_tables = {}
    name = "decoder.stage_6.t1."
    if (self.decoder_math != "bf16x3" or self.device.type != "cuda" or os.environ.get("CRN_CT_PAR", "1") == "0"
        or name not in self.convs or not hasattr(self.be, "convt_par_fwd")):
      return
    shape = tuple(self.store.off[name + "weight"][1])
    if shape[0] != 16 or shape[1] > 16 or tuple(shape[2:]) != (7, 7, 7) or tuple(self.resolution) != (128, 128, 128):
      return
    if shape[1] == 2 and os.environ.get("CRN_CT_RES", "1") != "0":
      # two classes (h7): 8 parities x 2 classes are ONE 16-column block and the layer's weights stay resident in LDS
      kind_, tabs = "res", (G.convt_res_fwd_table, G.convt_res_dgrad_table)
    elif shape[1] >= int(os.environ.get("CRN_CT_PAR_MIN", "9")):
      kind_, tabs = "par", (G.convt_par_fwd_table, G.convt_par_dgrad_table)
    else:
      return
    self.convs[name].ct_kind = kind_
    wofs = self.store.offset(name + "weight")
    for kind, fn, field in (("dec", tabs[0], "ct_f"), ("bwd", tabs[1], "ct_d")):
      tab, nbytes = fn(shape, wofs)
      img = t.zeros(nbytes, dtype=t.uint8, device=self.device)
      setattr(self.convs[name], field, img)
      self.ct_tables[kind] = (t.as_tensor(tab, device=self.device), img, None)

  def _build_operands(self, reg, idx_parts):
    """Operand blocks of the encoder engine: which layers, where in eng.wop, and the layer tables of the
    conversion launches that follow each weight pack (conv_geometry.operand_table)."""
    self.wop = None
    self.op_tables = {}
    self._build_ct_images()
    if self.decoder_math != "bf16x3":
      return
    groups = {"enc_early": [], "enc_late": [], "dec": [], "bwd": []}
    slices = []
    po, eo = 0, 0
    # Which layers: measured (tools/e2d_parity.py, profiles/r02_e2d_parity.txt).  The B=1 / nbt=0 training fixtures
    # amplify any rounding difference ~400x (the fp32 engine itself sits 5.7e-4 from the reference there); the
    # forward pass of stages 2-3 in split-bf16 would raise that to 1.2e-3, stages 4-5 leave it where it is, and
    # the data gradient does not enter the logits at all.  The 1x1 layers gain nothing over the fp32 pointwise
    # kernel yet (13 vs 14 us) and stay there.  CRN_E2D_KINDS=all / CRN_E2D_FWD_STAGES=2345: tuning aids.
    kinds = os.environ.get("CRN_E2D_KINDS", "3x3")
    for (name, fwd, dgrad, repeat, nref), parts in zip(reg, idx_parts):
      src_f = po; po += len(parts[0]) + len(parts[1])
      src_d = po
      if dgrad is not None:
        po += len(parts[2])
      if (name, "fwd") in BF16X3_LAUNCHES and os.environ.get("CRN_BF3_SLABS", "1") != "0":
        # decoder bf16x3 layers: slab images of the forward and the data-gradient weights
        groups["dec"].append((src_f, eo, fwd, True))
        slices.append((name, "wop_f", eo, G.slab_entries(fwd))); eo += G.slab_entries(fwd)
        groups["bwd"].append((src_d, eo, dgrad, True))
        slices.append((name, "wop_d", eo, G.slab_entries(dgrad))); eo += G.slab_entries(dgrad)
        self.convs[name].wop_kind = "slab"
        continue
      if not self.encoder_e2d or not name.startswith("encoder.") or name.startswith("encoder.stage1"):
        continue
      self.convs[name].wop_kind = "e2d"
      only_dgrad = kinds == "3x3+dgrad1x1" and fwd.window != (1, 3, 3)      # tuning aid
      if kinds == "3x3" and fwd.window != (1, 3, 3):
        continue
      fwd_stages = os.environ.get("CRN_E2D_FWD_STAGES", "45")
      if G.operand_eligible(fwd) and name[len("encoder.stage")] in fwd_stages and not only_dgrad:
        groups["enc_early" if name.startswith("encoder.stage2") else "enc_late"].append((src_f, eo, fwd))
        slices.append((name, "wop_f", eo, G.operand_entries(fwd))); eo += G.operand_entries(fwd)
      if dgrad is not None and len(parts[2]) and G.operand_eligible(dgrad):     # (not the layers whose data-gradient layout is the slab's)
        groups["bwd"].append((src_d, eo, dgrad))
        slices.append((name, "wop_d", eo, G.operand_entries(dgrad))); eo += G.operand_entries(dgrad)
    if not eo:
      return
    self.wop = t.zeros(eo * 32, dtype=t.uint8, device=self.device)
    for name, field, e0, n in slices:
      setattr(self.convs[name], field, self.wop[e0 * 32:(e0 + n) * 32])
    for k, layers in groups.items():
      if layers:
        desc, blocks = G.operand_table(layers)
        self.op_tables[k] = (t.as_tensor(desc, device=self.device), blocks)

  def _operands(self, *groups):
    for k in groups:
      if k in self.op_tables:
        self.be.bf3_operands(self.packed, self.op_tables[k], self.wop)
      if k in self.ct_tables:
        tab, img, _ = self.ct_tables[k]
        self.be.bf3_gather_image(self.store.params, tab, img)

  def _tiles_dev(self, parts):
    """Device tables of one pack / un-pack: the parts that are plain transposes as LDS blocks (conv_geometry.mat_index,
    crn_copy_mats_f32: runs of 64+ floats on both sides), the rest as 8x8 tiles (tile_index).  CRN_COPY_MATS=0: tiles
    only (the round-2 copies)."""
    mats, rest = (G.mat_index(parts) if os.environ.get("CRN_COPY_MATS", "1") != "0" else
                  (np.zeros((0, 16), np.int32), parts))
    tl = G.tile_index(rest)
    return (t.as_tensor(tl[0], device=self.device), t.as_tensor(tl[1].view(np.int64), device=self.device),
            t.as_tensor(tl[2] if tl[2].size else np.zeros(1, np.int32), device=self.device),
            t.as_tensor(mats, device=self.device))

  def bucket_unpack_tiles(self, i: int):
    if self._bucket_dev is None:
      self._bucket_dev = [self._tiles_dev(sel) for sel in self._bucket_tiles]
    return self._bucket_dev[i]

  @property
  def weights_dirty(self) -> bool:
    return self._weights_dirty

  @weights_dirty.setter
  def weights_dirty(self, v: bool):
    self._weights_dirty = v
    if v:
      self.dgrad_dirty = True

  def pack_weights(self, part: str = "all"):
    """flat parameter slab -> packed forward weights and biases ("enc" / "dec": one of the two pieces)."""
    tiles = {"all": self.pack_tiles, "enc": self.pack_tiles_enc, "dec": self.pack_tiles_dec,
             "enc_early": self.pack_tiles_enc_early, "enc_late": self.pack_tiles_enc_late}[part]
    if not _TIMING_ONLY_SKIP_COPIES & 1:
      self.be.copy_tiles(self.store.params, self.packed, tiles)
    self._operands(*{"all": ("enc_early", "enc_late", "dec"), "enc": ("enc_early", "enc_late"), "dec": ("dec",),
                     "enc_early": ("enc_early",), "enc_late": ("enc_late",)}[part])
    if part in ("all", "dec"):
      self._weights_dirty = False

  def pack_dgrad_weights(self):
    """flat parameter slab -> packed data-gradient weights (1 launch; backward only)."""
    if not _TIMING_ONLY_SKIP_COPIES & 2:
      self.be.copy_tiles(self.store.params, self.packed, self.pack_tiles_bwd)
    self._operands("bwd")
    self.dgrad_dirty = False

  # -------------------------------------------------------------------- plans
  def plan(self, batch: int, hw=IMAGE_HW) -> "Plan":
    """The plan (buffers + launch sequences) of one batch size and image size; (batch, 256, 256) is keyed by `batch` alone."""
    hw = (int(hw[0]), int(hw[1]))
    key = batch if hw == IMAGE_HW else (batch, hw[0], hw[1])
    p = self.plans.get(key)
    if p is None:
      p = Plan(self, batch, hw)
      self.plans[key] = p
    return p

  # ----------------------------------------------------------------- optimizer
  def adam_step(self, lr: float, eps: float, betas=(0.9, 0.999), grad_scale: float = 1.0):
    """torch.optim.Adam step on the whole parameter slab (state.py:65, pipeline.py:230)."""
    if self.adam_m is None:
      self.adam_m = t.zeros_like(self.store.params)
      self.adam_v = t.zeros_like(self.store.params)
    self.adam_t += 1
    self.be.adam_step(self.store.params, self.store.grads, self.adam_m, self.adam_v,
                      self.store.params.numel(), lr, betas[0], betas[1], eps, grad_scale, self.adam_t)
    self.weights_dirty = True

  def adam_step_graphable(self, lr: float, eps: float, betas=(0.9, 0.999), grad_scale: float = 1.0, launch: bool = True):
    """The same step split for HIP-graph replay: the scalars (incl. the bias corrections of this step count) go to
    a small device buffer with an ordinary launch, the update itself (`launch`) reads them from there and so can
    be a node of a captured graph."""
    if self.adam_m is None:
      self.adam_m = t.zeros_like(self.store.params)
      self.adam_v = t.zeros_like(self.store.params)
    if self.adam_hyper is None:
      self.adam_hyper = t.zeros(8, dtype=t.float32, device=self.device)
    self.adam_t += 1
    self.be.adam_set_hyper(self.adam_hyper, lr, betas[0], betas[1], eps, grad_scale, self.adam_t)
    if launch:
      self.adam_update_from_hyper()

  def adam_bucket_hook(self):
    """Bucket hook of a step without a gradient exchange (Plan.backward): Adam on the finished slice of the slab,
    on the side stream right behind the bucket's un-pack -- the update of everything but the last, small bucket
    (stem + encoder stages 2-3) then runs under the rest of backward instead of after it (0.16 ms for the 36 M
    parameters in one piece).  The scalars come from adam_step_graphable(launch=False)."""
    def hook(gslice: t.Tensor):
      lo, n = gslice.storage_offset(), gslice.numel()
      self.be.adam_step_hyper(self.store.params[lo:lo + n], gslice, self.adam_m[lo:lo + n], self.adam_v[lo:lo + n], n,
                              self.adam_hyper)
    return hook

  def adam_update_from_hyper(self):
    self.be.adam_step_hyper(self.store.params, self.store.grads, self.adam_m, self.adam_v,
                            self.store.params.numel(), self.adam_hyper)
    self.weights_dirty = True


def _no_exchange(grads: t.Tensor):
  """Bucket hook of a step without a gradient exchange."""


class Plan:
  """Buffers + the explicit forward / backward sequences for one batch size."""

  def __init__(self, eng: Engine, B: int, hw=IMAGE_HW):
    self.eng, self.B = eng, B
    H, W = check_image_hw(hw)
    self.hw = (H, W)
    self.stem_hw = (H // 2, W // 2)          # after the 7x7 / 2 stem (resnet50.py:122-124)
    # the stem's own kernels (csrc/stem_conv.hip: H even, W a multiple of 8, <= 4096 workgroup partial sums); CRN_STEM=0: the
    # generic engine on the space-to-depth view, as before round 5
    parts = int(eng.be.lib.crn_stem_conv_parts(B, H, W)) if hasattr(eng.be, "stem_conv_fwd") else 0
    self.stem_fast = os.environ.get("CRN_STEM", "1") != "0" and 0 < parts <= 4096
    self.dec_stats_fused = os.environ.get("CRN_DEC_STATS_FUSE", "1") != "0" and hasattr(eng.be, "conv_fwd_stats")
    self.pool_hw = (H // 4, W // 4)          # after the 3x3 / 2 max-pool (:128-131): stage 2 runs here
    self.be = eng.be
    self.generation = 0            # bumped by every forward: CoreNet's autograd node checks it in backward
    self.graphs = {}               # captured training steps (CoreNet.train_step): loss name -> CUDAGraph
    self.eager_steps = 0           # fused steps run eagerly on this plan (the first one also sizes every workspace)
    self.eval_graph = None         # captured eval-mode forward (CoreNet._forward_eval_graph)
    self.ray_side = os.environ.get("CRN_RAY_SIDE", "1") != "0"      # the ray scatters of the backward on the side stream
    self.eval_eager = 0
    self._views = {}
    dev = eng.device
    self.dev = dev
    f = lambda *shape: t.zeros(*shape, dtype=eng.dtype, device=dev)
    # static inputs of the captured training step (a replayed graph reads fixed addresses)
    self.in_image = t.zeros(B, 3, H, W, dtype=t.uint8, device=dev)
    self.in_v2s = f(B, 4, 4)
    self.in_off = f(B, 3)
    self.img = f(B, 3, H, W)
    self.y1 = f(B, 64, *self.stem_hw)
    self.gy1 = f(B, 64, *self.stem_hw); self.gy1b = f(B, 64, *self.stem_hw)
    self.p1 = f(B, 64, *self.pool_hw)
    self.p1_arg = t.zeros(B, 64, *self.pool_hw, dtype=t.int32, device=dev)
    # encoder block buffers
    self.blocks = []
    cin, (hi, wi) = 64, self.pool_hw
    stage_hw = {}
    for name, blocks, filt, stride in ENC_STAGES:
      for bl in blocks:
        st = stride if bl == "a" else 1
        ho, wo = (hi + st - 1) // st, (wi + st - 1) // st        # 1x1 stride-2 conv without padding: ceil (resnet50.py:94-97)
        blk = dict(prefix=f"encoder.{name}.{bl}.", cin=cin, f=filt, stride=st, hin=hi, win=wi, h=ho, w=wo,
                   down=(bl == "a"), final=(bl == blocks[-1]), stage=name)
        blk["ya"] = f(B, filt[0], ho, wo); blk["yb"] = f(B, filt[1], ho, wo); blk["yc"] = f(B, filt[2], ho, wo)
        if blk["down"]:
          blk["ys"] = f(B, filt[2], ho, wo)
        blk["out"] = f(B, filt[2], ho, wo)
        # gradients
        blk["gpre"] = f(B, filt[2], ho, wo); blk["gyc"] = f(B, filt[2], ho, wo)
        blk["gab"] = f(B, filt[1], ho, wo); blk["gyb"] = f(B, filt[1], ho, wo)
        blk["gaa"] = f(B, filt[0], ho, wo); blk["gya"] = f(B, filt[0], ho, wo)
        if blk["down"]:
          blk["gys"] = f(B, filt[2], ho, wo)
          blk["gin"] = f(B, cin, hi, wi)
          if st == 2:
            # the stride-2 1x1 convs (op_a and the shortcut, resnet50.py:94-97) read a compacted copy of the
            # sub-sampled block input, and their data gradients meet in a compact buffer before they are
            # scattered back: plain tensors -> the pointwise kernel instead of strided views
            blk["xs"] = f(B, cin, ho, wo); blk["gxs"] = f(B, cin, ho, wo)
        if st == 2 and self.blocks:
          self.blocks[-1]["xs_next"] = blk          # the block in front writes this block's compacted input with its tail
        self.blocks.append(blk)
        cin, hi, wi = filt[2], ho, wo
      stage_hw[name] = (hi, wi)
    self.stage_hw = stage_hw
    # stage-final pre-ReLU features + the 3 offset channels (Q6), and their grads
    self.feat = {"stage2": f(B, 256 + 3, *stage_hw["stage2"]), "stage3": f(B, 512 + 3, *stage_hw["stage3"]),
                 "stage4": f(B, 1024 + 3, *stage_hw["stage4"]), "stage5": f(B, 2048 + 3, *stage_hw["stage5"])}
    self.gfeat = {k: t.zeros_like(v) for k, v in self.feat.items()}
    self.avg = f(B, 2048); self.gavg = f(B, 2048)
    L = eng.latent
    self.z0 = f(B, L + 3); self.gz0 = f(B, L + 3); self.gv0 = f(B, L + 3)
    # decoder: concat buffers U_k (input of stage k), mid tensors W_k
    C = eng.num_classes
    sk = eng.skip_ch
    self.dec = {2: dict(cin=256, cmid=256, cout=128, r=4), 3: dict(cin=128 + sk[2], cmid=128, cout=64, r=8),
                4: dict(cin=64 + sk[3], cmid=64, cout=32, r=16), 5: dict(cin=32 + sk[4], cmid=32, cout=16, r=32),
                6: dict(cin=16 + sk[5], cmid=16, cout=C, r=64)}
    for k, d in self.dec.items():
      r = d["r"]
      d["u"] = f(B, d["cin"], r, r, r); d["gu"] = f(B, d["cin"], r, r, r)
      d["w"] = f(B, d["cmid"], r, r, r); d["gw"] = f(B, d["cmid"], r, r, r)
      d["gv1"] = f(B, d["cin"], r, r, r); d["gv2"] = f(B, d["cmid"], r, r, r)
    self.logits = f(B, C, 128, 128, 128)
    self.glogits = f(B, C, 128, 128, 128)
    self.skip_src = {2: "stage5", 3: "stage4", 4: "stage3", 5: "stage2"}
    self.skip_hw = {k: stage_hw[self.skip_src[k]] for k in self.skip_src}      # (h, w) of the 2-D map each skip samples
    # compressed skip maps are channel-last [B][h][w][C]: the ray-sample gather then fetches four channels of a
    # pixel per load (crn_ray_sample_fwd); their gradients stay channel-major (scatter-add per channel plane)
    self.smap = {k: f(B, self.skip_hw[k][0], self.skip_hw[k][1], sk[k]) for k in sk}
    # the saved index tensors of the ray-sample backward (autograd keeps them in the reference: ray_traced_skip_connection.py:
    # 118-135): one uint16 per voxel of the 8^3 ... 64^3 skip grids, written by the training-mode gather, read by the scatter
    self.ray_idx = ({k: t.zeros(B, (2 * self.dec[k]["r"]) ** 3, dtype=t.int16, device=dev) for k in sk}
                    if hasattr(self.be, "ray_sample_fwd_idx") and os.environ.get("CRN_RAY_IDX", "1") != "0" and
                    max(h * w for h, w in self.skip_hw.values()) < 65535 else None)
    gs_n = {k: B * sk[k] * self.skip_hw[k][0] * self.skip_hw[k][1] for k in sk}
    self.gsmap_slab = f(sum(gs_n.values()))
    self.gsmap, o = {}, 0
    for k in sk:
      self.gsmap[k] = self.gsmap_slab[o:o + gs_n[k]].view(B, sk[k], *self.skip_hw[k]); o += gs_n[k]
    self.layer_mats = f(4, B, 16)
    self._layer_scale_list = [128.0 / (2 * self.dec[k]["r"]) for k in (2, 3, 4, 5)]
    self.layer_scales = t.tensor([[128.0 / (2 * self.dec[k]["r"])] * 3 + [1.0] for k in (2, 3, 4, 5)],
                                 dtype=eng.dtype, device=dev).reshape(4, 1, 1, 4)
    self.offset = f(B, 3)
    self.loss = f(1)
    self.gt = t.zeros(B, 128, 128, 128, dtype=t.int32, device=dev)
    # the side stream, the events of the weight packs and their "pending" flags belong to the ENGINE: packed weights
    # and the packed gradient slab are shared by the plans of all batch sizes (forward(B=4), forward+backward(B=2),
    # backward(B=4) must see one consistent state)
    self.side = eng.side_stream()
    use_side = self.side is not None
    self._side_ev, self._side_i = [], 0
    self._side_done = t.cuda.Event() if use_side else None
    self._ximg_ev = t.cuda.Event() if use_side else None     # hand-over of the logits layer's operand image (forward_decoder)
    self._ct_ximg = self._ct_ximg_buf = None
    # Every hand-over to the side stream costs the data-gradient chain an event record (a marker packet between two dependent
    # launches: ~3.5 us).  The encoder's weight gradients are therefore handed over one bottleneck at a time (3-4 convolutions behind
    # ONE event) and a gradient bucket's un-pack rides on the hand-over in front of it: 7.58 -> 7.44 ms per step
    # (profiles/r05_wgrad_handover.txt; per conv 0, per 1 / 2 / 3 / 6 blocks, per bucket 99; CRN_WG_BATCH_DEC=1: decoder stages too)
    self.wg_batch = int(os.environ.get("CRN_WG_BATCH", "1")) if use_side else 0      # bottlenecks per hand-over (0: per conv)
    self.wg_batch_dec = os.environ.get("CRN_WG_BATCH_DEC", "0") == "1"
    self._handed_over = False
    self._g_compact = None
    self._wg_pending = []
    self._wg_blocks = 0
    # the ray-traced skip path (offset channels -> 1x1 compress -> ray sample; backward: scatter -> compress gradients)
    # hangs off the encoder's stage outputs and joins the main chain only at the consuming decoder stage (backward: at the
    # start of the encoder's backward pass): it runs on the side stream beside the encoder / decoder chain
    self.async_skip = use_side and os.environ.get("CRN_ASYNC_SKIP", "1") != "0"
    self._skip_ev = {k: (t.cuda.Event(), t.cuda.Event()) for k in (2, 3, 4, 5)} if use_side else None
    self._skip_bwd_ev = {k: t.cuda.Event() for k in (2, 3, 4, 5)} if use_side else None
    self._skip_bwd_done = t.cuda.Event() if use_side else None
    self._skip_async_live = False
    self._bucket_ev = [t.cuda.Event() for _ in GRAD_BUCKET_LABELS] if use_side else None

  # ------------------------------------------------------------------ cached views
  def _cached(self, key, fn):
    v = self._views.get(key)
    if v is None:
      v = fn()
      self._views[key] = v
    return v

  def vw(self, x: t.Tensor) -> V.View:
    return self._cached(("v", x.data_ptr(), tuple(x.shape), tuple(x.stride())), lambda: V.view_of(x))

  def s2d(self, x: t.Tensor, c1: int, r) -> V.View:
    """space-to-depth / pixel-shuffle view of channels [0, c1) of x; the 2x2x2 views of the transposed
    convolutions are parity major (conv_geometry.convt_fwd), the stem's 1x2x2 view is channel major."""
    pm = tuple(r) == (2, 2, 2)
    return self._cached(("s2d", x.data_ptr(), tuple(x.shape), tuple(x.stride()), c1, r),
                        lambda: V.space_to_depth_view(self.vw(x).channels(0, c1), r, parity_major=pm))

  def flat(self, x: t.Tensor) -> V.View:
    return self._cached(("flat", x.data_ptr(), tuple(x.shape)), lambda: V.flat_channel_view(self.vw(x)))

  def strided(self, x: t.Tensor, step) -> V.View:
    return self._cached(("str", x.data_ptr(), tuple(x.shape), step), lambda: V.strided_view(self.vw(x), step))

  # ------------------------------------------------------------------ helpers
  probes = None     # bench.py: {name: [(start_event, end_event), ...]} around selected launches

  def _probe(self, name: str, fn):
    if self.probes is None or name not in self.probes:
      return fn()
    a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record()
    self.probes[name].append((a, b))

  def _stats(self, bn: BN, x: t.Tensor, S: int, sB: int, pre_relu: bool, training: bool):
    if not training:
      return            # eval mode: forward_encoder filled every scale / shift already (Engine.bn_eval_affine)
    with _lib.roctx_range("bn_stats C%d S%d" % (bn.C, S)):
      self._bn_stats(bn, x, S, sB, pre_relu, training)

  def _bn_stats(self, bn: BN, x: t.Tensor, S: int, sB: int, pre_relu: bool, training: bool):
    self.be.bn_stats(x, self.B, bn.C, S, sB, pre_relu, bn.gamma, bn.beta, bn.rmean, bn.rvar, bn.nbt,
                     BN_EPS, BN_MOMENTUM, training, bn.scale, bn.shift, bn.saved)

  trace = None      # tools/layer_times.py: list of (label, start_event, end_event) for every conv launch

  def _timed(self, label, fn):
    if _lib.ROCTX:                 # CRN_ROCTX=1: a rocprofv3 marker range per layer and direction
      with _lib.roctx_range(label):
        return fn()
    if self.trace is None:
      return fn()
    a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    a.record(); fn(); b.record()
    self.trace.append((label, a, b))

  def _math(self, cv: Conv, direction: str) -> str:
    if cv.wop_kind == "e2d" and (direction == "fwd" and cv.wop_f is not None or
                                 direction == "dgrad" and cv.wop_d is not None):
      return "e2d"
    return "bf16x3" if self.eng.decoder_math == "bf16x3" and (cv.name, direction) in BF16X3_LAUNCHES else "fp32"

  conv_positions = None     # tools/layer_times.py: {layer name: logical output positions per sample} when tracing

  @staticmethod
  def _e2d_shape_ok(v: V.View, window) -> bool:
    """The shapes crn_conv2d_bf3 covers (csrc/conv_e2d.hip): channel planes read in 16-byte pieces, 3 x 3 layers in tiles of 16 x 4 or
    8 x 8 positions, 1 x 1 layers in runs of 64.  Every stage map of a 256 x 256 image qualifies; the 14 x 14 / 7 x 7 maps of a
    224 x 224 image do not and take the fp32 engine."""
    S = v.D * v.H * v.W
    if S % 4:
      return False
    if tuple(window) == (1, 3, 3):
      tw = 16 if v.W % 16 == 0 else 8
      return v.D == 1 and v.W % tw == 0 and v.H % (64 // tw) == 0
    return S % 64 == 0

  @staticmethod
  def _wg2d_shape_ok(v: V.View, window) -> bool:
    """crn_conv_wgrad_2d_bf3: K = positions in steps of 32, 3 x 3 layers shift aligned 8-position groups."""
    return (v.D * v.H * v.W) % 32 == 0 and (tuple(window) == (1, 1, 1) or v.W % 8 == 0)

  def _conv(self, cv: Conv, x: V.View, tr, y: V.View, accumulate=False):
    g = cv.fwd
    if self.trace is not None and self.conv_positions is not None:
      self.conv_positions[cv.name] = y.D * y.H * y.W
    if cv.wop_f is not None and cv.wop_kind == "e2d" and self._e2d_shape_ok(y, g.window):
      self._timed("fwd   " + cv.name, lambda: self.be.conv2d_bf3(
          x, tr, cv.wop_f, g.npad, cv.bias, 0, y, g.window, g.pad_lo, accumulate))
      return
    self._timed("fwd   " + cv.name, lambda: self.be.conv_fwd(
        x, tr, cv.wf, g.npad, cv.bias, 0, y, g.window, g.pad_lo, 0, accumulate, boxes=(g.n_boxes, g.c_boxes),
        math=self._math(cv, "fwd"), wslab=cv.wop_f if cv.wop_kind == "slab" else None))

  def _dgrad(self, cv: Conv, dy: V.View, dx: V.View, accumulate=False):
    g = cv.dgrad
    if cv.wop_d is not None and cv.wop_kind == "e2d" and self._e2d_shape_ok(dx, g.window):
      self._timed("dgrad " + cv.name, lambda: self.be.conv2d_bf3(
          dy, None, cv.wop_d, g.npad, None, 0, dx, g.window, g.pad_lo, accumulate))
      return
    self._timed("dgrad " + cv.name, lambda: self.be.conv_fwd(
        dy, None, cv.wd, g.npad, None, 0, dx, g.window, g.pad_lo, 0, accumulate, boxes=(g.n_boxes, g.c_boxes),
        math=self._math(cv, "dgrad"), wslab=cv.wop_d if cv.wop_kind == "slab" else None))

  def _dgrad_bn_bwd(self, cv: Conv, dy: V.View, g: t.Tensor, x: t.Tensor, Cn: int, S: int, b, dx: t.Tensor, cprev: Conv) -> bool:
    """Data gradient g of decoder conv `cv` + backward of the BatchRenorm `b` in front of it (input x, output gradient g,
    input gradient dx; dsum = the bias gradient of the conv `cprev` that produced x).  True: the norm's backward is done
    (its sums came out of the conv launch, crn_conv_fwd_bf3_slabs_bnbwd); False: g is written, the caller runs bn_bwd."""
    eng, be = self.eng, self.be
    gvw = self.vw(g)
    fusable = (eng.bn_bwd_fuse and hasattr(be, "conv_dgrad_bn_bwd") and cv.wop_kind == "slab" and cv.wop_d is not None
               and self._math(cv, "dgrad") == "bf16x3" and cv.name not in eng.bn_bwd_fuse_skip)
    if not fusable:
      if eng.defer_reduce:
        be.splitk_defer()                        # g is read next by the norm's backward, which adds up the splits
      self._dgrad(cv, dy, gvw)
      return False
    gd = cv.dgrad
    done = [False]
    if eng.defer_reduce:
      be.splitk_defer()                          # (a launch that splits cannot fuse: its sum goes to the norm's own first pass)
    def run():
      done[0] = be.conv_dgrad_bn_bwd(dy, cv.wop_d, gd.npad, gvw, g, gd.window, gd.pad_lo, (gd.n_boxes, gd.c_boxes),
                                     x, Cn * S, self.B, Cn, S, True, b.gamma, b.scale, b.shift, b.saved, dx, Cn * S,
                                     b.dgamma, b.dbeta, dsum=cprev.dbias, ndsum=cprev.n_ref)
    self._timed("dgrad " + cv.name, run)
    return done[0]

  def _wgrad(self, cv: Conv, x: V.View, tr, dy: V.View):
    """Weight gradient of one conv.  On the GPU it goes to a second HIP stream: it only reads
    (saved activation, dy) and writes its own slice of the packed gradient slab, so it can run
    beside the data-gradient chain; most layers below 32^3 / 64^2 cannot fill 256 CUs alone."""
    g = cv.fwd
    math = self._math(cv, "wgrad")
    if self.stem_fast and cv.name == "encoder.stage1.conv.":
      math = "stem"               # (falls back to the generic engine by itself where it does not apply)
    if (self.eng.encoder_e2d and self.eng.wgrad_2d and g.window in ((1, 1, 1), (1, 3, 3)) and self._wg2d_shape_ok(dy, g.window)
        and (cv.name.startswith("encoder.stage") and not cv.name.startswith("encoder.stage1")
             or cv.name.startswith("decoder.rt_skip"))):
      math = "bf16x3_2d"          # both operands straight from HBM, K = positions (csrc/conv_e2d.hip)
    if cv.ct_kind == "par" and math == "bf16x3":
      math = "ct_par"             # the logits layer with > 8 classes: parity-walk kernel (csrc/convt_par.hip; generic engine in deterministic mode)
    if self.side is None or self.trace is not None:
      self._timed("wgrad " + cv.name, lambda: self.be.conv_wgrad(
          x, tr, dy, cv.gwf, g.npad, g.window, g.pad_lo, False, boxes=(g.n_boxes, g.c_boxes), math=math, **self._ximg_kw(math)))
      return
    if self.wg_batch and (cv.name.startswith("encoder.") or self.wg_batch_dec):
      # the encoder's small weight gradients are handed to the side stream one bottleneck at a time (_flush_wgrads): ONE event on
      # the data-gradient chain per block instead of one per convolution
      self._wg_pending.append((cv, x, tr, dy, math, self._ximg_kw(math)))
      return
    if self._handed_over:
      # the side stream already waits for an event recorded after dy became final (the skip path's hand-over, just before this
      # call, with nothing launched on the main stream in between): no second marker on the data-gradient chain
      self._handed_over = False
      self._side_i = max(self._side_i, 1)
      with t.cuda.stream(self.side), _lib.pinned_stream(self.side), _lib.roctx_range("wgrad " + cv.name):
        self.be.conv_wgrad(x, tr, dy, cv.gwf, g.npad, g.window, g.pad_lo, False, boxes=(g.n_boxes, g.c_boxes), math=math,
                           **self._ximg_kw(math))
      return
    if self._side_i == len(self._side_ev):
      self._side_ev.append(t.cuda.Event())
    ev = self._side_ev[self._side_i]; self._side_i += 1
    ev.record()                                   # dy (and the zeroed slab) are ready on the main stream
    with t.cuda.stream(self.side), _lib.pinned_stream(self.side), _lib.roctx_range("wgrad " + cv.name):
      self.side.wait_event(ev)
      self.be.conv_wgrad(x, tr, dy, cv.gwf, g.npad, g.window, g.pad_lo, False, boxes=(g.n_boxes, g.c_boxes), math=math,
                         **self._ximg_kw(math))

  def _ximg_kw(self, math: str) -> dict:
    """The operand image the forward pass left for the parity-walk weight gradient (forward_decoder), consumed once."""
    if math != "ct_par" or self._ct_ximg is None:
      return {}
    img, self._ct_ximg = self._ct_ximg, None
    return {"ximg": img}

  def _flush_wgrads(self, then=None):
    """Launches the weight gradients collected since the last flush on the side stream, behind ONE event of the main stream;
    `then` (a gradient bucket's un-pack + hook) runs behind them in the same hand-over."""
    if not self._wg_pending and then is None:
      return
    if self._side_i == len(self._side_ev):
      self._side_ev.append(t.cuda.Event())
    ev = self._side_ev[self._side_i]; self._side_i += 1
    ev.record()
    with t.cuda.stream(self.side), _lib.pinned_stream(self.side):
      self.side.wait_event(ev)
      for cv, x, tr, dy, math, kw in self._wg_pending:
        g = cv.fwd
        with _lib.roctx_range("wgrad " + cv.name):
          self.be.conv_wgrad(x, tr, dy, cv.gwf, g.npad, g.window, g.pad_lo, False, boxes=(g.n_boxes, g.c_boxes), math=math, **kw)
      if then is not None:
        then()
    self._wg_pending = []
    self._wg_blocks = 0

  def _join_side(self):
    """Main stream waits for every weight gradient issued on the side stream."""
    self._flush_wgrads()
    if self.side is not None and self._side_i:
      self._side_done.record(self.side)
      t.cuda.current_stream().wait_event(self._side_done)
      self._side_i = 0

  def _bias_grad(self, cv: Conv, dy: t.Tensor, S: int, sB: int):
    self.be.bias_grad(dy, self.B, cv.n_ref, S, sB, cv.dbias)

  # ------------------------------------------------------------------ forward
  def forward(self, image_u8: t.Tensor, v2s: t.Tensor, offset: t.Tensor, training: bool) -> t.Tensor:
    self._skip_async_live = bool(training and self.async_skip and self.trace is None)
    if self._skip_async_live:
      self._decoder_inputs(v2s, offset)            # the skip path starts inside the encoder: it needs them now
    self.forward_encoder(image_u8, training)
    try:
      return self.forward_decoder(v2s, offset, training)
    finally:
      self._skip_async_live = False

  def _decoder_inputs(self, v2s: t.Tensor, offset: t.Tensor):
    """Sampling offset and the layer matrices v2s @ scale(128 / r) of the four skip grids (reconstruction_decoder.py:111-116)."""
    B = self.B
    if (hasattr(self.be, "decoder_inputs") and v2s.dtype == self.layer_mats.dtype and offset.dtype == self.offset.dtype
        and v2s.is_contiguous() and offset.is_contiguous() and v2s.device == self.layer_mats.device):
      self.be.decoder_inputs(v2s, offset, self._layer_scale_list, self.layer_mats, self.offset)      # one launch
      return
    self.offset.copy_(offset)
    v = v2s.to(self.layer_mats.dtype).reshape(1, B, 4, 4)
    self.layer_mats.copy_((v * self.layer_scales).reshape(4, B, 16))    # exact: column scaling

  def _skip_fwd(self, k: int):
    """Skip connection into decoder stage k+1 (reconstruction_decoder.py:97-117): offset channels, 1x1 compress, ray sample
    straight into the skip channels of the stage's concat buffer."""
    eng, be, B = self.eng, self.be, self.B
    d = self.dec[k]
    out = self.dec[k + 1]["u"]
    ft = self.feat[self.skip_src[k]]
    (hh, ww), ro = self.skip_hw[k], 2 * d["r"]
    be.fill_offset_channels(ft, B, ft.stride(0), ft.shape[2] * ft.shape[3], ft.shape[1] - 3, self.offset)
    self._conv(eng.convs[f"decoder.rt_skip_{k}.compress_channels."], self.vw(ft), None,
               self.vw(self.smap[k].permute(0, 3, 1, 2)))
    if self.ray_idx is not None and self.training:
      self._probe(f"ray_sample_fwd_{ro}", lambda: be.ray_sample_fwd_idx(
          self.smap[k], self.smap[k].stride(0), B, eng.skip_ch[k], hh, ww, self.layer_mats[k - 2],
          self.offset, out[:, d["cout"]:], out.stride(0), ro, ro, ro, self.ray_idx[k], map_sC=1, map_sP=eng.skip_ch[k]))
    else:
      self._probe(f"ray_sample_fwd_{ro}", lambda: be.ray_sample_fwd(
          self.smap[k], self.smap[k].stride(0), B, eng.skip_ch[k], hh, ww, self.layer_mats[k - 2],
          self.offset, out[:, d["cout"]:], out.stride(0), ro, ro, ro, map_sC=1, map_sP=eng.skip_ch[k]))

  def _skip_fwd_async(self, stage: str):
    """The stage's pre-ReLU feature map is final on the main stream: its skip path goes to the side stream."""
    k = {v: kk for kk, v in self.skip_src.items()}[stage]
    ready, done = self._skip_ev[k]
    ready.record()
    with t.cuda.stream(self.side), _lib.pinned_stream(self.side):
      self.side.wait_event(ready)
      self._skip_fwd(k)
      done.record(self.side)

  def forward_encoder(self, image_u8: t.Tensor, training: bool):
    """ResNet-50 features + global average (resnet50.py:176-186).  In eval mode the result does not depend
    on the sampling offset, so multi-offset inference (super_resolution.py:123-125) runs it once."""
    if tuple(image_u8.shape) != tuple(self.in_image.shape):       # (backend.preprocess writes B*3*H*W floats into self.img)
      raise ValueError(f"plan for images {tuple(self.in_image.shape)}, got {tuple(image_u8.shape)}")
    eng, be, B = self.eng, self.be, self.B
    self.generation += 1
    # (_dgrad_pack_pending / _gpacked_zeroed stay set until a backward consumes them: a second forward before
    # the backward must not forget that the side stream still owes it the data-gradient weights)
    if training and self.side is not None and (eng.dgrad_dirty or eng.weights_dirty):
      # work that nothing on the main stream needs yet goes to the side stream, under the encoder: the decoder's
      # forward weights (needed ~1.4 ms later), the data-gradient weights and the zeroing of the packed gradient
      # slab (needed by backward)
      dec_later = eng.weights_dirty
      if dec_later:
        eng.pack_weights("enc_early")             # stem + stage2 (0.2 M weights): needed at once
      eng._pack_ev.record()                      # the parameters are final on the main stream
      with t.cuda.stream(self.side), _lib.pinned_stream(self.side):
        self.side.wait_event(eng._pack_ev)
        if dec_later:
          eng.pack_weights("enc_late")            # stage3-5 (23 M): needed ~0.5 ms into the encoder
          eng._enc_late_packed.record(self.side)
          eng.pack_weights("dec")
          eng._dec_packed.record(self.side)
        if eng.dgrad_dirty:
          eng.pack_dgrad_weights()
        be.zero(eng.gpacked)
        eng._dgrad_packed.record(self.side)
      eng._dgrad_pack_pending = True
      eng._dec_pack_pending = dec_later
      eng._enc_late_pending = dec_later
      eng._gpacked_zeroed = True
    elif eng.weights_dirty:
      eng.pack_weights()
    cv, bn = eng.convs, eng.bns
    self.training = training
    if not training:
      eng.bn_eval_affine()
    be.preprocess(image_u8, self.img)
    # stem (resnet50.py:122-131)
    c1 = cv["encoder.stage1.conv."]
    b1 = bn["encoder.stage1_part2.bn."]
    H1, W1 = self.stem_hw
    if self.stem_fast:
      # the stem on its own kernel (csrc/stem_conv.hip); in training the partial sums of its norm come out of the same launch
      parts = [0]
      if self.trace is not None and self.conv_positions is not None:
        self.conv_positions[c1.name] = H1 * W1
      def stem():
        parts[0] = be.stem_conv_fwd(self.img, c1.wf, c1.bias, self.y1, training)
      self._timed("fwd   " + c1.name, stem)
      if training:
        be.bn_finalize(parts[0], 64, B * H1 * W1, b1.gamma, b1.beta, b1.rmean, b1.rvar, b1.nbt, BN_EPS, BN_MOMENTUM,
                       b1.scale, b1.shift, b1.saved)
    else:
      self._conv(c1, self.s2d(self.img, 3, (1, 2, 2)), None, self.vw(self.y1))
      self._stats(b1, self.y1, H1 * W1, 64 * H1 * W1, False, training)
    be.maxpool_fwd(self.y1, b1.scale, b1.shift, B, 64, H1, W1, self.p1, self.p1_arg)
    cur = self.p1
    for blk in self.blocks:
      if eng._enc_late_pending and blk["stage"] != "stage2":
        t.cuda.current_stream().wait_event(eng._enc_late_packed)
        eng._enc_late_pending = False
      cur = self._block_fwd(blk, cur, training)
      if blk["final"] and self._skip_async_live:
        self._skip_fwd_async(blk["stage"])
    f5 = self.feat["stage5"]
    S5 = f5.shape[2] * f5.shape[3]
    be.relu_mean_fwd(f5, B, 2048, S5, f5.stride(0), self.avg)

  def forward_decoder(self, v2s: t.Tensor, offset: t.Tensor, training: bool) -> t.Tensor:
    """Offset channels, skip compression + ray sampling, 3D decoder (reconstruction_decoder.py:97-151)."""
    eng, be, B = self.eng, self.be, self.B
    cv, bn = eng.convs, eng.bns
    self.generation += 1
    if eng._dec_pack_pending:
      t.cuda.current_stream().wait_event(eng._dec_packed)
      eng._dec_pack_pending = False
    skip_async = self._skip_async_live
    if not skip_async:
      self._decoder_inputs(v2s, offset)
    # decoder (reconstruction_decoder.py:136-151)
    L = eng.latent
    s = eng.store
    be.linear_fwd(self.avg, s.view("decoder.stage_0.weight"), s.view("decoder.stage_0.bias"),
                  B, 2048, L, self.z0, L + 3)
    be.fill_offset_channels(self.z0, B, L + 3, 1, L, self.offset)
    b = bn["decoder.stage_1.b1."]
    self._stats(b, self.z0, 1, L + 3, True, training)
    zv = self.vw(self.z0.view(B, L + 3, 1, 1, 1))
    u2 = self.dec[2]["u"]
    self._conv(cv["decoder.stage_1.t1."], zv, Transform(b.scale, b.shift, pre_relu=True),
               self.flat(u2))
    for k in range(2, 7):
      d = self.dec[k]
      r, S = d["r"], d["r"] ** 3
      p = f"decoder.stage_{k}."
      b1_, b2_ = bn[p + "b1."], bn[p + "b2."]
      if skip_async and k == 3:
        # the skip paths run on the side stream in the order of the encoder's stages (64^3 first, 8^3 last): the event behind the
        # LAST one -- the 8^3 skip into this stage -- covers all four; one marker on the decoder's chain instead of four
        t.cuda.current_stream().wait_event(self._skip_ev[2][1])
      self._stats(b1_, d["u"], S, d["cin"] * S, True, training)
      if training and eng.defer_reduce:
        be.splitk_defer()                      # (stages 2-4 split K: the statistics below add the partial sums up)
      cc = cv[p + "c1."]
      parts = [0]
      if (training and self.dec_stats_fused and cc.wop_kind == "slab" and cc.wop_f is not None and self._math(cc, "fwd") == "bf16x3"
          and self.trace is None):
        # the partial sums of the norm behind c1 come out of c1's launch where it does not split its reduction (stages 5-6 at the
        # bench batch): crn_conv_fwd_bf3_slabs_stats + crn_batch_renorm_finalize instead of a statistics pass over w
        gq = cc.fwd
        def c1_stats():
          parts[0] = be.conv_fwd_stats(self.vw(d["u"]), Transform(b1_.scale, b1_.shift, pre_relu=True), cc.wop_f, gq.npad, cc.bias, 0,
                                       self.vw(d["w"]), gq.window, gq.pad_lo, (gq.n_boxes, gq.c_boxes), d["cmid"], True)
        self._probe(f"conv3d_stage{k}_c1_fwd", lambda: self._timed("fwd   " + cc.name, c1_stats))
      else:
        self._probe(f"conv3d_stage{k}_c1_fwd", lambda: self._conv(
            cc, self.vw(d["u"]), Transform(b1_.scale, b1_.shift, pre_relu=True), self.vw(d["w"])))
      if parts[0] > 0:
        be.bn_finalize(parts[0], d["cmid"], B * S, b2_.gamma, b2_.beta, b2_.rmean, b2_.rvar, b2_.nbt, BN_EPS, BN_MOMENTUM,
                       b2_.scale, b2_.shift, b2_.saved)
      else:
        self._stats(b2_, d["w"], S, d["cmid"] * S, True, training)
      out = self.dec[k + 1]["u"] if k < 6 else self.logits
      ct = cv[p + "t1."]
      self._ct_ximg = None
      if (training and ct.ct_kind == "par" and self.side is not None and self.trace is None and hasattr(be, "convt_ximage")
          and self._math(ct, "wgrad") == "bf16x3"):
        # the parity-walk weight gradient of the logits layer reads T(w) as an operand image (crn_convt_s2k7_ximage): made here on the
        # side stream, which is idle under the decoder's forward pass -- the norm's scale / shift are final, backward finds it ready
        self._ximg_ev.record()
        with t.cuda.stream(self.side), _lib.pinned_stream(self.side):
          self.side.wait_event(self._ximg_ev)
          self._ct_ximg_buf = be.convt_ximage(d["w"], Transform(b2_.scale, b2_.shift, pre_relu=True), self._ct_ximg_buf)   # (this plan's own buffer)
          self._ct_ximg = self._ct_ximg_buf
      if ct.ct_f is not None and self._math(ct, "fwd") == "bf16x3":
        # the logits layer with > 8 classes: parity-walk kernel (csrc/convt_par.hip)
        if self.trace is not None and self.conv_positions is not None:
          self.conv_positions[ct.name] = S
        self._probe(f"conv3d_stage{k}_t1_fwd", lambda: self._timed("fwd   " + ct.name, lambda: be.convt_par_fwd(
            d["w"], Transform(b2_.scale, b2_.shift, pre_relu=True), ct.ct_f, ct.bias, out, d["cout"],
            host_table=eng.ct_tables["dec"][2], resident=ct.ct_kind == "res")))
      else:
        ov = self.s2d(out, d["cout"], (2, 2, 2))
        self._probe(f"conv3d_stage{k}_t1_fwd", lambda: self._conv(ct, self.vw(d["w"]), Transform(b2_.scale, b2_.shift, pre_relu=True), ov))
      if k < 6 and not skip_async:
        self._skip_fwd(k)
    if training:
      be.add_i64(eng.store.nbt, eng.store.nbt.numel(), 1)     # batch_renorm.py:57
    return self.logits

  def _block_fwd(self, blk, cur: t.Tensor, training: bool) -> t.Tensor:
    eng, be, B = self.eng, self.be, self.B
    cv, bn = eng.convs, eng.bns
    p = blk["prefix"]
    h = blk["h"]; S = h * blk["w"]
    f1, f2, f3 = blk["f"]
    if blk["stride"] == 2:
      if not blk.pop("xs_filled", False):            # (filled by the tail launch of the block in front: crn_batch_renorm_stats_tail)
        be.stride2_gather(cur, blk["xs"])
      xin = self.vw(blk["xs"])
    else:
      xin = self.vw(cur)
    ba, bb, bc = bn[p + "op_a.bn."], bn[p + "op_b.bn."], bn[p + "op_c.bn."]
    # (training: every conv is followed directly by the statistics of its norm, which can add up the partial sums of
    # a split-K conv itself: crn_splitk_defer)
    defer = be.splitk_defer if training and self.eng.defer_reduce else (lambda: None)
    defer()
    self._conv(cv[p + "op_a.conv."], xin, None, self.vw(blk["ya"]))
    self._stats(ba, blk["ya"], S, f1 * S, False, training)
    defer()
    self._conv(cv[p + "op_b.conv."], self.vw(blk["ya"]), Transform(ba.scale, ba.shift, post_relu=True),
               self.vw(blk["yb"]))
    self._stats(bb, blk["yb"], S, f2 * S, False, training)
    if blk["final"]:
      pre, sB_pre = self.feat[blk["stage"]], self.feat[blk["stage"]].stride(0)
    else:
      pre, sB_pre = None, 0
    # training on the HIP backend: the statistics of op_c's norm and the block tail (norm + shortcut + ReLU) are one
    # launch (crn_batch_renorm_stats_tail) -- so the shortcut branch of a down-sampling block goes before op_c
    fused_tail = training and hasattr(be, "bn_stats_tail") and self.eng.fuse_tail
    res, rsc, rsh = cur, None, None
    def shortcut():
      bs = bn[p + "shortcut.bn."]
      defer()
      self._conv(cv[p + "shortcut.conv."], xin, None, self.vw(blk["ys"]))
      self._stats(bs, blk["ys"], S, f3 * S, False, training)
      return blk["ys"], bs.scale, bs.shift
    if blk["down"] and fused_tail:
      res, rsc, rsh = shortcut()
    defer()
    self._conv(cv[p + "op_c.conv."], self.vw(blk["yb"]), Transform(bb.scale, bb.shift, post_relu=True),
               self.vw(blk["yc"]))
    if fused_tail:
      nxt = blk.get("xs_next") if os.environ.get("CRN_FUSE_GATHER", "1") != "0" else None
      with _lib.roctx_range("bn_stats_tail C%d S%d" % (bc.C, S)):
        be.bn_stats_tail(blk["yc"], B, f3, S, f3 * S, bc.gamma, bc.beta, bc.rmean, bc.rvar, bc.nbt, BN_EPS, BN_MOMENTUM,
                         True, bc.scale, bc.shift, bc.saved, res, rsc, rsh, f3 * S, pre, sB_pre, blk["out"], f3 * S, True,
                         y2=nxt["xs"] if nxt is not None else None, W=blk["w"])
      if nxt is not None:
        nxt["xs_filled"] = True
    else:
      self._stats(bc, blk["yc"], S, f3 * S, False, training)
      if blk["down"]:
        res, rsc, rsh = shortcut()
      be.affine_add_relu(blk["yc"], bc.scale, bc.shift, res, rsc, rsh, B, f3, S,
                         f3 * S, f3 * S, pre, sB_pre, blk["out"], f3 * S, True)
    blk["in"] = cur
    return blk["out"]

  # ------------------------------------------------------------------ backward
  def _grads_ready(self, label: str, hook):
    """Bucket `label` of the grad slab is complete once everything issued so far has run: unpack its conv
    weight gradients and hand the slice to `hook` (an async all-reduce).  Both go to the side stream behind
    the bucket's weight gradients, so the data-gradient chain on the main stream never waits for them."""
    if hook is None:
      self._flush_wgrads()
      return
    eng = self.eng
    i = [b[0] for b in eng.grad_buckets].index(label)
    _, lo, hi = eng.grad_buckets[i]
    tiles = eng.bucket_unpack_tiles(i)
    def run():
      if (tiles[0].numel() or tiles[3].numel()) and not _TIMING_ONLY_SKIP_COPIES & 4:
        self.be.copy_tiles(eng.gpacked, eng.store.grads, tiles, reverse=True)
      hook(eng.store.grads[lo:hi])
    if self.side is None or self.trace is not None:
      return run()
    if self.wg_batch:
      return self._flush_wgrads(then=run)          # (its event also covers the bucket's bias / norm gradients on the main stream)
    self._bucket_ev[i].record()                  # bias / norm gradients of the bucket are written on the main stream
    with t.cuda.stream(self.side), _lib.pinned_stream(self.side):
      self.side.wait_event(self._bucket_ev[i])
      run()

  def backward(self, glogits: t.Tensor, grad_hook=None):
    """Fills eng.store.grads with d loss / d params given d loss / d logits.  With `grad_hook`, finished
    ranges of the slab are handed over while the rest of backward still runs (GRAD_BUCKET_LABELS)."""
    eng, be, B = self.eng, self.be, self.B
    cv, bn = eng.convs, eng.bns
    assert self.training, "backward needs a training-mode forward"
    if eng._dgrad_pack_pending:
      t.cuda.current_stream().wait_event(eng._dgrad_packed)
      eng._dgrad_pack_pending = False
    elif eng.dgrad_dirty:
      eng.pack_dgrad_weights()
    if not eng._gpacked_zeroed:
      be.zero(eng.gpacked)
    eng._gpacked_zeroed = False
    skips_on_side = self.async_skip and self.side is not None and self.trace is None and self.ray_side
    if skips_on_side:
      # the four skip-map gradients are written and read on the side stream only (scatter -> compress gradients): zeroed there,
      # off the step's chain (a memset is a launch of its own)
      with t.cuda.stream(self.side), _lib.pinned_stream(self.side):
        be.zero(self.gsmap_slab)
    else:
      be.zero(self.gsmap_slab)                     # the four skip-map gradients: one launch instead of four memsets
    if grad_hook is None and self.side is not None and self.trace is None:
      grad_hook = _no_exchange                     # un-pack the finished buckets on the side stream all the same
    L = eng.latent
    g_out = glogits
    for k in range(6, 1, -1):
      d = self.dec[k]
      r, S = d["r"], d["r"] ** 3
      ro, So = 2 * r, (2 * r) ** 3
      p = f"decoder.stage_{k}."
      b1_, b2_ = bn[p + "b1."], bn[p + "b2."]
      ctot = g_out.shape[1]
      if k < 6:    # ray-traced skip: scatter-add, compress conv grads
        skip_bwd_async = self.async_skip and self.side is not None and self.trace is None
        if skip_bwd_async:
          # off the data-gradient chain: the feature-map gradients are only needed when the encoder's backward starts.
          # The scatter goes with it (CRN_RAY_SIDE=0 keeps it on the main stream: +0.07 ms per step).  Beside the split-bf16
          # convolutions of round 3 its sums differed from run to run by 1e-2 ... 1e-1 of the map's range: three dependent
          # MFMAs of a neighbour wave with idle cycles between them made VALU results of the scatter's waves go missing
          # (tools/mfma_neighbour.py).  The convolutions issue their three products as one block of adjacent
          # MFMAs now (mfma3, csrc/conv_bf3.hip): beside them the scatter is exact in 82 of 82 steps and in 660 of 660
          # two-kernel runs; test_run_to_run_gradient_spread_default_mode watches it.  (Round 5: the scatter has no projection left.)
          ray_side = self.ray_side
          if not ray_side:
            self._ray_bwd(k, g_out)
          self._skip_bwd_ev[k].record()             # (also the hand-over event of this stage's transposed-conv weight gradient below)
          with t.cuda.stream(self.side), _lib.pinned_stream(self.side):
            self.side.wait_event(self._skip_bwd_ev[k])
            if ray_side:
              self._ray_bwd(k, g_out)
            self._skip_bwd(k, g_out, on_side=True)
            if k == 2:
              self._skip_bwd_done.record(self.side)
          self._handed_over = True                   # nothing was launched on the main stream since that event
        else:
          self._skip_bwd(k, g_out, on_side=False)
      ct = cv[p + "t1."]
      gv = self.s2d(g_out, d["cout"], (2, 2, 2))
      tr2 = Transform(b2_.scale, b2_.shift, pre_relu=True)
      self._wgrad(ct, self.vw(d["w"]), tr2, gv)
      self._handed_over = False
      if k == 6:   # gradient of the logits comes from the loss kernel; below it is a bn_bwd output
        if (self.side is not None and self.trace is None and not self.wg_batch_dec and os.environ.get("CRN_BIAS_GRAD_SIDE", "1") != "0"):
          # one pass over the loss gradient (470 MB at 14 classes) whose result is only read by the bucket's un-pack: it follows the
          # layer's weight gradient on the side stream (ordered behind the same hand-over event) instead of sitting in front of the
          # data-gradient chain
          with t.cuda.stream(self.side), _lib.pinned_stream(self.side):
            self._bias_grad(ct, g_out, So, ctot * So)
        else:
          self._bias_grad(ct, g_out, So, ctot * So)
      cc = cv[p + "c1."]
      # every conv bias gradient below is sum(dx) of the norm that consumes the conv output: fused
      # into bn_bwd (dsum) instead of a second pass over dx.
      # A data gradient is the output gradient of the norm in front of its conv: where the launch does not split its
      # reduction (stages 5-6 at the bench batch) it also leaves that norm's two backward sums, and the norm's backward
      # is its second pass alone (_dgrad_bn_bwd)
      if ct.ct_d is not None and self._math(ct, "dgrad") == "bf16x3":
        self._timed("dgrad " + ct.name, lambda: be.convt_par_dgrad(g_out, d["cout"], ct.ct_d, d["gv2"], False,
                                                                     host_table=eng.ct_tables["bwd"][2], resident=ct.ct_kind == "res"))
        ct_done = False
      else:
        ct_done = self._dgrad_bn_bwd(ct, gv, d["gv2"], d["w"], d["cmid"], S, b2_, d["gw"], cc)
      if not ct_done:
        be.bn_bwd(d["w"], d["cmid"] * S, d["gv2"], d["cmid"] * S, B, d["cmid"], S, True, False,
                  b2_.gamma, b2_.scale, b2_.shift, b2_.saved, d["gw"], d["cmid"] * S, b2_.dgamma, b2_.dbeta,
                  dsum=cc.dbias, ndsum=cc.n_ref)
      tr1 = Transform(b1_.scale, b1_.shift, pre_relu=True)
      self._wgrad(cc, self.vw(d["u"]), tr1, self.vw(d["gw"]))
      cprev = cv[f"decoder.stage_{k - 1}.t1."]        # produced this stage's input (first n_ref channels)
      if not self._dgrad_bn_bwd(cc, self.vw(d["gw"]), d["gv1"], d["u"], d["cin"], S, b1_, d["gu"], cprev):
        be.bn_bwd(d["u"], d["cin"] * S, d["gv1"], d["cin"] * S, B, d["cin"], S, True, False,
                  b1_.gamma, b1_.scale, b1_.shift, b1_.saved, d["gu"], d["cin"] * S, b1_.dgamma, b1_.dbeta,
                  dsum=cprev.dbias, ndsum=cprev.n_ref)
      g_out = d["gu"]
      if self.wg_batch_dec:
        self._flush_wgrads()
      if k == 3:
        self._grads_ready("decoder.stage_3.", grad_hook)
    # stage_1 / stage_0
    c1 = cv["decoder.stage_1.t1."]
    b = bn["decoder.stage_1.b1."]
    zv = self.vw(self.z0.view(B, L + 3, 1, 1, 1))
    gv = self.flat(g_out)
    self._wgrad(c1, zv, Transform(b.scale, b.shift, pre_relu=True), gv)
    self._dgrad(c1, gv, self.vw(self.gv0.view(B, L + 3, 1, 1, 1)))
    be.bn_bwd(self.z0, L + 3, self.gv0, L + 3, B, L + 3, 1, True, False, b.gamma, b.scale, b.shift,
              b.saved, self.gz0, L + 3, b.dgamma, b.dbeta)
    s = eng.store
    be.linear_bwd(self.avg, s.view("decoder.stage_0.weight"), self.gz0, L + 3, B, 2048, L, self.gavg,
                  s.view("decoder.stage_0.weight", grad=True), s.view("decoder.stage_0.bias", grad=True))
    self._grads_ready("decoder.stage_0.", grad_hook)
    # encoder
    if self.async_skip and self.side is not None and self.trace is None:
      t.cuda.current_stream().wait_event(self._skip_bwd_done)          # gfeat of all four stages
    f5, g5 = self.feat["stage5"], self.gfeat["stage5"]
    last = self.blocks[-1]
    S5 = f5.shape[2] * f5.shape[3]
    be.relu_mean_bwd(f5, self.gavg, B, 2048, S5, f5.stride(0), last["gpre"], 2048 * S5, False)
    be.affine_add_relu(last["gpre"], None, None, g5, None, None, B, 2048, S5, 2048 * S5, g5.stride(0),
                       None, 0, last["gpre"], 2048 * S5, False)
    g_in = None          # gradient wrt the block's output (post-ReLU), None for the last block
    for blk in reversed(self.blocks):
      g_in = self._block_bwd(blk, g_in)
      self._wg_blocks += 1
      if self._wg_blocks >= self.wg_batch:
        self._flush_wgrads()
      if blk["prefix"] in GRAD_BUCKET_LABELS:
        self._grads_ready(blk["prefix"], grad_hook)
    # stem: g_in = d p1
    b1 = bn["encoder.stage1_part2.bn."]
    H1, W1 = self.stem_hw
    be.maxpool_bwd(g_in, self.p1_arg, B, 64, H1, W1, self.gy1)
    S1 = H1 * W1
    cs = cv["encoder.stage1.conv."]
    be.bn_bwd(self.y1, 64 * S1, self.gy1, 64 * S1, B, 64, S1, False, False, b1.gamma, b1.scale, b1.shift,
              b1.saved, self.gy1b, 64 * S1, b1.dgamma, b1.dbeta, dsum=cs.dbias, ndsum=cs.n_ref)
    self._wgrad(cs, self.s2d(self.img, 3, (1, 2, 2)), None, self.vw(self.gy1b))
    # packed weight grads -> reference layout inside the flat grad slab (1 launch)
    if grad_hook is not None:
      self._grads_ready("", grad_hook)
      self._join_side()
      return
    self._join_side()
    be.copy_tiles(eng.gpacked, eng.store.grads, eng.unpack_tiles, reverse=True)

  def _ray_bwd(self, k: int, g_out: t.Tensor):
    """Scatter-add of the skip channels' gradient into the 2-D map gradient (ray_traced_skip_connection.py:135, autograd)."""
    d = self.dec[k]
    ro, (hh, ww) = 2 * d["r"], self.skip_hw[k]
    if self.ray_idx is not None:
      self.be.ray_sample_bwd_idx(g_out[:, d["cout"]:], g_out.stride(0), self.B, self.eng.skip_ch[k], ro, ro, ro,
                                 self.ray_idx[k], self.gsmap[k], self.gsmap[k].stride(0), hh, ww, False)
      return
    self.be.ray_sample_bwd(g_out[:, d["cout"]:], g_out.stride(0), self.B, self.eng.skip_ch[k], ro, ro, ro,
                           self.layer_mats[k - 2], self.offset, self.gsmap[k], self.gsmap[k].stride(0), hh, ww, False)

  def _skip_bwd(self, k: int, g_out: t.Tensor, on_side: bool):
    """Backward of the skip connection into decoder stage k+1 (g_out = gradient of that stage's concat buffer)."""
    eng, be, B = self.eng, self.be, self.B
    d = self.dec[k]
    (hh, ww), ns = self.skip_hw[k], eng.skip_ch[k]
    if not on_side:                             # (asynchronous skip path: the caller ran the scatter on the main stream)
      self._ray_bwd(k, g_out)
    cs = eng.convs[f"decoder.rt_skip_{k}.compress_channels."]
    ft = self.feat[self.skip_src[k]]
    if on_side:                                 # already on the weight-gradient stream: no hand-over event
      g = cs.fwd
      math = ("bf16x3_2d" if (eng.encoder_e2d and eng.wgrad_2d and self._wg2d_shape_ok(self.vw(self.gsmap[k]), g.window))
              else self._math(cs, "wgrad"))
      with _lib.roctx_range("wgrad " + cs.name):
        be.conv_wgrad(self.vw(ft), None, self.vw(self.gsmap[k]), cs.gwf, g.npad, g.window, g.pad_lo, False,
                      boxes=(g.n_boxes, g.c_boxes), math=math)
    else:
      self._wgrad(cs, self.vw(ft), None, self.vw(self.gsmap[k]))
    self._bias_grad(cs, self.gsmap[k], hh * ww, ns * hh * ww)
    self._dgrad(cs, self.vw(self.gsmap[k]), self.vw(self.gfeat[self.skip_src[k]]))

  def _block_bwd(self, blk, g_out: Optional[t.Tensor]) -> t.Tensor:
    eng, be, B = self.eng, self.be, self.B
    cv, bn = eng.convs, eng.bns
    p = blk["prefix"]
    h = blk["h"]; S = h * blk["w"]
    f1, f2, f3 = blk["f"]
    ba, bb, bc = bn[p + "op_a.bn."], bn[p + "op_b.bn."], bn[p + "op_c.bn."]
    gpre = blk["gpre"]
    cc, cb, ca = cv[p + "op_c.conv."], cv[p + "op_b.conv."], cv[p + "op_a.conv."]
    # d pre = (out > 0 ? g_out : 0) [+ the gradient that arrived through the skip connection]: formed by the backward launch of
    # op_c's norm itself on the HIP backend (crn_batch_renorm_bwd_head), by a launch of its own otherwise
    # (g_out None: gpre already holds d pre -- last block of the encoder)
    fused_head = g_out is not None and hasattr(be, "bn_bwd_head") and eng.fuse_tail
    # g_out may still be in the compact form the down-sampling block behind this one left it in (below): the fused head reads
    # it from there, anything else gets it expanded first
    gc, self._g_compact = self._g_compact, None
    if gc is not None and not fused_head:
      be.stride2_scatter(gc, g_out)
      gc = None
    if blk["final"]:
      act, sB_act, g2, sB_g2 = (self.feat[blk["stage"]], self.feat[blk["stage"]].stride(0),
                                self.gfeat[blk["stage"]], self.gfeat[blk["stage"]].stride(0))
    else:
      act, sB_act, g2, sB_g2 = blk["out"], f3 * S, None, 0
    # conv bias gradients = sum(dx) of the following norm, fused into bn_bwd (dsum)
    if fused_head:
      be.bn_bwd_head(blk["yc"], f3 * S, gpre, f3 * S, g_out, f3 * S, act, sB_act, g2, sB_g2, B, f3, S, bc.gamma, bc.scale,
                     bc.shift, bc.saved, blk["gyc"], f3 * S, bc.dgamma, bc.dbeta, dsum=cc.dbias, ndsum=cc.n_ref,
                     g_compact=gc, W=blk["w"])
    else:
      if g_out is not None:
        be.relu_bwd_add(g_out, act, g2, B, f3, S, f3 * S, sB_act, sB_g2, gpre, f3 * S)
      be.bn_bwd(blk["yc"], f3 * S, gpre, f3 * S, B, f3, S, False, False, bc.gamma, bc.scale, bc.shift,
                bc.saved, blk["gyc"], f3 * S, bc.dgamma, bc.dbeta, dsum=cc.dbias, ndsum=cc.n_ref)
    trb = Transform(bb.scale, bb.shift, post_relu=True)
    tra = Transform(ba.scale, ba.shift, post_relu=True)
    self._wgrad(cc, self.vw(blk["yb"]), trb, self.vw(blk["gyc"]))
    if self.eng.defer_reduce:
      be.splitk_defer()                       # the norm's backward below is the only reader of gab: it adds up the splits
    self._dgrad(cc, self.vw(blk["gyc"]), self.vw(blk["gab"]))
    be.bn_bwd(blk["yb"], f2 * S, blk["gab"], f2 * S, B, f2, S, False, True, bb.gamma, bb.scale, bb.shift,
              bb.saved, blk["gyb"], f2 * S, bb.dgamma, bb.dbeta, dsum=cb.dbias, ndsum=cb.n_ref)
    self._wgrad(cb, self.vw(blk["ya"]), tra, self.vw(blk["gyb"]))
    if self.eng.defer_reduce:
      be.splitk_defer()
    self._dgrad(cb, self.vw(blk["gyb"]), self.vw(blk["gaa"]))
    be.bn_bwd(blk["ya"], f1 * S, blk["gaa"], f1 * S, B, f1, S, False, True, ba.gamma, ba.scale, ba.shift,
              ba.saved, blk["gya"], f1 * S, ba.dgamma, ba.dbeta, dsum=ca.dbias, ndsum=ca.n_ref)
    cur = blk["in"]
    xin = self.vw(blk["xs"]) if blk["stride"] == 2 else self.vw(cur)
    self._wgrad(ca, xin, None, self.vw(blk["gya"]))
    if blk["down"]:
      bs = bn[p + "shortcut.bn."]
      csn = cv[p + "shortcut.conv."]
      be.bn_bwd(blk["ys"], f3 * S, gpre, f3 * S, B, f3, S, False, False, bs.gamma, bs.scale, bs.shift,
                bs.saved, blk["gys"], f3 * S, bs.dgamma, bs.dbeta, dsum=csn.dbias, ndsum=csn.n_ref)
      self._wgrad(csn, xin, None, self.vw(blk["gys"]))
      gin = blk["gin"]
      gv = self.vw(blk["gxs"]) if blk["stride"] == 2 else self.vw(gin)
      self._dgrad(ca, self.vw(blk["gya"]), gv)
      self._dgrad(csn, self.vw(blk["gys"]), gv, accumulate=True)
      if blk["stride"] == 2:
        if hasattr(be, "bn_bwd_head") and eng.fuse_tail and os.environ.get("CRN_FUSE_GATHER", "1") != "0":
          self._g_compact = blk["gxs"]                # expanded by its reader: the norm backward at the head of the block in front
        else:
          be.stride2_scatter(blk["gxs"], gin)         # every element of gin is written (zeros between the samples)
      return gin
    # identity block: d in = d pre + dgrad(op_a)
    self._dgrad(ca, self.vw(blk["gya"]), self.vw(gpre), accumulate=True)
    return gpre
