#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING THE REFERENCE (read-only) in this
container and checks the oracle restatement against it on the way.

Run:  python oracle/gen_golden.py            (needs /root/reference; CPU only)

Nothing of the reference is copied: the outputs are input/expected-output
vectors only.  Weights come from oracle.corenet_oracle.make_state(seed), so the
GPU box can regenerate identical weights without the reference.
"""
import os
import sys
import types

import numpy as np
import torch as t

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("CORENET_REFERENCE", "/root/reference/src")
sys.path.insert(0, REF)
# dataclasses_jsonschema is not installed; forward/backward never call to_dict
_m = types.ModuleType("dataclasses_jsonschema")
_m.JsonSchemaMixin = type("JsonSchemaMixin", (), {})
sys.modules["dataclasses_jsonschema"] = _m

from corenet import configuration as C                      # noqa: E402  (reference)
from corenet.model import core_net, batch_renorm, losses    # noqa: E402
from corenet.model import ray_traced_skip_connection as rts  # noqa: E402
from corenet import voxel_metrics                            # noqa: E402
from oracle import corenet_oracle as O                       # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
t.manual_seed(0)
t.set_num_threads(8)


def maxrel(a, b):
  a, b = a.double(), b.double()
  return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def ref_model(num_classes, sd):
  cfg = C.CoreNetConfig(decoder=C.DecoderConfig(
      resolution=(128, 128, 128), num_output_channels=num_classes,
      last_upscale_factor=2, latent_channels=64, skip_fraction=0.75))
  net = core_net.CoreNet(cfg)
  net.load_state_dict({k: v.clone() for k, v in sd.items()})
  return net


def gen_model(tag, num_classes, nbt, batch, loss_name, training=True, store_all_grads=False):
  sd = O.make_state(seed=0, num_classes=num_classes, nbt=nbt)
  image, v2s, off, grid = O.synthetic_batch(batch, seed=0, num_classes=num_classes)
  net = ref_model(num_classes, sd)
  net.train(training)
  logits = net(image, v2s, off)
  loss_fn = getattr(losses, loss_name)
  loss = loss_fn(grid, logits)
  out = dict(logits_sub=logits.detach()[:, :, ::16, ::16, ::16].numpy(),
             logits_sum=np.float64(logits.double().sum().item()),
             logits_abs_sum=np.float64(logits.double().abs().sum().item()),
             loss=np.float32(loss.item()))
  # the oracle restatement on the same inputs
  sd_o = {k: v.clone() for k, v in sd.items()}
  if training:
    for k in sd_o:
      if sd_o[k].dtype == t.float32 and ("running" not in k):
        sd_o[k].requires_grad_(True)
  lo = O.corenet_forward(sd_o, image, v2s, off, training=training)
  print(f"[{tag}] oracle vs reference logits max-rel = {maxrel(lo.detach(), logits.detach()):.3e}")
  assert maxrel(lo.detach(), logits.detach()) < 1e-4
  if training:
    loss.backward()
    lo_loss = getattr(O, loss_name)(grid, lo)
    lo_loss.backward()
    gn, worst = {}, 0.0
    # A conv bias in front of a train-mode BatchRenorm has a true gradient of exactly 0; what either side holds
    # there is summation-order noise (it depends on how many threads the machine gives torch), so the error of
    # every parameter is taken relative to max(its own gradient scale, 1e-3 of the largest gradient of the model)
    gmax = max(float(p.grad.abs().max()) for _, p in net.named_parameters())
    for name, p in net.named_parameters():
      gn[name] = np.float64(p.grad.double().norm().item())
      den = max(float(p.grad.abs().max()), 1e-3 * gmax)
      worst = max(worst, float((sd_o[name].grad.double() - p.grad.double()).abs().max()) / den)
    print(f"[{tag}] oracle vs reference worst param-grad error (of the tensor's scale) = {worst:.3e}")
    assert worst < 2e-3
    if store_all_grads:
      # element-wise vectors of EVERY parameter gradient (a fixed strided subsample of <= 512 elements per tensor):
      # this fixture (B=2, num_batches_tracked=30000: r/d clamps live, statistics over two samples) is the
      # well-conditioned one, so the whole backward pass can be pinned element by element
      # ... and the SAME arithmetic in fp64 (the oracle, which equals the reference bit for bit in fp32 above; the
      # camera stays fp32 so that the gather indices are the same): what the gradients would be without rounding.
      # gnoise = the error of the reference's own fp32 gradients against it, per tensor over all elements, on the
      # scale used everywhere here (the tensor's own max, floored at 1e-3 of the model's largest gradient) -- the
      # conditioning of each tensor, measured: up to 1e-1 in encoder stage 5 although the logits move by 2e-6.
      sd64 = {k: (v.double() if v.dtype == t.float32 else v.clone()) for k, v in sd.items()}
      for k in sd64:
        if sd64[k].dtype == t.float64 and "running" not in k:
          sd64[k].requires_grad_(True)
      feats, avg = O.resnet50_features(O.preprocess_image_caffe(image).double(), sd64, True)
      l64 = O.decoder_forward(feats, avg, sd64, v2s, off, (128, 128, 128), True)
      getattr(O, loss_name)(grid, l64).backward()
      print(f"[{tag}] fp32 vs fp64 logits max-rel = {maxrel(logits.detach(), l64.detach()):.3e}")
      worst_noise = 0.0
      for name, p in net.named_parameters():
        g, g64 = p.grad.reshape(-1), sd64[name].grad.reshape(-1)
        st = max(1, -(-g.numel() // 512))
        out["gsub::" + name] = g[::st].numpy().copy()
        out["g64sub::" + name] = g64[::st].float().numpy().copy()
        out["gmax::" + name] = np.float32(g64.abs().max().item())
        den = max(float(g64.abs().max()), 1e-3 * gmax)
        out["gnoise::" + name] = np.float32(float((g.double() - g64).abs().max()) / den)
        worst_noise = max(worst_noise, float(out["gnoise::" + name]))
      print(f"[{tag}] reference fp32 vs fp64 gradients: worst tensor {worst_noise:.3e} of its scale")
    out["grad_names"] = np.array(list(gn.keys()))
    out["grad_norms"] = np.array(list(gn.values()))
    # a few full gradients (small tensors) for direct comparison
    for name in ["decoder.stage_6.t1.weight", "decoder.stage_6.b2.weight",
                 "decoder.rt_skip_5.compress_channels.weight",
                 "decoder.stage_0.bias", "encoder.stage1.conv.bias"]:
      out["grad::" + name] = dict(net.named_parameters())[name].grad.numpy()
    # running stats after the step
    sdn = net.state_dict()
    for name in ["decoder.stage_6.b1.running_mean", "decoder.stage_6.b1.running_var",
                 "encoder.stage1_part2.bn.running_mean", "encoder.stage5.c.op_c.bn.running_var"]:
      out["buf::" + name] = sdn[name].numpy()
      assert maxrel(sd_o[name], sdn[name]) < 1e-4
  np.savez_compressed(os.path.join(OUT, f"model_{tag}.npz"), **out)


def gen_batch_renorm():
  out = {}
  g = t.Generator().manual_seed(7)
  x = t.randn(3, 5, 4, 6, generator=g) * 2 + 0.7
  out["x"] = x.numpy()
  for tag, nbt, training in (("train0", 0, True), ("train30k", 30000, True), ("eval", 123, False)):
    bn = batch_renorm.BatchRenorm(5, eps=1e-3)
    with t.no_grad():
      bn.weight.copy_(t.tensor([1.0, 0.5, 2.0, 1.5, 0.8]))
      bn.bias.copy_(t.tensor([0.0, 0.1, -0.2, 0.3, 1.0]))
      bn.running_mean.copy_(t.tensor([0.5, 0.9, -0.3, 2.5, 0.7]))
      bn.running_var.copy_(t.tensor([4.0, 0.2, 9.0, 1.0, 30.0]))
      bn.num_batches_tracked.fill_(nbt)
    bn.train(training)
    xi = x.clone().requires_grad_(True)
    y = bn(xi)
    gy = t.randn(y.shape, generator=t.Generator().manual_seed(8))
    y.backward(gy)
    out[f"{tag}_y"] = y.detach().numpy()
    out[f"{tag}_gx"] = xi.grad.numpy()
    out[f"{tag}_gw"] = bn.weight.grad.numpy()
    out[f"{tag}_gb"] = bn.bias.grad.numpy()
    out[f"{tag}_rm"] = bn.running_mean.numpy()
    out[f"{tag}_rv"] = bn.running_var.numpy()
    out["gy"] = gy.numpy()
    # oracle check
    sd = {"weight": bn.weight.detach().clone(), "bias": bn.bias.detach().clone(),
          "running_mean": t.tensor([0.5, 0.9, -0.3, 2.5, 0.7]),
          "running_var": t.tensor([4.0, 0.2, 9.0, 1.0, 30.0]),
          "num_batches_tracked": t.tensor(nbt)}
    yo = O.batch_renorm(x, sd, "", training)
    assert maxrel(yo, y.detach()) < 1e-6, tag
    assert maxrel(sd["running_var"], bn.running_var) < 1e-6
  np.savez_compressed(os.path.join(OUT, "batch_renorm.npz"), **out)
  print("[batch_renorm] ok")


def gen_sample_grid2d():
  """SampleGrid2d fwd/bwd incl. an edge-case camera (outside image + behind camera)."""
  out = {}
  g = t.Generator().manual_seed(11)
  B, Cin, Cs, h, w, R = 2, 7, 5, 16, 16, 16
  mod = rts.SampleGrid2d(Cin, Cs, (R, R, R))
  with t.no_grad():
    mod.compress_channels.weight.copy_(t.randn(Cs, Cin, 1, 1, generator=g))
    mod.compress_channels.bias.copy_(t.randn(Cs, generator=g))
  src = t.randn(B, Cin, h, w, generator=g)
  cam = O.canonical_camera()
  m0 = cam @ O.scale([1.0 / R] * 3)
  # edge case: shift the grid so ~30% projects outside and a slab is behind the camera
  m1 = cam @ O.translate(t.tensor([0.35, -0.2, -0.55])) @ O.scale([1.6 / R] * 3)
  mats = t.stack([m0, m1])
  off = t.tensor([[0.5, 0.5, 0.5], [0.25, 0.75, 0.5]])
  srcg = src.clone().requires_grad_(True)
  y = mod(srcg, mats, off)
  gy = t.randn(y.shape, generator=g)
  y.backward(gy)
  out.update(src=src.numpy(), weight=mod.compress_channels.weight.detach().numpy(),
             bias=mod.compress_channels.bias.detach().numpy(), mats=mats.numpy(),
             off=off.numpy(), y=y.detach().numpy(), gy=gy.numpy(),
             gsrc=srcg.grad.numpy(),
             gweight=mod.compress_channels.weight.grad.numpy(),
             gbias=mod.compress_channels.bias.grad.numpy())
  yo = O.sample_grid2d(src, mod.compress_channels.weight.detach(),
                       mod.compress_channels.bias.detach(), mats, off, (R, R, R))
  mism = int((yo != y.detach()).sum())
  frac_zero = float((y.detach()[:, 0] == 0).float().mean())
  print(f"[sample_grid2d] oracle vs reference mismatching elements: {mism}; zero fraction {frac_zero:.2f}")
  assert mism == 0
  # index parity at the real decoder scales with the canonical camera
  for res, hw in ((8, 8), (16, 16), (32, 32), (64, 64)):
    mod = rts.SampleGrid2d(1, 1, (res,) * 3)
    with t.no_grad():
      mod.compress_channels.weight.fill_(1.0); mod.compress_channels.bias.fill_(0.0)
    idmap = t.arange(hw * hw, dtype=t.float32).reshape(1, 1, hw, hw) + 1
    m = (cam @ O.scale([1.0 / 128] * 3) @ O.scale([128.0 / res] * 3))[None]
    o = t.full((1, 3), 0.5)
    yr = mod(idmap, m, o)
    yo = O.ray_sample(idmap, m, o, (res,) * 3)
    n = int((yr.detach() != yo).sum())
    print(f"[ray index parity] res {res}: mismatches {n}")
    assert n == 0
    out[f"idx_{res}"] = yr.detach().numpy().astype(np.int32)[0, 0]
  np.savez_compressed(os.path.join(OUT, "sample_grid2d.npz"), **out)


def gen_losses():
  out = {}
  g = t.Generator().manual_seed(5)
  logits = t.randn(2, 5, 6, 7, 8, generator=g)
  gt = t.randint(0, 5, (2, 6, 7, 8), generator=g)
  out["logits"], out["gt"] = logits.numpy(), gt.numpy()
  for name in ("iou_agnostic", "iou_fgbg", "xent", "xent_times_iou_agnostic", "xent_times_iou_fgbg"):
    l = logits.clone().requires_grad_(True)
    v = getattr(losses, name)(gt, l)
    v.backward()
    out[name] = np.float32(v.item())
    out[name + "_grad"] = l.grad.numpy()
    vo = getattr(O, name)(gt, logits)
    assert abs(float(vo) - float(v)) < 1e-6
  # per-voxel loss weights (losses.py:47-49,99-102,134-136), incl. a slab of zeros
  w = t.rand(2, 6, 7, 8, generator=g) * 1.5
  w[0, 0] = 0
  out["weights"] = w.numpy()
  for name in ("iou_agnostic", "iou_fgbg", "xent", "xent_times_iou_agnostic", "xent_times_iou_fgbg"):
    l = logits.clone().requires_grad_(True)
    v = getattr(losses, name)(gt, l, w)
    v.backward()
    out[name + "_w"] = np.float32(v.item())
    out[name + "_w_grad"] = l.grad.numpy()
    assert abs(float(getattr(O, name)(gt, logits, w)) - float(v)) < 1e-6
  np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)
  print("[losses] ok")


def gen_metrics():
  """Mean IoU exactly as the reference computes it: voxel_metrics.compute_tfpn + compute_voxel_metrics (NaN for a
  class with tp == 0, voxel_metrics.py:118-138), then the pandas mean over the non-void class columns of
  evaluation_results.py:188-210,262-266 (`mm.iloc[:, 1:-1].T.mean().iou`, NaNs skipped).  evaluation_results itself
  cannot be imported here (tensorboard, ...), so those four lines of DataFrame handling are reproduced with pandas."""
  import dataclasses
  import pandas
  g = t.Generator().manual_seed(21)
  out, cms = {}, []
  for i, k in enumerate((2, 5, 14, 14, 14)):
    cm = t.randint(0, 60, (k, k), generator=g)
    if i == 3:
      cm[5] = 0; cm[:, 5] = 0; cm[9, 9] = 0        # an absent class and a never-hit class
    if i == 4:
      cm[1:] = 0                                   # only void in the ground truth: every class is NaN
    cms.append(cm)
    tfpn = voxel_metrics.compute_tfpn(cm.to(t.float64))
    vm = voxel_metrics.compute_voxel_metrics(tfpn)
    classes = ["void"] + [f"c{j}" for j in range(1, k)]
    df = pandas.DataFrame({f.name: getattr(vm, f.name).numpy() for f in dataclasses.fields(vm)}, index=classes).T
    df = pandas.concat([df, pandas.DataFrame({"iou": [0.0], "precision": [0.0], "recall": [0.0]}, index=["__global__"]).T], axis=1)
    miou = float(df.iloc[:, 1:-1].T.mean().iou)
    out[f"cm_{i}"] = cm.numpy(); out[f"miou_{i}"] = np.float64(miou); out[f"iou_{i}"] = vm.iou.numpy()
    mo = O.mean_iou(cm)
    assert (np.isnan(miou) and np.isnan(mo)) or abs(mo - miou) < 1e-12, (i, mo, miou)
  np.savez_compressed(os.path.join(OUT, "metrics.npz"), **out)
  print("[metrics] ok")


def gen_super_resolution():
  """x2 super-resolution (super_resolution.py:46-129) of the h7 eval model, B=1: the reference's own
  SuperResolutionInference + super_resolution_from_state, 8 full forwards.  corenet.pipeline / corenet.state
  cannot be imported here (jq, tensorboard, ...): they are replaced by empty modules that only provide the
  base class / type name super_resolution.py refers to; every line that computes is the reference's."""
  pm = types.ModuleType("corenet.pipeline"); pm.InferenceFn = type("InferenceFn", (), {})
  sm = types.ModuleType("corenet.state"); sm.State = type("State", (), {})
  sys.modules["corenet.pipeline"] = pm; sys.modules["corenet.state"] = sm
  import corenet as _c
  _c.pipeline, _c.state = pm, sm
  from corenet import super_resolution as SR
  sd = O.make_state(seed=0, num_classes=2, nbt=100, logit_scale=2e-4)     # logits O(1): the pmf is well conditioned
  image, v2s, off, _ = O.synthetic_batch(1, seed=0, num_classes=2)
  net = ref_model(2, sd); net.eval()
  state = sm.State(); state.model = net
  camera = O.canonical_camera()[None]
  v2v = O.scale([128.0] * 3)[None]                      # view -> voxel of the native grid (pipeline.py:148)
  go = t.full((1, 3), 0.5)
  with t.no_grad():
    sr = SR.super_resolution_from_state(state)
    pmf = sr(image, camera, v2v, go, (256, 256, 256))
    po = O.super_resolution(sd, image, camera, v2v, go, 2)
  print(f"[super_resolution] oracle vs reference pmf max-abs = {float((po - pmf).abs().max()):.3e}")
  assert float((po - pmf).abs().max()) < 1e-5
  np.savez_compressed(os.path.join(OUT, "super_resolution_h7_x2.npz"),
                      pmf_sub=pmf[:, :, ::16, ::16, ::16].numpy(), pmf_odd=pmf[:, :, 1::32, 1::32, 1::32].numpy(),
                      pmf_sum=np.float64(pmf.double().sum().item()),
                      fg_sum=np.float64(pmf[:, 1].double().sum().item()),
                      native_offsets=sr.get_native_offsets((256, 256, 256), go).numpy())
  print("[super_resolution] ok")


def write_data_fixture(root):
  """A tiny synthetic dataset in the reference's on-disk format (doc/data_format_and_coordinate_systems.md:9-31,
  dataset.py:48-51): two scenes of 3 + 2 objects over three classes, meshes of 8-40 triangles, 24x32 images
  (lossless WebP for the high-realism image, PNG for the low-realism one).  Written by this script, not taken
  from the reference's data."""
  import io
  import json
  import PIL.Image
  rng = np.random.RandomState(7)
  classes = [("03001627", "chair"), ("02958343", "car"), ("04256520", "sofa")]
  meshes = {("03001627", "aa11"): 8, ("02958343", "bb22"): 40, ("04256520", "cc33"): 13, ("02958343", "dd44"): 21}
  for (lab, name), nt in meshes.items():
    os.makedirs(os.path.join(root, "meshes", lab), exist_ok=True)
    pngs = np.empty((), dtype=object); pngs[()] = [b"", b"\x89PNG-not-decoded"]     # read with .scalar() (scene.py:149)
    np.savez(os.path.join(root, "meshes", lab, name + ".npz"),
             vertices=(rng.rand(nt, 3, 3).astype(np.float32) - 0.5),
             normals=rng.randn(nt, 3, 3).astype(np.float32), material_ids=rng.randint(0, 2, nt).astype(np.int32),
             texcoords=rng.rand(nt, 3, 2).astype(np.float32), diffuse_colors=rng.rand(2, 3).astype(np.float32),
             diffuse_texture_pngs=pngs)

  def encode(img, fmt, **kw):
    b = io.BytesIO(); PIL.Image.fromarray(img).save(b, fmt, **kw); return np.array(b.getvalue())

  def rigid(scale, angle, tr):
    c, s_ = np.cos(angle), np.sin(angle)
    m = np.eye(4, dtype=np.float32)
    m[:3, :3] = scale * np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]], np.float32)
    m[:3, 3] = tr
    return m

  camera = O.canonical_camera().numpy().astype(np.float32)
  scenes = {"s0": [("03001627", "aa11"), ("02958343", "bb22"), ("04256520", "cc33")],
            "s1": [("02958343", "dd44"), ("03001627", "aa11")]}
  os.makedirs(os.path.join(root, "scenes"), exist_ok=True)
  for i, (sid, objs) in enumerate(scenes.items()):
    o2w = np.stack([rigid(0.25 + 0.1 * k, 0.7 * k + i, [0.3 + 0.2 * k, 0.1 * i, 0.4 - 0.15 * k]) for k in range(len(objs))])
    view = rigid(1.0, 0.3 + i, [0.5, 0.45, 0.55]).astype(np.float32)
    np.savez(os.path.join(root, "scenes", sid + ".npz"),
             scene_filename=np.array(sid.encode()), mesh_labels=np.array([o[0] for o in objs]),
             mesh_filenames=np.array([o[1] for o in objs]),
             mesh_visible_fractions=rng.rand(len(objs)).astype(np.float32),
             mesh_object_to_world_transforms=o2w.astype(np.float32), view_transform=view, camera_transform=camera,
             opengl_image=encode(rng.randint(0, 256, (24, 32, 3)).astype(np.uint8), "PNG"),
             pbrt_image=encode(rng.randint(0, 256, (24, 32, 3)).astype(np.uint8), "WEBP", lossless=True))
  with open(os.path.join(root, "dataset.json"), "w") as fl:
    json.dump({"classes": [{"id": c, "human_readable": h} for c, h in classes],
               "files": ["scenes/s0.npz", "scenes/s1.npz"]}, fl, indent=1)


def gen_data_path():
  """N2: the reference's own scene reader, dataset and collate (scene.py:106-151, dataset.py:89-196,
  batched_example.py:68-95) run on the fixture dataset -> tests/golden/data_path.npz.  Import shims, none of
  which computes anything that is pinned: empty `google.*` modules (file_system.py:22,25 - only gs:// paths use
  them), empty `corenet.cc.fill_voxels` / `corenet.geometry.voxelization` (GL, not used by `batch`), and a
  JsonSchemaMixin.from_dict that builds the two config dataclasses from the JSON dict."""
  import dataclasses
  import typing
  for name in ("google", "google.api_core", "google.api_core.exceptions", "google.cloud", "google.cloud.storage"):
    sys.modules.setdefault(name, types.ModuleType(name))
  sys.modules["google"].api_core = sys.modules["google.api_core"]
  sys.modules["google.api_core"].exceptions = sys.modules["google.api_core.exceptions"]
  sys.modules["google"].cloud = sys.modules["google.cloud"]
  sys.modules["google.cloud"].storage = sys.modules["google.cloud.storage"]

  def from_dict(cls, v):
    kw = {}
    for f in dataclasses.fields(cls):
      ft = typing.get_type_hints(cls)[f.name]
      args = typing.get_args(ft)
      if args and dataclasses.is_dataclass(args[0]):
        kw[f.name] = [from_dict(args[0], e) for e in v[f.name]]
      else:
        kw[f.name] = v[f.name]
    return cls(**kw)
  _m.JsonSchemaMixin.from_dict = classmethod(from_dict)
  import corenet as _c
  for name in ("corenet.cc.fill_voxels", "corenet.geometry.voxelization"):
    sys.modules[name] = types.ModuleType(name)
  import corenet.cc as _cc
  import corenet.geometry as _cg
  _cc.fill_voxels = sys.modules["corenet.cc.fill_voxels"]
  _cg.voxelization = sys.modules["corenet.geometry.voxelization"]
  from corenet.data import dataset as RD
  from corenet.data import batched_example as RB
  from corenet.data import scene as RS

  root = os.path.join(OUT, "n2_dataset")
  write_data_fixture(root)
  out = {}
  for hr in (True, False):
    ds = RD.CoReNetDatasetImpl(os.path.join(root, "dataset.json"), os.path.join(root, "meshes"), high_realism=hr)
    els = [ds[i] for i in range(len(ds))]
    b = RB.batch(els)
    tag = "hr" if hr else "lr"
    out[tag + "_input_image"] = b.input_image.numpy()
    if hr:
      out["classes"] = np.array(list(ds.classes))
      out["vertices"] = b.vertices.numpy()
      out["view_transform"] = b.view_transform.numpy(); out["camera_transform"] = b.camera_transform.numpy()
      out["mesh_num_tri"] = t.cat(b.mesh_num_tri).numpy(); out["num_meshes"] = np.array([len(v) for v in b.mesh_num_tri])
      out["mesh_labels"] = t.cat(b.mesh_labels).numpy(); out["scene_id"] = np.array(b.scene_id)
      out["grid_sampling_offset"] = b.grid_sampling_offset.numpy()
      out["raw_vertices"] = t.cat([e.mesh_vertices for e in els]).numpy()
      out["o2w"] = t.cat([e.o2w_transforms for e in els]).numpy()
      # oracle restatement of the collate geometry against the reference
      vo = O.batch_vertices([(e.mesh_vertices, e.mesh_num_tri, e.view_transform, e.o2w_transforms) for e in els])
      assert t.equal(vo, b.vertices), float((vo - b.vertices).abs().max())
      vds = RD.CoReNetDataset(ds, ds.classes)
      big = RD.CoReNetDataset(t.utils.data.ConcatDataset([vds] * 5), ds.classes)      # 10 virtual elements
      out["shuffle_1234"] = big.shuffle(1234).indices.numpy()
      out["fraction_02_07"] = big.take_fraction(0.2, 0.7).indices.numpy()
      out["slice_of_shuffle"] = big.shuffle(7)[3:8].indices.numpy()
  from corenet import distributed as RDist                      # rank's share of a dataset (distributed.py:203-230)
  # torch 2.10's Sampler.__init__ no longer takes the data source the reference passes up (torch 1.7 API)
  t.utils.data.Sampler.__init__ = lambda self, *a, **k: None
  for pad in (True, False):
    out[f"sampler_pad{int(pad)}"] = np.stack([np.pad(
        RDist.DistributedSampler(list(range(10)), r, 4, pad).indices.numpy(), (0, 1), constant_values=-1)[:3]
        for r in range(4)])
  sc = RS.load_from_npz(os.path.join(root, "scenes", "s0.npz"), os.path.join(root, "meshes"), load_extra_fields=True)
  out["s0_normals_1"] = sc.normals[1].numpy(); out["s0_material_ids_2"] = sc.material_ids[2].numpy()
  out["s0_visible"] = sc.mesh_visible_fractions.numpy()
  np.savez_compressed(os.path.join(OUT, "data_path.npz"), **out)
  print("[data_path] ok:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
  if "--only-nbt30k" in sys.argv:
    gen_model("h7_train_b2_nbt30k", 2, 30000, 2, "iou_fgbg", store_all_grads=True)
    sys.exit(0)
  if "--only-b4-nbt30k" in sys.argv:
    gen_model("h7_train_b4_nbt30k", 2, 30000, 4, "iou_fgbg", store_all_grads=True)
    sys.exit(0)
  if "--only-super-resolution" in sys.argv:
    gen_super_resolution()
    sys.exit(0)
  if "--only-data-path" in sys.argv:
    gen_data_path()
    sys.exit(0)
  if "--only-losses-metrics" in sys.argv:
    gen_losses()
    gen_metrics()
    sys.exit(0)
  gen_batch_renorm()
  gen_sample_grid2d()
  gen_losses()
  gen_metrics()
  gen_model("h7_train_b1", 2, 0, 1, "iou_fgbg")
  gen_model("h7_train_b2_nbt30k", 2, 30000, 2, "iou_fgbg", store_all_grads=True)
  gen_model("h7_train_b4_nbt30k", 2, 30000, 4, "iou_fgbg", store_all_grads=True)
  gen_model("h7_eval_b1", 2, 100, 1, "iou_fgbg", training=False)
  gen_model("m9_train_b1", 14, 0, 1, "xent_times_iou_agnostic")
  gen_super_resolution()
  gen_data_path()
  print("golden fixtures written to", OUT)
