"""GPU parity of the whole model path through the drop-in boundary
(corenet_amd.model.core_net.CoreNet) against the golden vectors generated from the
imported reference and against the oracle.

Tolerances.  north_star: logits within 1e-3 relative (fp32).  The reference model
itself, in fp32 vs fp64 on these synthetic inputs, moves by 2e-6 (eval) / 3e-4
(train, B=1) in the logits and by 4e-5 (stage_6.t1) ... 5e-2 (encoder) in the
gradients (measured with the oracle, DESIGN.md "Conditioning"), so gradient checks
deeper than the last layers are norm/cosine checks; the tight backward checks are
per kernel (test_kernels_gpu.py) plus the fp64 host-wiring test (test_host_cpu.py)."""
import os

import numpy as np
import pytest
import torch as t

from oracle import corenet_oracle as O

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


MATHS = ["bf16x3", "fp32"]        # the product default (what bench.py measures and decode_state builds) and the fp32 engine


def _model(nc, sd, decoder_math="fp32"):
  from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
  m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), nc, 2, 64, 0.75)), device="cuda", decoder_math=decoder_math)
  missing = m.load_state_dict(sd)
  assert not missing.missing_keys and not missing.unexpected_keys
  return m


def relerr(a, b):
  a = t.as_tensor(a).double().cpu(); b = t.as_tensor(b).double().cpu()
  return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def test_state_dict_surface():
  sd = O.make_state(0, 2)
  m = _model(2, sd)
  got = m.state_dict()
  assert list(got.keys()) == list(sd.keys())
  for k in sd:
    assert tuple(got[k].shape) == tuple(sd[k].shape) and got[k].dtype == sd[k].dtype, k
    assert t.equal(got[k].cpu(), sd[k]), k
  assert sum(p.numel() for p in m.parameters()) == 36141888
  enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
  m.encoder.load_state_dict(enc)                                      # state.py:69
  assert m.config.to_dict()["decoder"]["resolution"] == (128, 128, 128)


def test_forward_eval_golden():
  z = np.load(os.path.join(G, "model_h7_eval_b1.npz"))
  m = _model(2, O.make_state(0, 2, nbt=100)).eval()
  image, v2s, off, grid = O.synthetic_batch(1, 0, 2)
  with t.no_grad():
    logits = m(image.cuda(), v2s.cuda(), off.cuda())
  assert logits.shape == (1, 2, 128, 128, 128) and logits.dtype == t.float32
  assert relerr(logits[:, :, ::16, ::16, ::16], z["logits_sub"]) < 1e-4
  assert abs(float(logits.double().sum()) - float(z["logits_sum"])) < 1e-4 * float(z["logits_abs_sum"])
  from corenet_amd.model import losses
  assert abs(float(losses.iou_fgbg(grid.cuda(), logits)) - float(z["loss"])) < 1e-5


def TRAIN_LOGIT_TOL(B):
  """Train-mode logits against the reference fixtures.  north_star's tolerance is 1e-3 relative.  The B = 1 fixtures sit at ~4.5e-4
  of it: BatchRenorm over ONE sample at num_batches_tracked = 0 divides by the standard deviation of a single map, which amplifies
  any difference in summation order ~400x (the oracle in fp32 vs fp64 moves by as much; DESIGN section 4, "Conditioning"), so they
  keep the north_star bar.  With B >= 2 the statistics are conditioned like in training and the library measures <= 1e-4: bar 2e-4
  (VERDICT round 5, item 7)."""
  return 1e-3 if B == 1 else 2e-4


def test_bf16x3_mode_against_goldens_and_fp32_mode():
  """Engine(decoder_math="bf16x3"): decoder stages 4-6 on the split-bf16 MFMA engine.  It must hold the same
  bars as the fp32 mode -- eval logits 1e-4 against the reference fixture, train logits 1e-3 (north_star), loss,
  last-layer gradient, running statistics -- and stay close to the fp32 mode of this library on the bench batch
  (B=4, train); measured errors are printed."""
  from corenet_amd.model import losses
  z = np.load(os.path.join(G, "model_h7_eval_b1.npz"))
  m = _model(2, O.make_state(0, 2, nbt=100), "bf16x3").eval()
  image, v2s, off, grid = O.synthetic_batch(1, 0, 2)
  with t.no_grad():
    logits = m(image.cuda(), v2s.cuda(), off.cuda())
  e = relerr(logits[:, :, ::16, ::16, ::16], z["logits_sub"])
  print(f"bf16x3 eval logits vs reference fixture: {e:.2e}")
  assert e < 1e-4
  assert abs(float(losses.iou_fgbg(grid.cuda(), logits)) - float(z["loss"])) < 1e-5
  for tag, nc, nbt, B, lossname in (("h7_train_b1", 2, 0, 1, "iou_fgbg"), ("h7_train_b2_nbt30k", 2, 30000, 2, "iou_fgbg"),
                                    ("m9_train_b1", 14, 0, 1, "xent_times_iou_agnostic")):
    z = np.load(os.path.join(G, f"model_{tag}.npz"))
    m = _model(nc, O.make_state(0, nc, nbt=nbt), "bf16x3").train()
    image, v2s, off, grid = O.synthetic_batch(B, 0, nc)
    logits = m(image.cuda(), v2s.cuda(), off.cuda())
    e = relerr(logits[:, :, ::16, ::16, ::16], z["logits_sub"])
    print(f"bf16x3 {tag} logits vs reference fixture: {e:.2e}")
    assert e < TRAIN_LOGIT_TOL(B), (tag, e)
    loss = getattr(losses, lossname)(grid.cuda(), logits)
    assert abs(float(loss) - float(z["loss"])) < 2e-4 * max(1.0, abs(float(z["loss"])))
    loss.backward()
    eg = relerr(m.get_parameter("decoder.stage_6.t1.weight").grad, z["grad::decoder.stage_6.t1.weight"])
    print(f"bf16x3 {tag} grad stage_6.t1.weight: {eg:.2e}")
    assert eg < 2e-3
    for k in z.files:
      if k.startswith("buf::"):
        assert relerr(m.state_dict()[k[5:]], z[k]) < 1e-4, k
  # bench batch, train mode, both math modes of this library from the same state
  sd = O.make_state(0, 2, nbt=0)
  ma, mb = _model(2, sd, "bf16x3").train(), _model(2, sd, "fp32").train()
  image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(4, 0, 2)]
  la = float(ma.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", lr=4e-4, adam_eps=1e-4))
  lb = float(mb.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", lr=4e-4, adam_eps=1e-4))
  e = relerr(ma.engine.plan(4).logits, mb.engine.plan(4).logits)
  print(f"bf16x3 vs fp32 mode, B=4 train logits: {e:.2e}; loss {la:.6f} vs {lb:.6f}; gradient slab {_slab_err(ma, mb):.2e}")
  assert e < 1e-3 and abs(la - lb) < 1e-4 * abs(lb)


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("hw", [(224, 224), (320, 256)])
def test_other_image_sizes_match_oracle(math, hw):
  """The reference encoder is fully convolutional (resnet50.py:176-186) and the decoder only sees the stage maps through the
  ray-traced skips, so the model runs on any image size; here a plan is built per (batch, H, W).  224 x 224 (stage-5 map 7 x 7: odd
  extents, 49 positions -- no multiple of 4, off every float4 fast path) and 320 x 256 (H != W) in eval mode against the oracle to
  1e-4 on every voxel, and one training forward + backward (logits 1e-3, last layer's gradient and the 64^3 skip's compression
  gradient element-wise); the 256 x 256 plan of the same model still gives what a fresh model gives."""
  from corenet_amd.model import losses
  sd = O.make_state(0, 2, nbt=30000)
  m = _model(2, sd, math)
  _, v2s, off, grid = O.synthetic_batch(2, 0, 2)
  image = t.randint(0, 256, (2, 3) + hw, generator=t.Generator().manual_seed(hw[0]), dtype=t.uint8)
  with t.no_grad():
    want = O.corenet_forward({k: v.clone() for k, v in sd.items()}, image, v2s, off, training=False)
    got = m.eval()(image.cuda(), v2s.cuda(), off.cuda())
  e_eval = relerr(got, want)
  # the default-size plan of the same engine is its own set of buffers: same logits as a fresh model
  i2, v2, o2, _ = O.synthetic_batch(2, 0, 2)
  with t.no_grad():
    a = m(i2.cuda(), v2.cuda(), o2.cuda())
    bref = _model(2, sd, math).eval()(i2.cuda(), v2.cuda(), o2.cuda())
    again = m(image.cuda(), v2s.cuda(), off.cuda())
  assert relerr(a, bref) < 1e-5 and t.equal(again, got)
  so = {k: v.clone() for k, v in sd.items()}
  for k in so:
    if so[k].dtype == t.float32 and "running" not in k: so[k].requires_grad_(True)
  lo = O.corenet_forward(so, image, v2s, off, training=True)
  O.iou_fgbg(grid, lo).backward()
  logits = m.train()(image.cuda(), v2s.cuda(), off.cuda())
  losses.iou_fgbg(grid.cuda(), logits).backward()
  e_train = relerr(logits, lo.detach())
  params = dict(m.named_parameters())
  e_g = {n: relerr(params[n].grad, so[n].grad) for n in ("decoder.stage_6.t1.weight", "decoder.rt_skip_5.compress_channels.weight",
                                                           "decoder.rt_skip_2.compress_channels.weight")}
  print(f"[{math} {hw}] eval logits {e_eval:.1e}, train logits {e_train:.1e}, gradients {', '.join(f'{k} {v:.1e}' for k, v in e_g.items())}")
  assert e_eval < 1e-4 and e_train < 1e-3, (e_eval, e_train)
  assert e_g["decoder.stage_6.t1.weight"] < 5e-4 and max(e_g.values()) < 2e-2, e_g


@pytest.mark.parametrize("hw", [(250, 256), (256, 30), (255, 255)])
def test_image_size_is_validated(hw):
  """H and W must be multiples of 4 and >= 32 (engine.check_image_hw: the stem's 2 x 2 space-to-depth view, the pooling cells):
  anything else raises ValueError on every entry point BEFORE a kernel runs or a buffer is allocated, and the model keeps working."""
  sd = O.make_state(0, 2, nbt=100)
  m = _model(2, sd)
  image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(2, 0, 2)]
  bad = t.randint(0, 256, (2, 3) + hw, dtype=t.uint8, device="cuda")
  for mode in ("train", "eval"):
    getattr(m, mode)()
    with pytest.raises(ValueError, match="multiples of 4"):
      m(bad, v2s, off)
  with pytest.raises(ValueError, match="multiples of 4"):
    m.train().train_step(bad, v2s, off, grid.to(t.int32))
  with pytest.raises(ValueError, match="multiples of 4"):
    m.eval().multi_offset_pmf(bad, v2s, off[None])
  with pytest.raises(ValueError, match="grid"):
    m.train().train_step(image, v2s, off, grid[:, :64].to(t.int32))
  with pytest.raises(ValueError):
    m.engine.plan(2).forward_encoder(bad, training=False)          # the plan itself refuses, not only the module
  assert list(m.engine.plans.keys()) == [2]


def test_forward_eval_b4_matches_oracle():
  """The bench batch size (B=4/GPU, h7.json5:42) end to end: eval-mode logits of the HIP path against the oracle on
  the same four images, 1e-4 relative (eval mode is conditioned at 2e-6), every voxel compared."""
  sd = O.make_state(0, 2, nbt=100)
  m = _model(2, sd).eval()
  image, v2s, off, grid = O.synthetic_batch(4, 0, 2)
  with t.no_grad():
    logits = m(image.cuda(), v2s.cuda(), off.cuda())
    want = O.corenet_forward({k: v.clone() for k, v in sd.items()}, image, v2s, off, training=False)
  assert logits.shape == (4, 2, 128, 128, 128)
  for b in range(4):
    assert relerr(logits[b], want[b]) < 1e-4, b
  assert float((logits[0] - logits[1]).abs().max()) > 0           # different images, not one sample four times


@pytest.mark.parametrize("nc", [2, 14])
def test_bf16x3_eval_b4_matches_oracle(nc):
  """The HEADLINE mode of bench.py (decoder_math="bf16x3": split-bf16 MFMA in decoder stages 3-6 and the encoder's 3x3
  layers) at the bench batch, against the oracle and not against this library's own fp32 mode: eval-mode logits, every
  voxel, 5e-5 relative (fp32 mode beside it: 2e-5), for the h7 (C=2) and the m7 / m9 (C=14) heads."""
  sd = O.make_state(0, nc, nbt=100)
  image, v2s, off, grid = O.synthetic_batch(4, 0, nc)
  with t.no_grad():
    want = O.corenet_forward({k: v.clone() for k, v in sd.items()}, image, v2s, off, training=False)
    errs = {}
    for math in ("bf16x3", "fp32"):
      m = _model(nc, sd, math).eval()
      logits = m(image.cuda(), v2s.cuda(), off.cuda())
      assert logits.shape == (4, nc, 128, 128, 128)
      errs[math] = max(relerr(logits[b], want[b]) for b in range(4))
      del m
  print(f"B=4 C={nc} eval logits vs oracle, every voxel: bf16x3 {errs['bf16x3']:.2e}, fp32 {errs['fp32']:.2e}")
  assert errs["bf16x3"] < 5e-5 and errs["fp32"] < 2e-5, errs        # (measured 2.1e-5 / 4.4e-6; the bars were 1e-4 until round 6)


GRAD_NOISE_FACTOR, GRAD_ERR_FLOOR, GRAD_BAR_CAP, GRAD_OUTLIER_CAP = 8.0, 2e-3, 0.1, 0.5   # see the docstring below
# the split-bf16 mode multiplies with 16 mantissa bits (two bf16 terms per operand, lo.lo dropped): its products carry
# 2^-16 where the reference's fp32 products carry 2^-24, and on these fixtures its gradients measure ~2x the noise of the
# fp32 mode (round 3).  Its bar is therefore twice the fp32 mode's -- still capped at 0.1 of a tensor's scale.
GRAD_MODE_FACTOR = {"fp32": 1.0, "bf16x3": 2.0}


@pytest.mark.parametrize("math", ["fp32", "bf16x3"])
@pytest.mark.parametrize("fixture,B", [("h7_train_b2_nbt30k", 2), ("h7_train_b4_nbt30k", 4)])
def test_all_parameter_gradients_against_fp64_truth(math, fixture, B):
  """EVERY parameter gradient, element by element, on the h7 B=2 and B=4 (the bench batch) fixtures with
  num_batches_tracked = 30000 (r/d clamps live).  The logits of these fixtures are well conditioned (the reference
  moves by 2e-6 between fp32 and fp64); the fixtures hold a fixed strided subsample (<= 512 elements) of each of the
  266 gradient tensors from the reference's fp32 run (gsub) AND from the same arithmetic in fp64 (g64sub, the truth;
  oracle/gen_golden.py).  The HIP path must be as close to the truth as the reference's own fp32 run is:

    error(tensor) = the (1 - 1/128) quantile of |got - truth| / scale over the stored elements,
    bar(tensor)   = min(f x max(8 x the SAME quantile of the reference's own fp32 error, 2e-3), 0.1),  f = 1 (fp32 mode) / 2 (bf16x3),

  scale = max(the tensor's own max, 1e-3 of the model's largest gradient) (conv biases in front of a train-mode norm
  have a true gradient of 0: for tensors on that floor the bar is at least 2e-2, i.e. 2e-5 of the largest gradient).  The quantile on BOTH sides, because one thing is inherent: a post-ReLU activation within
  rounding of 0 has its mask decided by the rounding order (this library applies the norm as x*scale+shift, the
  reference as ((x-mean)/std*r+d)*gamma+beta), and a flipped mask moves ONE element of a gradient by one term of its
  sum.  The reference's fp32 run has such elements too (up to 1e-1 of a tensor's scale, `gnoise` in the fixture); with
  them inside the bar, as in round 3, the bar of the deepest encoder tensors was 0.78 of their scale -- vacuous.  Now
  no bar exceeds 0.1, at most 1/128 of a tensor's stored elements (4 of 512; one element of a tensor with fewer than 128) may lie above it, they must still be
  within half the scale, and the last layer (whose gradient sees no ReLU mask) gets no such allowance at all."""
  from corenet_amd.model import losses
  z = np.load(os.path.join(G, f"model_{fixture}.npz"))
  m = _model(2, O.make_state(0, 2, nbt=30000), math).train()
  image, v2s, off, grid = O.synthetic_batch(B, 0, 2)
  loss = losses.iou_fgbg(grid.cuda(), m(image.cuda(), v2s.cuda(), off.cuda()))
  loss.backward()
  gmax = max(float(z[k]) for k in z.files if k.startswith("gmax::"))
  rows, outliers, bars, over = [], [], [], []
  for name, p in m.named_parameters():
    want = t.as_tensor(z["g64sub::" + name]).double()
    ref32 = t.as_tensor(z["gsub::" + name]).double()
    g = p.grad.reshape(-1)
    got = g[::max(1, -(-g.numel() // 512))].double().cpu()
    assert got.shape == want.shape, name
    scale = max(float(z["gmax::" + name]), 1e-3 * gmax)
    e = ((got - want).abs() / scale).sort().values
    eref = ((ref32 - want).abs() / scale).sort().values
    k = 0 if name.startswith("decoder.stage_6.t1.") else max(1, e.numel() // 128)    # elements that may be mask flips
    err, worst, noise = float(e[-(k + 1)]), float(e[-1]), float(eref[-(k + 1)])
    bar = min(GRAD_MODE_FACTOR[math] * max(GRAD_NOISE_FACTOR * noise, GRAD_ERR_FLOOR), GRAD_BAR_CAP)
    if float(z["gmax::" + name]) < 1e-3 * gmax:
      # the tensor's true gradient is (all but) zero -- conv biases in front of a train-mode norm -- and `scale` is the
      # floor, 1e-3 of the model's largest gradient: both sides hold summation noise only, bounded at 2e-5 of that gradient
      bar = max(bar, 2e-2)
    if int((e > bar).sum()) > k:
      over.append((name, int((e > bar).sum()), k, e.numel(), bar, [float(v) for v in e[-4:]], [float(v) for v in eref[-4:]]))
    rows.append((err / bar, err, noise, name)); bars.append(bar)
    outliers.append((worst, name))
  rows.sort(reverse=True); outliers.sort(reverse=True)
  errs = sorted(e for _, e, _, _ in rows)
  print(f"[{math} B={B}] {len(rows)} parameter gradients vs fp64 truth: median error {errs[len(errs) // 2]:.1e}, 90th "
        f"percentile {errs[len(errs) * 9 // 10]:.1e}, max {errs[-1]:.1e} of the tensor's scale; bars: median "
        f"{sorted(bars)[len(bars) // 2]:.1e}, max {max(bars):.1e}; closest to their bars (error / the reference's own fp32 "
        f"error at the same quantile): " + ", ".join(f"{n} {e:.1e}/{ns:.1e}" for _, e, ns, n in rows[:5]) +
        f"; largest single-element deviations: " + ", ".join(f"{n} {w:.1e}" for w, n in outliers[:3]))
  assert not over, over        # (tensor, elements above its bar, allowed, stored elements, bar, its / the reference's four largest errors)
  assert len(rows) == 266 and rows[0][0] <= 1.0, rows[:5]
  assert max(bars) <= GRAD_BAR_CAP and outliers[0][0] <= GRAD_OUTLIER_CAP, outliers[:3]


def decoder_gradient_bars(z, names, math):
  """Per-tensor element-wise bars of the decoder's weight gradients, of the tensor's scale: max(8 x the reference's own fp32-vs-fp64
  error on the fixture (`gnoise`), 2e-3), twice that for the split-bf16 mode (16 mantissa bits per product)."""
  return {n: GRAD_MODE_FACTOR[math] * max(GRAD_NOISE_FACTOR * float(z["gnoise::" + n]), GRAD_ERR_FLOOR) for n in names}


def decoder_weight_names(params):
  return [n for n in params if n.startswith("decoder.") and n.endswith(".weight") and
          (".c1." in n or ".t1." in n or "compress_channels" in n)]


@pytest.mark.parametrize("math", MATHS)
def test_decoder_gradients_every_element_vs_oracle(math):
  """EVERY element of every decoder convolution's weight gradient (stage_k.c1 / t1, rt_skip_k.compress_channels: 15 tensors,
  11.9 M elements) on the well-conditioned fixture's inputs (h7, B = 2, num_batches_tracked = 30000), against the oracle's
  autograd run here on the host (fp32; the fixture generator asserts oracle == reference on exactly these inputs).  Decoder
  tensors see no small-batch norm of the encoder, their conditioning is good (gnoise 2e-5 ... 6e-3 in the fixture), so the bar is
  per tensor max(8 x gnoise, 2e-3) of the tensor's scale on ALL elements -- no quantile, no sub-sample, no outlier allowance.
  This is the test the round-3/4 scatter glitch (a few hundred elements of the 64^3 skip-map gradient off by up to 6 % of its
  range) has to get past: tests/test_host_cpu.py::test_gradient_bars_catch_a_glitched_scatter injects exactly
  that fault into the oracle and asserts these bars flag it."""
  from corenet_amd.model import losses
  z = np.load(os.path.join(G, "model_h7_train_b2_nbt30k.npz"))
  sd = O.make_state(0, 2, nbt=30000)
  image, v2s, off, grid = O.synthetic_batch(2, 0, 2)
  m = _model(2, sd, math).train()
  losses.iou_fgbg(grid.cuda(), m(image.cuda(), v2s.cuda(), off.cuda())).backward()
  so = {k: v.clone() for k, v in sd.items()}
  for k in so:
    if so[k].dtype == t.float32 and "running" not in k: so[k].requires_grad_(True)
  O.iou_fgbg(grid, O.corenet_forward(so, image, v2s, off, training=True)).backward()
  params = dict(m.named_parameters())
  names = decoder_weight_names(params)
  assert len(names) == 15, names
  bars = decoder_gradient_bars(z, names, math)
  gmax = max(float(z[k]) for k in z.files if k.startswith("gmax::"))
  rows = []
  for n in names:
    got, want = params[n].grad.double().cpu(), so[n].grad.double()
    scale = max(float(want.abs().max()), 1e-3 * gmax)
    err = float((got - want).abs().max()) / scale
    rows.append((err / bars[n], err, bars[n], n, got.numel()))
  rows.sort(reverse=True)
  print(f"[{math}] decoder weight gradients, every element vs the oracle: worst " +
        ", ".join(f"{n} {e:.1e} (bar {b:.1e})" for _, e, b, n, _ in rows[:4]) + f"; {sum(r[4] for r in rows)} elements")
  assert rows[0][0] <= 1.0, rows[:4]


# element-wise bars of the five full gradients the fixtures store (oracle/gen_golden.py:87-90), by depth of the
# tensor below the loss: last layer / its norm / the 64^3 skip compression (through stage_6) / the latent bias
# (through the whole decoder; BatchRenorm over B=1 has zero gradient there).  The reference itself moves by
# 4e-5 ... 5e-2 between fp32 and fp64 on these inputs ("Conditioning").
FULL_GRAD_TOL = {"decoder.stage_6.t1.weight": 5e-4, "decoder.stage_6.b2.weight": 5e-4,
                 "decoder.rt_skip_5.compress_channels.weight": 2e-2, "decoder.stage_0.bias": 5e-2}
# measured (MI355X, round 2): 5.8e-5 / 3.0e-5 / 4.2e-3 / 1.6e-2 at worst over the three fixtures


@pytest.mark.parametrize("tag,nc,nbt,B,lossname", [("h7_train_b1", 2, 0, 1, "iou_fgbg"),
                                                   ("h7_train_b2_nbt30k", 2, 30000, 2, "iou_fgbg"),
                                                   ("m9_train_b1", 14, 0, 1, "xent_times_iou_agnostic")])
def test_train_forward_backward_golden(tag, nc, nbt, B, lossname):
  from corenet_amd.model import losses
  z = np.load(os.path.join(G, f"model_{tag}.npz"))
  sd = O.make_state(0, nc, nbt=nbt)
  m = _model(nc, sd).train()
  image, v2s, off, grid = O.synthetic_batch(B, 0, nc)
  logits = m(image.cuda(), v2s.cuda(), off.cuda())
  e = relerr(logits[:, :, ::16, ::16, ::16], z["logits_sub"])
  print(f"fp32 {tag} logits vs reference fixture: {e:.2e}")
  assert e < TRAIN_LOGIT_TOL(B), (tag, e)
  loss = getattr(losses, lossname)(grid.cuda(), logits)
  assert abs(float(loss) - float(z["loss"])) < 2e-4 * max(1.0, abs(float(z["loss"])))
  loss.backward()
  params = dict(m.named_parameters())
  # the five stored full gradients, element-wise
  for name, tol in FULL_GRAD_TOL.items():
    want = z["grad::" + name]
    if float(np.abs(want).max()) < 1e-12:
      assert float(params[name].grad.abs().max()) < 1e-6, name
      continue
    e = relerr(params[name].grad, want)
    print(f"[{tag}] grad {name}: max-abs-err/max = {e:.2e}")
    assert e < tol, (name, e)
  # a conv bias in front of a train-mode BatchRenorm has a true gradient of exactly 0: what both sides hold is
  # rounding noise, bounded against the scale of the weight gradient of the same conv
  noise = float(params["encoder.stage1.conv.bias"].grad.abs().max())
  scale = float(params["encoder.stage1.conv.weight"].grad.abs().max())
  assert noise <= 1e-2 * scale + 1e-12 and float(np.abs(z["grad::encoder.stage1.conv.bias"]).max()) <= 1e-2 * scale + 1e-12
  names, norms = list(z["grad_names"]), z["grad_norms"]
  bad = []
  for n, want in zip(names, norms):
    if n.endswith("conv.bias") or n.endswith("c1.bias") or ".t1.bias" in n and "stage_6" not in n:
      continue                     # biases in front of a train-mode BatchRenorm: true gradient is 0
    got = float(params[n].grad.double().norm())
    # B = 1 / nbt = 0 fixtures: statistics over ONE sample amplify rounding ~400x (the reference's own fp32 run is 5e-2 ... 2e-1 from
    # fp64 in the encoder there): 25 % on the norm is all those fixtures can hold.  The well-conditioned fixture (B = 2,
    # nbt = 30000) stores the reference's own fp32-vs-fp64 error per tensor (`gnoise`, of the tensor's scale): its norms are held to
    # 10 x max(8 x gnoise, 2e-3) -- |norm error| <= (max / rms of the tensor, < 10 measured) x the element-wise bar
    tol = 0.25
    if "gnoise::" + n in z.files:
      tol = min(0.25, 10.0 * max(8.0 * float(z["gnoise::" + n]), 2e-3))
    if abs(got - want) > tol * want + 1e-12:
      bad.append((n, got, float(want), tol))
  assert not bad, bad[:8]
  # running statistics were stepped like the reference's
  sdn = m.state_dict()
  for k in z.files:
    if k.startswith("buf::"):
      assert relerr(sdn[k[5:]], z[k]) < 1e-4, k
  assert int(sdn["decoder.stage_1.b1.num_batches_tracked"]) == nbt + 1


@pytest.mark.parametrize("math", MATHS)
def test_backward_matches_oracle_cosine(math):
  """Direction of every parameter gradient vs the oracle's autograd (B=1, h7)."""
  from corenet_amd.model import losses
  sd = O.make_state(0, 2, nbt=0)
  m = _model(2, sd, math).train()
  image, v2s, off, grid = O.synthetic_batch(1, 0, 2)
  loss = losses.iou_fgbg(grid.cuda(), m(image.cuda(), v2s.cuda(), off.cuda()))
  loss.backward()
  so = {k: v.clone() for k, v in sd.items()}
  for k in so:
    if so[k].dtype == t.float32 and "running" not in k: so[k].requires_grad_(True)
  O.iou_fgbg(grid, O.corenet_forward(so, image, v2s, off, training=True)).backward()
  worst = 1.0
  for n, p in m.named_parameters():
    if n.endswith("conv.bias") or n.endswith("c1.bias") or (".t1.bias" in n and "stage_6" not in n):
      continue
    a, b = p.grad.double().cpu().reshape(-1), so[n].grad.double().reshape(-1)
    if float(b.norm()) < 1e-12:          # e.g. stage_0 at B=1: BatchRenorm over one sample has zero gradient
      assert float(a.norm()) < 1e-6, n
      continue
    cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
    worst = min(worst, cos)
    assert cos > 0.995, (n, cos)        # measured: worst 0.9983 (bf16x3) / 0.9986 (fp32) at B = 1, nbt = 0 (round 5)
  print("worst gradient cosine vs oracle:", worst)


@pytest.mark.parametrize("math", MATHS)
def test_training_trajectory_matches_oracle(math):
  """Six consecutive training steps (forward, iou_fgbg, backward, Adam, BatchRenorm running statistics) of the
  HIP path next to the oracle driven by torch.optim.Adam, same init and sample: the loss curves stay together
  (measured 3e-3 relative at worst over 8 steps at B=1; small batches amplify rounding)."""
  sd = O.make_state(0, 2, nbt=0)
  m = _model(2, sd, math).train()
  image, v2s, off, grid = O.synthetic_batch(1, 0, 2)
  so = {k: v.clone() for k, v in sd.items()}
  params = []
  for k in so:
    if so[k].dtype == t.float32 and "running" not in k:
      so[k].requires_grad_(True); params.append(so[k])
  opt = t.optim.Adam(params, lr=4e-4, eps=1e-4)
  gi, gg = [x.cuda() for x in (image, v2s, off)], grid.cuda().to(t.int32)
  first = None
  for step in range(6):
    lg = float(m.train_step(gi[0], gi[1], gi[2], gg, "iou_fgbg", lr=4e-4, adam_eps=1e-4))
    opt.zero_grad()
    lo = O.iou_fgbg(grid, O.corenet_forward(so, image, v2s, off, training=True)); lo.backward(); opt.step()
    assert abs(lg - float(lo)) < 2e-2 * abs(float(lo)), (step, lg, float(lo))
    first = first if first is not None else lg
  assert lg < 0.9 * first                                  # and it learns
  assert relerr(m.state_dict()["decoder.stage_6.b1.running_mean"], so["decoder.stage_6.b1.running_mean"].detach()) < 2e-2


@pytest.mark.parametrize("math", MATHS)
def test_mean_iou_parity_on_trained_weights(math):
  """Mean-IoU parity (BASELINE metric) on weights that were actually trained: 80 HIP training steps on a fixed
  synthetic batch, then the eval-mode forward of the HIP path and of the oracle on those weights -- logits within the
  north_star tolerance, confusion matrices (fused argmax + histogram vs the oracle's) and mean IoU equal."""
  from corenet_amd import voxel_metrics as VM
  sd = O.make_state(0, 2, nbt=0)
  m = _model(2, sd, math).train()
  image, v2s, off, grid = O.synthetic_batch(2, 0, 2)
  gi, gg = [x.cuda() for x in (image, v2s, off)], grid.cuda().to(t.int32)
  for _ in range(80):
    m.train_step(gi[0], gi[1], gi[2], gg, "iou_fgbg", lr=4e-4, adam_eps=1e-4)
  m.eval()
  with t.no_grad():
    lh = m(*gi)
    so = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    lo = O.corenet_forward(so, image, v2s, off, training=False)
  assert relerr(lh, lo) < 1e-3
  _, cm = VM.argmax_confusion(lh, gg, 2)
  cmo = O.confusion_matrix(grid, O.extract_labels(lo), 2)
  iou_h, iou_o = VM.mean_iou(cm), O.mean_iou(cmo)
  assert int((cm.cpu() - cmo).abs().sum()) <= 8                                      # ties in the argmax at most
  # (80 steps at momentum 0.01 leave the running statistics far from the batch statistics, so the eval-mode IoU
  # itself is still low; the point is that both paths agree on it)
  assert abs(iou_h - iou_o) < 1e-4, (iou_h, iou_o)


@pytest.mark.parametrize("math", ["bf16x3", "fp32"])
def test_deterministic_mode_two_runs_bit_identical(math):
  """CRN_DETERMINISTIC / crn_set_deterministic: five training steps (forward, iou_fgbg, backward, Adam, running
  statistics) run twice from the same state give bit-identical parameters, Adam moments, buffers and losses -- weight
  gradients without position splits, ordered bias-gradient sums, fixed-point ray-sample scatter.  The default mode is
  run beside it (its atomics make two runs differ; printed, not asserted) and its losses must follow the deterministic
  trajectory while the trajectory is still conditioned (three steps, 2e-3): the switch changes the ORDER of the sums,
  nothing else."""
  from corenet_amd.backend import default_backend
  be = default_backend()
  sd = O.make_state(0, 2, nbt=0)
  image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(2, 0, 2)]
  gi = grid.to(t.int32)

  def run():
    m = _model(2, sd, math).train()
    ls = [float(m.train_step(image, v2s, off, gi, "iou_fgbg", lr=4e-4, adam_eps=1e-4)) for _ in range(5)]
    t.cuda.synchronize()
    e = m.engine
    return ls, [x.clone() for x in (e.store.params, e.store.buffers, e.adam_m, e.adam_v, e.store.grads)]

  try:
    be.set_deterministic(True)
    la, ta = run()
    lb, tb = run()
  finally:
    be.set_deterministic(False)
  assert la == lb, (la, lb)
  for name, a, b in zip(("params", "buffers", "adam_m", "adam_v", "grads"), ta, tb):
    assert t.equal(a, b), (name, float((a - b).abs().max()))
  lc, tc = run()
  ld, td = run()
  print(f"[{math}] deterministic: 5 steps twice, bit-identical (loss {la[-1]:.6f}); default mode: run-to-run max |d param| "
        f"{float((tc[0] - td[0]).abs().max()):.2e}, vs deterministic {float((tc[0] - ta[0]).abs().max()):.2e}")
  # (Adam moves every parameter by ~lr per step whatever the size of its gradient, so last-bit differences in tiny
  # gradients become 1e-3 differences in parameters within a few steps -- and this fixture, BatchRenorm over two samples
  # with num_batches_tracked = 0, is chaotic from there on: 24 default-mode runs from the same state gave third losses
  # within 5e-4 of each other, fourth within 6e-3, fifth between 0.778 and 0.830.  The first three steps are the check.)
  for i in range(3):
    assert abs(lc[i] - la[i]) < 2e-3 * abs(la[i]), (i, lc, la)
  assert abs(lc[-1] - la[-1]) < 0.15 * abs(la[-1]), (lc, la)


def test_native_rccl_from_the_library_single_rank():
  """crn_comm_* / crn_allreduce_f32 (include/corenet_hip.h): the library's own RCCL communicator, bootstrapped from a
  unique id, on ONE rank (all this box can host): the in-place all-reduce is the identity and stream-ordered, and the
  fused train step driven through GradientSync(native=True) -- all-reduces enqueued on the engine's side stream
  behind each bucket's un-pack, rank 0's BatchRenorm buffers riding on the first bucket -- lands bit for bit on the
  parameters and buffers of the step without an exchange."""
  from corenet_amd import distributed as D
  comm = D.NativeComm(0, 1)
  assert comm.version > 0
  x = t.randn(1 << 20, device="cuda"); y = x.clone()
  comm.all_reduce(x)
  t.cuda.synchronize()
  assert t.equal(x, y)
  comm.close()
  sd = O.make_state(0, 2, nbt=0)
  image, v2s, off, grid = [v.cuda() for v in O.synthetic_batch(1, 0, 2)]
  gi = grid.to(t.int32)
  ma, mb = _model(2, sd, "bf16x3").train(), _model(2, sd, "bf16x3").train()
  be = ma.engine.be
  sync = D.GradientSync(1, force=True, native=True).attach(ma.engine)
  assert sync.native is not None and sync.describe()["transport"].startswith("native RCCL")
  try:
    be.set_deterministic(True)                     # (so that the two replicas can be compared bit for bit)
    for _ in range(2):
      la = ma.train_step(image, v2s, off, gi, "iou_fgbg", lr=4e-4, adam_eps=1e-4, all_reduce=sync)
      lb = mb.train_step(image, v2s, off, gi, "iou_fgbg", lr=4e-4, adam_eps=1e-4)
    t.cuda.synchronize()
  finally:
    be.set_deterministic(False)
  assert len(sync.pushed) == len(ma.engine.grad_buckets) == 8 and float(la) == float(lb)
  assert t.equal(ma.engine.store.params, mb.engine.store.params) and t.equal(ma.engine.store.buffers, mb.engine.store.buffers)
  sync.native.close()


def test_inactive_gradient_sync_is_the_step_without_an_exchange():
  """An ATTACHED exchange object with nothing to exchange (world_size 1, no `force`: what a single-GPU run of a
  multi-GPU script constructs) must drive exactly the step without an exchange.  Round 3 sent it down the bucket-hook
  path, whose inactive branch ran the per-bucket Adam on the side stream while the step's Adam scalars (and, on the
  first step, the zero-fill of the moments) were still in flight on the optimizer stream.  Deterministic
  mode, three steps from the same state: parameters, moments and buffers bit-identical to all_reduce=None."""
  from corenet_amd import distributed as D
  from corenet_amd.backend import default_backend
  be = default_backend()
  sd = O.make_state(0, 2, nbt=0)
  image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(2, 0, 2)]
  gi = grid.to(t.int32)
  try:
    be.set_deterministic(True)
    outs = []
    for use_sync in (False, True):
      m = _model(2, sd, "bf16x3").train()
      sync = D.GradientSync(1).attach(m.engine) if use_sync else None
      assert sync is None or not sync._active()
      ls = [float(m.train_step(image, v2s, off, gi, "iou_fgbg", lr=4e-4, adam_eps=1e-4, all_reduce=sync)) for _ in range(3)]
      t.cuda.synchronize()
      e = m.engine
      outs.append((ls, [x.clone() for x in (e.store.params, e.store.buffers, e.adam_m, e.adam_v)]))
      assert e.adam_t == 3
  finally:
    be.set_deterministic(False)
  assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
  for name, a, b in zip(("params", "buffers", "adam_m", "adam_v"), outs[0][1], outs[1][1]):
    assert t.equal(a, b), (name, float((a - b).abs().max()))


def _sync_state(dst, src):
  """dst becomes an exact replica of src (parameters, buffers, step counters, Adam moments)."""
  de, se = dst.engine, src.engine
  for a, b in ((de.store.params, se.store.params), (de.store.buffers, se.store.buffers), (de.store.nbt, se.store.nbt)):
    a.copy_(b)
  if se.adam_m is not None:
    if de.adam_m is None:
      de.adam_m, de.adam_v = t.zeros_like(se.adam_m), t.zeros_like(se.adam_v)
    de.adam_m.copy_(se.adam_m); de.adam_v.copy_(se.adam_v)
  de.adam_t = se.adam_t
  de.weights_dirty = True


def _slab_err(ma, m):
  """worst max-abs-err / max over the gradient buckets of two engines (two runs of the SAME step differ by the
  summation order of the weight-gradient atomics, measured <= 2e-3; a doubled gradient is 1.0)."""
  worst = 0.0
  for _, lo, hi in m.engine.grad_buckets:
    worst = max(worst, relerr(ma.engine.store.grads[lo:hi], m.engine.store.grads[lo:hi]))
  return worst


@pytest.mark.parametrize("math", MATHS)
def test_train_step_reduces_loss_and_matches_autograd_path(math):
  """The reference's loop body (pipeline.py:224-230: optimizer.zero_grad(); loss = f(model(...)); loss.backward();
  optimizer.step()) through the drop-in's autograd node + FusedAdam against the fused `train_step`, THREE steps:
  same losses, same parameters, same Adam moments (a `.grad` that aliases the engine's gradient slab doubles every
  gradient from step 2 on: ADVICE r1), no memory growth per step, gradient accumulation adds up, and a backward
  whose forward was overwritten raises."""
  from corenet_amd import state as S
  from corenet_amd.model import losses
  sd = O.make_state(0, 2, nbt=0)
  m, ma = _model(2, sd, math).train(), _model(2, sd, math).train()
  image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(2, 0, 2)]
  opt = S.FusedAdam(ma, lr=4e-4, eps=1e-4)
  ls = []
  # Both paths launch the same kernels; in the default mode the weight-gradient and ray-sample atomics arrive in another
  # order every run, and BatchRenorm over B = 2 at nbt = 0 amplifies that noise to 1e-3 ... 6e-3 of the slab in the bf16x3
  # mode (measured, round 4: also between two runs of the SAME path).  The comparison therefore runs in deterministic mode
  # (crn_set_deterministic: ordered sums), where the two paths have to agree to the last bit of summation order.
  be = m.engine.be
  try:
    be.set_deterministic(True)
    for step in range(3):
      _sync_state(ma, m)               # same state in: the step itself is what is compared (training at B=2 is
      ls.append(float(m.train_step(image, v2s, off, grid, "iou_fgbg", lr=4e-4, adam_eps=1e-4)))    # chaotic over steps)
      opt.zero_grad()
      loss = losses.iou_fgbg(grid, ma(image, v2s, off))
      loss.backward()
      opt.step()
      assert abs(float(loss) - ls[-1]) < 1e-6 * abs(ls[-1]), (step, float(loss), ls[-1])
      e = _slab_err(ma, m)
      print(f"autograd path vs train_step ({math}, deterministic sums), step {step}: gradient slab err {e:.2e}")
      assert e < 1e-6, (step, e)
      assert ma.engine.adam_t == m.engine.adam_t == step + 1
      if step == 0:
        t.cuda.synchronize(); mem0 = t.cuda.memory_allocated()
    t.cuda.synchronize()
  finally:
    be.set_deterministic(False)
  assert t.cuda.memory_allocated() <= mem0 + (1 << 20), (mem0, t.cuda.memory_allocated())
  assert np.isfinite(ls[-1]) and ls[-1] < ls[0]
  assert relerr(ma.engine.adam_m, m.engine.adam_m) < 1e-2
  g = ma.get_parameter("decoder.stage_6.t1.weight")
  assert g.grad.data_ptr() != ma.engine.store.view("decoder.stage_6.t1.weight", grad=True).data_ptr()
  # accumulation: backward twice without zero_grad -> g1 + g2 (the same sample twice: 2 * g up to atomics order)
  opt.zero_grad(set_to_none=True)
  losses.iou_fgbg(grid, ma(image, v2s, off)).backward()
  g1 = g.grad.clone()
  losses.iou_fgbg(grid, ma(image, v2s, off)).backward()
  assert relerr(g.grad, 2 * g1) < 2e-3
  # torch.optim.Adam on the same parameters works too (the parameters are views of the engine's slab)
  topt = t.optim.Adam(ma.parameters(), lr=4e-4, eps=1e-4)
  topt.zero_grad()
  before = ma.engine.store.params.clone()
  losses.iou_fgbg(grid, ma(image, v2s, off)).backward()
  topt.step()
  assert float((ma.engine.store.params - before).abs().max()) > 1e-5
  l1 = losses.iou_fgbg(grid, ma(image, v2s, off))
  with t.no_grad():
    ma(image, v2s, off)
  with pytest.raises(RuntimeError, match="overwritten"):
    l1.backward()


def test_inference_forward_graph_replay_follows_the_weights():
  """Inference forwards (eval mode, no_grad) replay a captured HIP graph from the third call of a batch size on
  (CoreNet._forward_eval_graph) and re-derive the packed weights only when the parameter / buffer slabs were written:
  replayed logits equal the launch-by-launch forward bit for bit, for new inputs too, and follow the weights after an in-place
  torch update of a parameter, a load_state_dict and a fused training step; another batch size gets its own graph."""
  sd = O.make_state(0, 2, nbt=100)
  m = _model(2, sd).eval()
  batches = [[x.cuda() for x in O.synthetic_batch(2, s, 2)[:3]] for s in (0, 1)]
  def eager(b):
    m.engine.weights_dirty = True
    return m.engine.plan(2).forward(*b, training=False).clone()
  with t.no_grad():
    outs = [m(*batches[0]) for _ in range(4)]
    plan = m.engine.plan(2)
    assert plan.eval_graph is not None and plan.eval_eager == 2           # two launch-by-launch calls, then replays
    assert all(t.equal(o, outs[0]) for o in outs[1:])
    assert t.equal(m(*batches[1]), eager(batches[1]))                     # new inputs through the replay
    assert not t.equal(m(*batches[1]), outs[0])
    # (i) an in-place torch write to a parameter (what an optimizer does) is seen through the version counter
    p = m.get_parameter("decoder.stage_6.t1.weight")
    p.mul_(1.5)
    got = m(*batches[0])
    assert t.equal(got, eager(batches[0])) and not t.equal(got, outs[0])
    assert t.equal(m(*batches[0]), got)                                   # (replay again, same weights)
    # (ii) load_state_dict
    m.load_state_dict(O.make_state(1, 2, nbt=100))
    got = m(*batches[0])
    assert t.equal(got, eager(batches[0])) and not t.equal(got, outs[0])
    # (iii) another batch size has its own plan and graph
    b1 = [x.cuda() for x in O.synthetic_batch(1, 0, 2)[:3]]
    o1 = [m(*b1) for _ in range(3)]
    assert m.engine.plan(1).eval_graph is not None and t.equal(o1[0], o1[2])
    assert relerr(o1[2], got[:1]) < 1e-5            # (the same sample at another batch size: other tiles and splits, same math)
  # (iv) a fused training step in between: the next inference forward sees the stepped weights and the new running statistics
  image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(2, 0, 2)]
  m.train(); m.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", lr=1e-2, adam_eps=1e-4); m.eval()
  with t.no_grad():
    got2 = m(*batches[0])
    assert t.equal(got2, eager(batches[0])) and not t.equal(got2, got)
    assert t.equal(m(*batches[0]), got2)


@pytest.mark.parametrize("math", MATHS)
@pytest.mark.parametrize("B,nbt", [(2, 0), (4, 30000)])
def test_run_to_run_gradient_spread_default_mode(math, B, nbt):
  """Default (non-deterministic) mode: the same training step from the same state on two model instances may differ only by
  the order in which atomics arrive -- 1e-6 ... 1e-5 of a gradient bucket's range (tools/run_noise.py).  Round 4 found the
  ray-sample scatter of the 64^3 skip, launched on the side stream the moment its input was complete, returning sums that
  differed by 1e-2 from run to run (decoder.rt_skip_5's weight gradient: 1e-3 ... 7e-3 of its bucket) while every golden
  test still passed; the bar here is 1e-4 per bucket, and identical logits and losses."""
  sd = O.make_state(0, 2, nbt=nbt)
  image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(B, 0, 2)]
  gi = grid.to(t.int32)
  ms = [_model(2, sd, math).train() for _ in range(2)]
  for rep in range(2):
    ls = [float(m.train_step(image, v2s, off, gi, "iou_fgbg", lr=0.0, adam_eps=1e-4)) for m in ms]
    t.cuda.synchronize()
    assert ls[0] == ls[1], ls
    assert t.equal(ms[0].engine.plan(B).logits, ms[1].engine.plan(B).logits)
    g1, g2 = ms[0].engine.store.grads, ms[1].engine.store.grads
    per = {lb or "stem..stage3": float((g1[lo:hi] - g2[lo:hi]).abs().max() / g1[lo:hi].abs().max())
           for lb, lo, hi in ms[0].engine.grad_buckets}
    print(f"[{math} B={B} nbt={nbt}] run-to-run spread per gradient bucket: " + ", ".join(f"{k} {v:.1e}" for k, v in per.items()))
    assert max(per.values()) < 1e-4, per


@pytest.mark.parametrize("math", MATHS)
def test_train_step_hip_graph_replay_matches_launch_by_launch(math):
  """CoreNet.train_step(graph=True): the fused step captured into a HIP graph (inputs in plan-owned buffers, Adam's
  scalars in device memory) and replayed.  From identical states, every replayed step must equal the launch-by-launch
  step: loss, gradient slab (up to the atomics' summation order), Adam moments, step counters -- including a change of
  inputs and of the learning rate between replays."""
  sd = O.make_state(0, 2, nbt=0)
  m, mg = _model(2, sd, math).train(), _model(2, sd, math).train()
  batches = [[x.cuda() for x in O.synthetic_batch(2, s, 2)] for s in (0, 1)]
  for step in range(5):
    image, v2s, off, grid = batches[step % 2]
    lr = 4e-4 if step < 3 else 1e-4
    _sync_state(mg, m)
    la = float(m.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", lr=lr, adam_eps=1e-4, graph=False))
    lb = float(mg.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", lr=lr, adam_eps=1e-4, graph=True))
    assert abs(la - lb) < 1e-4 * abs(la), (step, la, lb)
    assert _slab_err(mg, m) < 1e-3, (step, _slab_err(mg, m))
    assert mg.engine.adam_t == m.engine.adam_t == step + 1
    d = (mg.engine.store.params - m.engine.store.params).abs()
    assert float(d.max()) <= 2 * lr * 1.05 and float(d.mean()) < 0.05 * lr, (step, float(d.max()), float(d.mean()))
    assert t.equal(mg.engine.store.nbt, m.engine.store.nbt)
  assert len(mg.engine.plan(2).graphs) == 1 and not m.engine.plan(2).graphs
  # an eval forward after replays sees the stepped parameters
  mg.eval(); m.eval()
  _sync_state(mg, m)
  with t.no_grad():
    assert relerr(mg(*batches[0][:3]), m(*batches[0][:3])) < 1e-5


def test_checkpoint_interop_on_gpu():
  """N4 (state.py:74-97) on the device: (i) a checkpoint in the reference's format -- model_state +
  torch.optim.Adam.state_dict() after two Adam steps, built here with torch.optim.Adam on the CPU like the reference
  would have saved it -- decodes into the HIP model: same eval logits as the oracle on those weights, and the next
  fused Adam step equals torch.optim.Adam's on the same gradients; (ii) encode_state -> decode_state of a model
  trained two HIP steps resumes bit-identically (parameters, buffers, moments, step count) and the resumed model
  takes the same third step."""
  import io
  from corenet_amd import state as S
  sd = O.make_state(3, 2, nbt=7)
  keys = list(sd.keys())
  pkeys = [k for k in keys if sd[k].dtype == t.float32 and "running" not in k]
  g = t.Generator().manual_seed(11)
  params = [sd[k].clone().requires_grad_(True) for k in pkeys]
  opt = t.optim.Adam(params, lr=4e-4, eps=1e-4)
  for _ in range(2):
    for p in params: p.grad = t.randn(p.shape, generator=g) * 0.01
    opt.step()
  model_state = {k: (params[pkeys.index(k)].detach().clone() if k in pkeys else sd[k].clone()) for k in keys}
  cfg = {"decoder": {"resolution": (128, 128, 128), "num_output_channels": 2, "last_upscale_factor": 2,
                     "latent_channels": 64, "skip_fraction": 0.75}}
  buf = io.BytesIO()
  t.save({"global_step": 77, "model_state": model_state, "model_config": cfg, "optimizer_state": opt.state_dict(),
          "extra_metadata": None}, buf)
  st = S.decode_state(buf.getvalue(), "cuda")
  assert st.global_step == 77 and st.model.engine.adam_t == 2 and st.model.engine.device.type == "cuda"
  got = st.model.state_dict()
  assert all(t.equal(got[k].cpu(), model_state[k]) for k in keys)
  image, v2s, off, grid = O.synthetic_batch(1, 0, 2)
  st.model.eval()
  with t.no_grad():
    lh = st.model(image.cuda(), v2s.cuda(), off.cuda())
    lo = O.corenet_forward({k: v.clone() for k, v in model_state.items()}, image, v2s, off, training=False)
  assert relerr(lh, lo) < 1e-4
  grads = [t.randn(p.shape, generator=g) * 0.01 for p in params]
  for p, gr in zip(params, grads): p.grad = gr
  opt.step()
  for k, gr in zip(pkeys, grads): st.model.engine.store.view(k, grad=True).copy_(gr)
  st.optimizer.step()
  for k, p in zip(pkeys, params):
    assert float((st.model.state_dict()[k].cpu() - p.detach()).abs().max()) < 1e-6, k
  # (ii) our own checkpoint, written from the GPU
  m = _model(2, O.make_state(0, 2, nbt=0), "bf16x3").train()      # (the product default, which decode_state builds too)
  o1 = S.FusedAdam(m, lr=4e-4, eps=1e-4)
  gi = [x.cuda() for x in (image, v2s, off)]; gg = grid.cuda().to(t.int32)
  for _ in range(2):
    m.train_step(gi[0], gi[1], gi[2], gg, "iou_fgbg", lr=4e-4, adam_eps=1e-4)
  raw = S.encode_state(S.State(global_step=2, model=m, optimizer=o1, extra_metadata={"k": 1}))
  st2 = S.decode_state(raw, "cuda")
  m2 = st2.model.train()
  assert st2.extra_metadata == {"k": 1} and m2.engine.adam_t == 2
  assert t.equal(m2.engine.store.params, m.engine.store.params) and t.equal(m2.engine.store.buffers, m.engine.store.buffers)
  assert t.equal(m2.engine.store.nbt, m.engine.store.nbt)
  assert t.equal(m2.engine.adam_m, m.engine.adam_m) and t.equal(m2.engine.adam_v, m.engine.adam_v)
  assert st2.optimizer.param_groups[0]["lr"] == 4e-4 and st2.optimizer.param_groups[0]["eps"] == 1e-4
  la = float(m.train_step(gi[0], gi[1], gi[2], gg, "iou_fgbg", lr=4e-4, adam_eps=1e-4))
  lb = float(m2.train_step(gi[0], gi[1], gi[2], gg, "iou_fgbg", lr=4e-4, adam_eps=1e-4))
  assert abs(la - lb) < 1e-4 * abs(la) and _slab_err(m2, m) < 1e-2


@pytest.mark.parametrize("math", MATHS)
def test_wrapped_in_distributed_data_parallel(math):
  """pipeline.py:199-200,224-230 unmodified: `DistributedDataParallel(model, device_ids=[dev])`, then
  `loss = f(ddp(...))`, `loss.backward()`, `optimizer.step()` -- world 1 over RCCL.  DDP broadcasts parameters and
  buffers from rank 0 (in place: they stay views of the engine's slabs), hooks every parameter's gradient
  accumulator and all-reduces its buckets; with one rank that is the identity, so losses and parameters must follow
  the fused train_step."""
  import torch.distributed as dist
  from torch.nn.parallel import DistributedDataParallel
  from corenet_amd import state as S
  from corenet_amd.model import losses
  dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29850 + os.getpid() % 100 + 1000 * MATHS.index(math)}", rank=0, world_size=1)
  try:
    sd = O.make_state(0, 2, nbt=0)
    m, ma = _model(2, sd, math).train(), _model(2, sd, math).train()
    image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(2, 0, 2)]
    ddp = DistributedDataParallel(ma, device_ids=[t.cuda.current_device()])
    assert ma.get_parameter("decoder.stage_6.t1.weight").data_ptr() == \
        ma.engine.store.view("decoder.stage_6.t1.weight").data_ptr()           # still views of the slab
    opt = S.FusedAdam(ma, lr=4e-4, eps=1e-4)
    for step in range(3):
      _sync_state(ma, m)
      lf = float(m.train_step(image, v2s, off, grid, "iou_fgbg", lr=4e-4, adam_eps=1e-4))
      opt.zero_grad()
      loss = losses.iou_fgbg(grid, ddp(image, v2s, off))
      loss.backward()
      opt.step()
      assert abs(float(loss) - lf) < 1e-4 * abs(lf), (step, float(loss), lf)
      e = _slab_err(ma, m)
      print(f"DDP-wrapped vs train_step, step {step}: gradient slab err {e:.2e}")
      assert e < 1e-3, (step, e)                    # measured 1.5e-6
    assert relerr(ma.engine.adam_m, m.engine.adam_m) < 1e-2
    d = (ma.engine.store.params - m.engine.store.params).abs()
    assert float(d.max()) <= 2 * 4e-4 * 1.05 and float(d.mean()) < 1e-4
    assert int(ma.state_dict()["decoder.stage_1.b1.num_batches_tracked"]) == 3
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("math", MATHS)
def test_overlapped_gradient_exchange_single_rank_rccl(math):
  """The bucketed backward + RCCL all-reduce on its own stream (distributed.GradientSync.push/wait,
  engine.GRAD_BUCKET_LABELS) against the plain one-slab backward: with one rank the all-reduce is the
  identity, so gradients and parameters must agree up to the atomics' summation order."""
  import torch.distributed as dist
  from corenet_amd import distributed as D
  dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29650 + os.getpid() % 200 + 1000 * MATHS.index(math)}", rank=0, world_size=1)
  try:
    sd = O.make_state(0, 2, nbt=0)
    image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(2, 0, 2)]
    ma, mb = _model(2, sd, math).train(), _model(2, sd, math).train()
    sync = D.GradientSync(1, force=True)
    assert sync.overlap
    for i in range(3):
      la = ma.train_step(image, v2s, off, grid, "iou_fgbg", lr=4e-4)
      lb = mb.train_step(image, v2s, off, grid, "iou_fgbg", lr=4e-4, all_reduce=sync)
      if i == 0:
        ga, gb = ma.engine.store.grads, mb.engine.store.grads
        assert sum(sync.pushed) == ga.numel() and len(sync.pushed) == len(mb.engine.grad_buckets)
        for _, lo, hi in mb.engine.grad_buckets:
          # two runs of the same step differ by the summation order of the weight-gradient atomics
          assert relerr(gb[lo:hi], ga[lo:hi]) < 2e-3, (lo, hi)
    t.cuda.synchronize()
    assert abs(float(la) - float(lb)) < 3e-3                                       # measured: 0.7-3e-4
    d = (mb.engine.store.params - ma.engine.store.params).abs()
    # Adam normalises every gradient to +-lr, so run-to-run noise on near-zero gradients moves those parameters
    # by up to 2*lr per step: bound the distance, do not count how many moved
    assert float(d.max()) <= 3 * 2 * 4e-4 * 1.05 and float(d.mean()) < 3e-4        # measured: mean 2-3e-5
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize("math", MATHS)
def test_pipeline_process_batch_from_dataset(math):
  """dataset elements -> batch (GPU) -> ground-truth voxelization -> v2s -> train step (pipeline.process_batch,
  reference TrainPipeline._process_batch pipeline.py:215-242): runs end to end on the fixture dataset (images
  replaced by 256x256 ones through the dataset's data_transforms hook), the loss is finite and equals the loss of
  the same step driven by hand from the batch's tensors."""
  import dataclasses
  from corenet_amd import pipeline
  from corenet_amd.data import batched_example as B, dataset as D
  root = os.path.join(G, "n2_dataset")
  g = t.Generator().manual_seed(5)
  def big_image(scene, el):
    return dataclasses.replace(el, input_image=t.randint(0, 256, (3, 256, 256), generator=g, dtype=t.uint8))
  ds = D.CoReNetDatasetImpl(os.path.join(root, "dataset.json"), os.path.join(root, "meshes"), data_transforms=[big_image])
  els = [ds[0], ds[1]]
  sd = O.make_state(0, 4, nbt=0)
  for task, nc in (("semantic", 4), ("fg_bg", 2)):
    sd = O.make_state(0, nc, nbt=0)
    ma, mb = _model(nc, sd, math).train(), _model(nc, sd, math).train()
    la = pipeline.process_batch(ma, els, task)
    ex = pipeline.voxelize_batch(B.batch(els), task)
    assert ex.grid.shape == (2, 128, 128, 128) and int(ex.grid.max()) == (3 if task == "semantic" else 1)
    v2s = ex.camera_transform @ t.diag(t.tensor([1 / 128.0] * 3 + [1.0])).cuda()
    lb = mb.train_step(ex.input_image, v2s, ex.grid_sampling_offset, ex.grid, pipeline.LOSS_OF_TASK[task])
    assert np.isfinite(float(la)) and abs(float(la) - float(lb)) < 1e-4
    pmf, ex2 = pipeline.evaluate_batch(lambda im, cam, v2x, off, res: ma(im, cam @ v2x.cpu().inverse().cuda(), off).softmax(1),
                                       els, task)
    assert pmf.shape == (2, nc, 128, 128, 128) and t.equal(ex2.grid, ex.grid)


@pytest.mark.parametrize("math", MATHS)
def test_super_resolution_x2_golden_and_encoder_reuse(math):
  """N1: x2 super-resolution through the drop-in of corenet.super_resolution.
  (i) golden pmf generated by the reference's SuperResolutionInference (oracle/gen_golden.py);
  (ii) sub-grid (iz,iy,ix) of the result == softmax of an ordinary eval forward at that offset:
       running the encoder once instead of 8 times changes nothing beyond run-to-run rounding."""
  from corenet_amd import super_resolution as SR
  z = np.load(os.path.join(G, "super_resolution_h7_x2.npz"))
  m = _model(2, O.make_state(0, 2, nbt=100, logit_scale=2e-4), math).eval()
  image, v2s, off, _ = O.synthetic_batch(1, 0, 2)
  camera = O.canonical_camera()[None].cuda()
  v2v = O.scale([128.0] * 3)[None].cuda()
  go = t.full((1, 3), 0.5).cuda()

  class State: pass
  st = State(); st.model = m
  sr = SR.super_resolution_from_state(st)
  assert sr.resolution == (128, 128, 128)
  native = sr.get_native_offsets((256, 256, 256), go)
  np.testing.assert_allclose(native.cpu().numpy(), z["native_offsets"], rtol=0, atol=0)
  pmf = sr(image.cuda(), camera, v2v, go, (256, 256, 256))
  assert pmf.shape == (1, 2, 256, 256, 256) and pmf.dtype == t.float32
  # the fixture's last layer is scaled (make_state(logit_scale=2e-4)) so that the logits are O(1) and the pmf is a
  # well-conditioned function of them: |d pmf| <= |d logit| / 4, eval logits agree to ~1e-5 relative
  def pmf_close(got, want, what):
    d = (got.cpu() - t.as_tensor(want).cpu()).abs()
    assert float(d.max()) < 1e-4 and float(d.mean()) < 1e-5, (what, float(d.max()), float(d.mean()))
  pmf_close(pmf[:, :, ::16, ::16, ::16], z["pmf_sub"], "golden even")
  pmf_close(pmf[:, :, 1::32, 1::32, 1::32], z["pmf_odd"], "golden odd")
  assert abs(float(pmf.double().sum()) - float(z["pmf_sum"])) < 1e-6 * float(z["pmf_sum"])
  assert abs(float(pmf[:, 1].double().sum()) - float(z["fg_sum"])) < 1e-4 * float(z["pmf_sum"])
  # the generic MultiOffsetInferenceFn path (reshape / permute on the host side) gives the same grid
  generic = SR.SuperResolutionInference(lambda im, cam, vv, offs: SR.CoreNetMultiOffset(m)(im, cam, vv, offs),
                                        (128, 128, 128))
  # (two runs of the network differ in the last bits: split-K partial sums are added with atomics)
  pmf_close(generic(image.cuda(), camera, v2v, go, (256, 256, 256)), pmf, "generic path")
  # encoder reuse == full forward per offset
  vs = (camera @ (v2v @ O.scale([0.5] * 3).cuda()).inverse())
  with t.no_grad():
    for n in (0, 3, 6, 7):
      iz, iy, ix = n // 4, (n // 2) % 2, n % 2
      logits = m(image.cuda(), vs, native[n])
      one = SR.CoreNetMultiOffset(m)(image.cuda(), camera, v2v @ O.scale([0.5] * 3).cuda(), native[n:n + 1])[0]
      pmf_close(pmf[:, :, iz::2, iy::2, ix::2], one, "single offset")
      pmf_close(pmf[:, :, iz::2, iy::2, ix::2], logits.softmax(1), "full forward")
  with pytest.raises(ValueError):
    sr(image.cuda(), camera, v2v, go, (192, 256, 256))
  m.train()
  with pytest.raises(AssertionError):
    sr(image.cuda(), camera, v2v, go, (256, 256, 256))
