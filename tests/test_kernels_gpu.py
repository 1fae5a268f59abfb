"""GPU parity tests: every HIP kernel (through the C ABI) against its contract
(tests/kernel_contract_emu.py) / the oracle on the same seeded inputs.
Integer/index work must be bit-exact; fp32 work within the stated tolerance."""
import numpy as np
import pytest
import torch as t

from oracle import corenet_oracle as O
from kernel_contract_emu import EmuBackend

pytestmark = pytest.mark.gpu


import os
EMU = EmuBackend()
# CRN_TEST_SELFCHECK=1 runs the test LOGIC against the contract emulator on CPU
# (a dry run of the tests themselves in the GPU-less container; never on the GPU box).
_SELF = os.environ.get("CRN_TEST_SELFCHECK") == "1"
DEV = "cpu" if _SELF else "cuda"


@pytest.fixture(scope="module")
def be():
  if _SELF:
    return EMU
  from corenet_amd.backend import HipBackend
  assert t.cuda.is_available(), "gpu tests need an MI355X"
  return HipBackend()


def close(a, b, rtol, name=""):
  a = a.detach().cpu().double(); b = b.detach().cpu().double()
  den = float(b.abs().max()) + 1e-30
  err = float((a - b).abs().max()) / den
  assert err <= rtol, f"{name}: max-abs-err/max = {err:.3e} > {rtol}"


# ------------------------------------------------------------------ conv engine
def _views(xc, make_view):
  """Build the same view over the CPU tensor and over its CUDA copy."""
  xg = xc.to(DEV)
  return make_view(xc), make_view(xg), xg


CONV_CASES = [
    # name, kind, wshape, pad, in dims (D,H,W), batch
    ("conv1x1", "conv", (64, 256, 1, 1), 0, (1, 64, 64), 2),
    ("conv3x3", "conv", (128, 128, 3, 3), 1, (1, 32, 32), 2),
    ("conv3x3_small", "conv", (512, 512, 3, 3), 1, (1, 8, 8), 2),
    ("conv1x1_odd", "conv", (24, 515, 1, 1), 0, (1, 32, 32), 2),
    ("conv1x1_expand", "conv", (256, 64, 1, 1), 0, (1, 64, 64), 4),      # one K chunk per tile: many units per workgroup
    ("conv1x1_splitk", "conv", (256, 1024, 1, 1), 0, (1, 16, 16), 4),     # long K, few positions: K splits
    ("conv1x1_wide", "conv", (2048, 512, 1, 1), 0, (1, 8, 8), 4),
    ("conv1x1_ragged", "conv", (40, 72, 1, 1), 0, (1, 12, 12), 3),        # partial tiles in every dimension
    ("conv3d_k3", "conv", (256, 256, 3, 3, 3), 1, (4, 4, 4), 2),
    ("conv3d_k5_8", "conv", (128, 224, 5, 5, 5), 2, (8, 8, 8), 1),
    ("conv3d_k5_32", "conv", (32, 56, 5, 5, 5), 2, (32, 32, 32), 1),
    ("conv3d_k5_64", "conv", (16, 28, 5, 5, 5), 2, (64, 64, 64), 1),
    ("convT_k3", "convT", (256, 128, 3, 3, 3), 1, (4, 4, 4), 2),
    ("convT_k7_16", "convT", (64, 32, 7, 7, 7), 3, (16, 16, 16), 1),
    ("convT_k7_64_c2", "convT", (16, 2, 7, 7, 7), 3, (64, 64, 64), 1),
    ("convT_k7_32_c14", "convT", (16, 14, 7, 7, 7), 3, (32, 32, 32), 1),
]


@pytest.mark.parametrize("name,kind,wshape,pad,dims,B", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(be, name, kind, wshape, pad, dims, B):
  from corenet_amd import views as V
  from corenet_amd.backend import Transform
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(hash(name) % 1000)
  w = t.randn(wshape, generator=g) / np.sqrt(np.prod(wshape[1:]))
  D, H, W = dims
  is2d = len(wshape) == 4
  if kind == "conv":
    cin, cout = wshape[1], wshape[0]
    fwd, dgr = G.conv_fwd(wshape, pad), G.conv_dgrad(wshape, pad)
    odims = dims
  else:
    cin, cout = wshape[0], wshape[1]
    fwd, dgr = G.convt_fwd(wshape, pad), G.convt_dgrad(wshape, pad)
    odims = (2 * D, 2 * H, 2 * W)
  x = t.randn((B, cin) + ((H, W) if is2d else dims), generator=g)
  scale = t.rand(cin, generator=g) + 0.5; shift = t.randn(cin, generator=g) * 0.3
  bias = t.randn(cout, generator=g)
  y = t.zeros((B, cout + 3) + ((H, W) if is2d else odims))     # written into a channel slice
  wf = EMU_pack(w, fwd); wd = EMU_pack(w, dgr)
  bpack = EMU_pack(bias, None, G.bias_index(cout, 8 if kind == "convT" else 1, fwd.npad, parity_major=(kind == "convT")))

  def yview(tens):
    v = V.view_of(tens).channels(0, cout)
    return V.space_to_depth_view(v, (2, 2, 2), parity_major=True) if kind == "convT" else v

  # forward with a fused pre-ReLU affine transform
  xg, yg = x.to(DEV), y.to(DEV)
  trc = Transform(scale, shift, pre_relu=True); trg = Transform(scale.to(DEV), shift.to(DEV), pre_relu=True)
  EMU.conv_fwd(V.view_of(x), trc, wf, fwd.npad, bpack, 0, yview(y), fwd.window, fwd.pad_lo)
  be.conv_fwd(V.view_of(xg), trg, wf.to(DEV), fwd.npad, bpack.to(DEV), 0, yview(yg), fwd.window, fwd.pad_lo, 0,
             boxes=(fwd.n_boxes, fwd.c_boxes))
  close(yg, y, 3e-5, name + " fwd")
  assert float(yg[:, cout:].abs().max()) == 0.0        # untouched channels of the concat buffer
  # cross-check the contract itself against torch's own op
  xt = x.relu() * scale.view(1, -1, *([1] * (x.dim() - 2))) + shift.view(1, -1, *([1] * (x.dim() - 2)))
  if kind == "conv":
    ref = (t.nn.functional.conv2d if is2d else t.nn.functional.conv3d)(xt, w, bias, padding=pad)
  else:
    ref = t.nn.functional.conv_transpose3d(xt, w, bias, stride=2, padding=pad, output_padding=1)
  close(y[:, :cout], ref, 2e-5, name + " contract-vs-torch")
  # data gradient (accumulating into an existing buffer)
  dy = t.randn(ref.shape, generator=g)
  dyb = t.zeros_like(y); dyb[:, :cout] = dy
  dx = t.randn(x.shape, generator=g); dxg = dx.to(DEV)
  dyg = dyb.to(DEV)
  EMU.conv_fwd(yview(dyb), None, wd, dgr.npad, None, 0, V.view_of(dx), dgr.window, dgr.pad_lo, accumulate=True)
  be.conv_fwd(yview(dyg), None, wd.to(DEV), dgr.npad, None, 0, V.view_of(dxg), dgr.window, dgr.pad_lo, 0, True,
             boxes=(dgr.n_boxes, dgr.c_boxes))
  close(dxg, dx, 3e-5, name + " dgrad")
  # weight gradient
  dw = t.zeros(wf.numel()); dwg = t.zeros(wf.numel(), device=DEV)
  EMU.conv_wgrad(V.view_of(x), trc, yview(dyb), dw, fwd.npad, fwd.window, fwd.pad_lo, True)
  be.conv_wgrad(V.view_of(xg), trg, yview(dyg), dwg, fwd.npad, fwd.window, fwd.pad_lo, True,
                boxes=(fwd.n_boxes, fwd.c_boxes))
  # entries of the packed gradient that belong to structural zeros of the weights (index -1) are never
  # scattered back; with tap boxes the library does not compute them
  real = t.as_tensor(fwd.index) >= 0
  close(t.where(real.to(DEV), dwg, t.zeros((), device=DEV)), t.where(real, dw, t.zeros(())), 5e-5, name + " wgrad")
  # un-packed gradient equals autograd's
  wref = w.clone().requires_grad_(True)
  if kind == "conv":
    out = (t.nn.functional.conv2d if is2d else t.nn.functional.conv3d)(xt, wref, None, padding=pad)
  else:
    out = t.nn.functional.conv_transpose3d(xt, wref, None, stride=2, padding=pad, output_padding=1)
  out.backward(dy)
  gw = t.zeros(w.numel())
  EMU.scatter(dw, t.as_tensor(fwd.index), gw)
  close(gw.view(wshape), wref.grad, 5e-5, name + " wgrad-vs-autograd")


E2D_CASES = [   # name, (Cout, Cin, k, k), (H, W), B: the encoder's stride-1 layers (resnet50.py:49-115) at batch 4 / 2
    ("s2_3x3", (64, 64, 3, 3), (64, 64), 2),
    ("s3_3x3", (128, 128, 3, 3), (32, 32), 4),
    ("s4_3x3", (256, 256, 3, 3), (16, 16), 4),          # split-K
    ("s5_3x3", (512, 512, 3, 3), (8, 8), 4),            # 8x8 tiles, split-K
    ("s2_1x1_64_256", (256, 64, 1, 1), (64, 64), 2),    # half a channel chunk
    ("s2_1x1_256_64", (64, 256, 1, 1), (64, 64), 2),
    ("s3_1x1_512_128", (128, 512, 1, 1), (32, 32), 4),
    ("s4_1x1_1024_256", (256, 1024, 1, 1), (16, 16), 4),
    ("s5_1x1_2048_512", (512, 2048, 1, 1), (8, 8), 4),
    ("s5_1x1_512_2048", (2048, 512, 1, 1), (8, 8), 1),
    ("1x1_192_64_b1", (64, 192, 1, 1), (8, 8), 1),      # one and a half chunks, a single tile
]


@pytest.mark.parametrize("name,wshape,hw,B", E2D_CASES, ids=[c[0] for c in E2D_CASES])
def test_conv2d_bf3_encoder_engine(be, name, wshape, hw, B):
  """crn_bf3_operands + crn_conv2d_bf3 (csrc/conv_e2d.hip) on the encoder's layer shapes: forward with the fused
  BatchRenorm-apply + ReLU transform and bias, and the accumulating data gradient, against the contract emulator on
  the fp32 packed weights (same 2e-5-of-range bar as the decoder's bf16x3 engine); the operand blocks themselves
  are compared bit for bit with the emulator's."""
  if _SELF:
    return
  from corenet_amd import views as V
  from corenet_amd.backend import Transform
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(len(name))
  w = t.randn(wshape, generator=g) / np.sqrt(np.prod(wshape[1:]))
  cout, cin, k = wshape[0], wshape[1], wshape[2]
  pad = k // 2
  fwd, dgr = G.conv_fwd(wshape, pad), G.conv_dgrad(wshape, pad)
  assert G.operand_eligible(fwd) and G.operand_eligible(dgr)
  H, W = hw
  x = t.randn((B, cin, H, W), generator=g)
  scale = t.rand(cin, generator=g) + 0.5; shift = t.randn(cin, generator=g) * 0.3
  bias = t.randn(cout, generator=g)
  wf, wd = EMU_pack(w, fwd), EMU_pack(w, dgr)
  packed = t.cat([wf, wd])
  nf, nd = G.operand_entries(fwd), G.operand_entries(dgr)
  desc, blocks = G.operand_table([(0, 0, fwd), (wf.numel(), nf, dgr)])
  wop = t.zeros((nf + nd) * 32, dtype=t.uint8)
  EMU.bf3_operands(packed, (t.as_tensor(desc), blocks), wop)
  wopg = t.zeros_like(wop).to(DEV)
  be.bf3_operands(packed.to(DEV), (t.as_tensor(desc).to(DEV), blocks), wopg)
  assert t.equal(wopg.cpu(), wop), name
  y = t.zeros((B, cout, H, W)); yg = y.to(DEV)
  trc = Transform(scale, shift, post_relu=True); trg = Transform(scale.to(DEV), shift.to(DEV), post_relu=True)
  xg = x.to(DEV)
  EMU.conv_fwd(V.view_of(x), trc, wf, fwd.npad, bias, 0, V.view_of(y), fwd.window, fwd.pad_lo)
  be.conv2d_bf3(V.view_of(xg), trg, wopg[:nf * 32], fwd.npad, bias.to(DEV), 0, V.view_of(yg), fwd.window, fwd.pad_lo)
  e = float((yg.cpu() - y).abs().max() / y.abs().max())
  print(f"e2d {name} fwd: max-abs-err/max = {e:.2e}")
  assert e <= 2e-5, (name, "fwd", e)
  dy = t.randn((B, cout, H, W), generator=g)
  dx = t.randn(x.shape, generator=g); dxg = dx.to(DEV)
  EMU.conv_fwd(V.view_of(dy), None, wd, dgr.npad, None, 0, V.view_of(dx), dgr.window, dgr.pad_lo, accumulate=True)
  be.conv2d_bf3(V.view_of(dy.to(DEV)), None, wopg[nf * 32:], dgr.npad, None, 0, V.view_of(dxg), dgr.window, dgr.pad_lo,
                accumulate=True)
  e = float((dxg.cpu() - dx).abs().max() / dx.abs().max())
  print(f"e2d {name} dgrad: max-abs-err/max = {e:.2e}")
  assert e <= 2e-5, (name, "dgrad", e)


@pytest.mark.parametrize("name,wshape,hw,B", E2D_CASES, ids=[c[0] for c in E2D_CASES])
def test_conv_wgrad_2d_bf3(be, name, wshape, hw, B):
  """crn_conv_wgrad_2d_bf3 (operands straight from HBM, K = positions) on the encoder's 1x1 and 3x3 shapes, with the
  fused input transform, accumulating into a non-zero dw: against the contract emulator, 2e-5 of the gradient's range."""
  if _SELF:
    return
  from corenet_amd import views as V
  from corenet_amd.backend import Transform
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(len(name) + 7)
  cout, cin = wshape[0], wshape[1]
  fwd = G.conv_fwd(wshape, wshape[2] // 2)
  H, W = hw
  x = t.randn((B, cin, H, W), generator=g); dy = t.randn((B, cout, H, W), generator=g)
  scale = t.rand(cin, generator=g) + 0.5; shift = t.randn(cin, generator=g) * 0.3
  trc = Transform(scale, shift, post_relu=True); trg = Transform(scale.to(DEV), shift.to(DEV), post_relu=True)
  dw0 = t.randn(cin * fwd.taps * fwd.npad, generator=g)
  dw = dw0.clone(); dwg = dw0.to(DEV)
  EMU.conv_wgrad(V.view_of(x), trc, V.view_of(dy), dw, fwd.npad, fwd.window, fwd.pad_lo, False)
  be.conv_wgrad(V.view_of(x.to(DEV)), trg, V.view_of(dy.to(DEV)), dwg, fwd.npad, fwd.window, fwd.pad_lo, False,
                math="bf16x3_2d")
  e = float((dwg.cpu() - dw).abs().max() / (dw - dw0).abs().max())
  print(f"wgrad 2d {name}: max-abs-err/max = {e:.2e}")
  assert e <= 2e-5, (name, e)
  dwz = t.full_like(dwg, 7.0)
  be.conv_wgrad(V.view_of(x.to(DEV)), None, V.view_of(dy.to(DEV)), dwz, fwd.npad, fwd.window, fwd.pad_lo, True,
                math="bf16x3_2d")
  dwr = t.zeros_like(dw)
  EMU.conv_wgrad(V.view_of(x), None, V.view_of(dy), dwr, fwd.npad, fwd.window, fwd.pad_lo, True)
  assert float((dwz.cpu() - dwr).abs().max() / dwr.abs().max()) <= 2e-5


BF3_CASES = [c for c in CONV_CASES if c[0] in ("conv3d_k5_32", "conv3d_k5_64", "convT_k7_16", "convT_k7_64_c2", "convT_k7_32_c14")] + [
    ("conv3d_k5_16_c112", "conv", (64, 112, 5, 5, 5), 2, (16, 16, 16), 2),
    ("convT_k7_32_c16", "convT", (32, 16, 7, 7, 7), 3, (32, 32, 32), 1),
    ("convT_k7_64_c14", "convT", (16, 14, 7, 7, 7), 3, (64, 64, 64), 1),       # m7 / m9 logits layer at full size
    ("conv3d_k5_8_c224", "conv", (128, 224, 5, 5, 5), 2, (8, 8, 8), 2),       # stage 3: 8^3 tiles, split-K
    ("convT_k7_8_c128", "convT", (128, 64, 7, 7, 7), 3, (8, 8, 8), 2),
]


@pytest.mark.parametrize("cout,dims,B", [(14, (64, 64, 64), 1), (14, (8, 16, 32), 2), (2, (8, 8, 16), 3), (16, (4, 16, 16), 2),
                                         (9, (12, 8, 48), 1)])
def test_convt_parity_walk_fwd_dgrad(be, cout, dims, B):
  """The parity-walk kernels of decoder stage_6.t1 (csrc/convt_par.hip: ConvTranspose3d 16 -> cout, k 7, stride 2, padding 3,
  output_padding 1, reconstruction_decoder.py:89-95) against torch's own conv_transpose3d / its adjoint on T(x) (the contract
  emulator): the weight image (crn_bf3_gather_image) bit for bit, forward with the fused pre-ReLU affine transform + bias into
  a channel slice of a wider buffer (the channels behind it untouched), data gradient plain and accumulating; 2e-5 of the
  output range like the generic split-bf16 engine, and equal to that engine's result to the same bar.  Shapes: the m7 / m9 layer
  at full size, tiles at every border, cout = 2 / 9 / 16 (one or two 8-channel halves of n)."""
  from corenet_amd.backend import Transform
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(cout * 100 + dims[0])
  D, H, W = dims
  wshape = (16, cout, 7, 7, 7)
  w = t.randn(wshape, generator=g) / np.sqrt(16 * 343 / 8)
  params = t.cat([t.randn(37, generator=g), w.reshape(-1), t.randn(5, generator=g)])       # the weight inside a larger slab
  x = t.randn(B, 16, D, H, W, generator=g)
  scale, shift = t.rand(16, generator=g) + 0.5, t.randn(16, generator=g) * 0.3
  bias = t.randn(cout, generator=g)
  imgs = {}
  for kind, fn in (("fwd", G.convt_par_fwd_table), ("dgrad", G.convt_par_dgrad_table)):
    tab, nbytes = fn(wshape, 37)
    ic, ig = t.zeros(nbytes, dtype=t.uint8), t.zeros(nbytes, dtype=t.uint8, device=DEV)
    EMU.bf3_gather_image(params, t.as_tensor(tab), ic)
    be.bf3_gather_image(params.to(DEV), t.as_tensor(tab).to(DEV), ig)
    assert t.equal(ig.cpu(), ic), (kind, "image")
    tab0, _ = fn(wshape, 0)
    imgs[kind] = (ic, ig, tab0)
  y = t.full((B, cout + 3, 2 * D, 2 * H, 2 * W), 0.25); yg = y.to(DEV)
  trc = Transform(scale, shift, pre_relu=True); trg = Transform(scale.to(DEV), shift.to(DEV), pre_relu=True)
  EMU.convt_par_fwd(x, trc, imgs["fwd"][0], bias, y, cout, host_table=imgs["fwd"][2])
  be.convt_par_fwd(x.to(DEV), trg, imgs["fwd"][1], bias.to(DEV), yg, cout, host_table=imgs["fwd"][2])
  e = float((yg.cpu() - y).abs().max() / y.abs().max())
  print(f"parity walk cout {cout} {dims} fwd: max-abs-err/max = {e:.2e}")
  assert e <= 2e-5, ("fwd", e)
  assert bool((yg[:, cout:] == 0.25).all())
  ref = t.nn.functional.conv_transpose3d(x.relu() * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1), w, bias, stride=2,
                                         padding=3, output_padding=1)
  assert float((yg[:, :cout].cpu() - ref).abs().max() / ref.abs().max()) <= 2e-5
  dy = t.randn(B, cout + 3, 2 * D, 2 * H, 2 * W, generator=g)
  dx0 = t.randn(B, 16, D, H, W, generator=g)
  for accumulate in (False, True):
    dx, dxg = dx0.clone(), dx0.to(DEV)
    EMU.convt_par_dgrad(dy, cout, imgs["dgrad"][0], dx, accumulate, host_table=imgs["dgrad"][2])
    be.convt_par_dgrad(dy.to(DEV), cout, imgs["dgrad"][1], dxg, accumulate, host_table=imgs["dgrad"][2])
    e = float((dxg.cpu() - dx).abs().max() / dx.abs().max())
    print(f"parity walk cout {cout} {dims} dgrad accumulate={accumulate}: max-abs-err/max = {e:.2e}")
    assert e <= 2e-5, ("dgrad", accumulate, e)
  xr = x.clone().requires_grad_(True)
  t.nn.functional.conv_transpose3d(xr, w, None, stride=2, padding=3, output_padding=1).backward(dy[:, :cout])
  dxg = t.zeros(B, 16, D, H, W, device=DEV)
  be.convt_par_dgrad(dy.to(DEV), cout, imgs["dgrad"][1], dxg, False, host_table=imgs["dgrad"][2])
  assert float((dxg.cpu() - xr.grad).abs().max() / xr.grad.abs().max()) <= 2e-5          # ... and autograd of torch's own op
  if _SELF:
    return
  # weight gradient (crn_convt_s2k7_wgrad_bf3 through conv_wgrad(math="ct_par")): the real entries of the layer's packed gradient
  # against the contract of the window-correlation form, un-packed against autograd of torch's own op, and accumulating
  from corenet_amd import views as V
  fwd = G.convt_fwd(wshape, 3)
  dyc = dy[:, :cout].contiguous(); dyg = dyc.to(DEV); xg = x.to(DEV)
  yv = lambda tt: V.space_to_depth_view(V.view_of(tt), (2, 2, 2), parity_major=True)
  dw = t.zeros(fwd.index.size); dwg = t.full((fwd.index.size,), 7.0, device=DEV)
  EMU.conv_wgrad(V.view_of(x), trc, yv(dyc), dw, fwd.npad, fwd.window, fwd.pad_lo, True)
  be.conv_wgrad(V.view_of(xg), trg, yv(dyg), dwg, fwd.npad, fwd.window, fwd.pad_lo, True, boxes=(fwd.n_boxes, fwd.c_boxes), math="ct_par")
  real = t.as_tensor(fwd.index) >= 0
  got = t.where(real, dwg.cpu(), t.zeros(())); want = t.where(real, dw, t.zeros(()))
  e = float((got - want).abs().max() / want.abs().max())
  print(f"parity walk cout {cout} {dims} wgrad: max-abs-err/max = {e:.2e}")
  assert e <= 2e-5, ("wgrad", e)
  assert float(t.where(real, t.zeros(()), dwg.cpu()).abs().max()) == 0.0             # structural zeros of the packed layout stay zero
  # the two ways the workgroups get T(x): the operand image made ahead by the caller (crn_convt_s2k7_ximage, what the engine does
  # under the forward pass) and the fused transform + split of the fp32 input (CRN_CT_XIMG=0) -- same contract
  img = be.convt_ximage(xg, trg)
  assert img is not None
  # ... the image itself, bit for bit: entry [b][chunk * 2 + (hi, lo)][position] = the 8 channels of the chunk of T(x) as bf16, hi = the
  # round-to-nearest-even bf16 of the fp32 value, lo = the bf16 of what is left (the split of conv_bf3.hip)
  tx = (x.relu() * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)).float()            # trc: pre_relu, then scale / shift
  hi = tx.to(t.bfloat16); lo = (tx - hi.float()).to(t.bfloat16)
  want_img = t.stack([hi[:, :8], lo[:, :8], hi[:, 8:], lo[:, 8:]], 1).reshape(B, 4, 8, -1).permute(0, 1, 3, 2).contiguous()   # [B][4][S][8]
  got_img = img[:want_img.numel() * 2].view(t.bfloat16).view(B, 4, -1, 8).cpu()
  assert t.equal(got_img.view(t.int16), want_img.view(t.int16)), "operand image differs from bf16 hi / lo of T(x)"
  for label, kw, env in (("image made ahead", {"ximg": img}, None), ("fused staging", {}, "0")):
    old_env = os.environ.get("CRN_CT_XIMG")
    if env is not None:
      os.environ["CRN_CT_XIMG"] = env
    try:
      dwg3 = t.full((fwd.index.size,), 3.0, device=DEV)
      be.conv_wgrad(V.view_of(xg), trg, yv(dyg), dwg3, fwd.npad, fwd.window, fwd.pad_lo, True, boxes=(fwd.n_boxes, fwd.c_boxes), math="ct_par", **kw)
    finally:
      if env is not None:
        if old_env is None: os.environ.pop("CRN_CT_XIMG")
        else: os.environ["CRN_CT_XIMG"] = old_env
    e3 = float((t.where(real, dwg3.cpu(), t.zeros(())) - want).abs().max() / want.abs().max())
    print(f"parity walk cout {cout} {dims} wgrad ({label}): max-abs-err/max = {e3:.2e}")
    assert e3 <= 2e-5, ("wgrad", label, e3)
  wref = w.clone().requires_grad_(True)
  t.nn.functional.conv_transpose3d(x.relu() * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1), wref, None, stride=2, padding=3,
                                   output_padding=1).backward(dyc)
  gw = t.zeros(w.numel())
  EMU.scatter(dwg.cpu(), t.as_tensor(fwd.index), gw)
  e = float((gw.view(wshape) - wref.grad).abs().max() / wref.grad.abs().max())
  print(f"parity walk cout {cout} {dims} wgrad vs torch autograd: max-abs-err/max = {e:.2e}")
  assert e <= 5e-5, ("wgrad-vs-autograd", e)
  dwg2 = dwg.clone()
  be.conv_wgrad(V.view_of(xg), trg, yv(dyg), dwg2, fwd.npad, fwd.window, fwd.pad_lo, False, boxes=(fwd.n_boxes, fwd.c_boxes), math="ct_par")
  assert float((t.where(real, dwg2.cpu(), t.zeros(())) - 2 * want).abs().max() / want.abs().max()) <= 4e-5
  if dims == (64, 64, 64):
    dyfull = dy.to(DEV); biasg = bias.to(DEV)
    for nm, fn in (("fwd", lambda: be.convt_par_fwd(xg, trg, imgs["fwd"][1], biasg, yg, cout)),
                   ("dgrad", lambda: be.convt_par_dgrad(dyfull, cout, imgs["dgrad"][1], dxg, False)),
                   ("wgrad", lambda: be.conv_wgrad(V.view_of(xg), trg, yv(dyg), dwg2, fwd.npad, fwd.window, fwd.pad_lo, False,
                                                   boxes=(fwd.n_boxes, fwd.c_boxes), math="ct_par"))):
      xs = [fn() for _ in range(2)]
      a, b_ = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
      t.cuda.synchronize(); a.record()
      for _ in range(5): fn()
      b_.record(); t.cuda.synchronize()
      print(f"parity walk cout {cout} B {B} {nm}: {a.elapsed_time(b_) / 5 * 1e3:.0f} us")


@pytest.mark.parametrize("dims,B", [((64, 64, 64), 2), ((8, 16, 32), 3), ((4, 8, 16), 1), ((12, 24, 16), 2)])
def test_convt_resident_weights_two_classes(be, dims, B):
  """The two-class (h7) kernels of decoder stage_6.t1 (convt_res_kernel, csrc/convt_par.hip: weights resident in LDS, one persistent
  workgroup per CU): forward with transform + bias into a channel slice, data gradient plain and accumulating, against torch's
  conv_transpose3d / its adjoint (2e-5 of the range) and autograd; tiles at every border, workgroups that walk several tiles and
  fewer tiles than CUs."""
  from corenet_amd.backend import Transform
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(dims[0] + B)
  D, H, W = dims
  wshape = (16, 2, 7, 7, 7)
  w = t.randn(wshape, generator=g) / np.sqrt(16 * 343 / 8)
  x = t.randn(B, 16, D, H, W, generator=g)
  scale, shift = t.rand(16, generator=g) + 0.5, t.randn(16, generator=g) * 0.3
  bias = t.randn(2, generator=g)
  imgs = {}
  for kind, fn in (("fwd", G.convt_res_fwd_table), ("dgrad", G.convt_res_dgrad_table)):
    tab, nbytes = fn(wshape, 0)
    ic, ig = t.zeros(nbytes, dtype=t.uint8), t.zeros(nbytes, dtype=t.uint8, device=DEV)
    EMU.bf3_gather_image(w.reshape(-1), t.as_tensor(tab), ic)
    be.bf3_gather_image(w.reshape(-1).to(DEV), t.as_tensor(tab).to(DEV), ig)
    assert t.equal(ig.cpu(), ic), (kind, "image")
    imgs[kind] = (ic, ig, tab)
  y = t.full((B, 5, 2 * D, 2 * H, 2 * W), 0.25); yg = y.to(DEV)
  trg = Transform(scale.to(DEV), shift.to(DEV), pre_relu=True)
  ref = t.nn.functional.conv_transpose3d(x.relu() * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1), w, bias, stride=2,
                                         padding=3, output_padding=1)
  be.convt_par_fwd(x.to(DEV), trg, imgs["fwd"][1], bias.to(DEV), yg, 2, host_table=imgs["fwd"][2], resident=True)
  e = float((yg[:, :2].cpu() - ref).abs().max() / ref.abs().max())
  print(f"resident weights {dims} B {B} fwd: max-abs-err/max = {e:.2e}")
  assert e <= 2e-5, ("fwd", e)
  assert bool((yg[:, 2:] == 0.25).all())
  dy = t.randn(B, 5, 2 * D, 2 * H, 2 * W, generator=g)
  xr = x.clone().requires_grad_(True)
  t.nn.functional.conv_transpose3d(xr, w, None, stride=2, padding=3, output_padding=1).backward(dy[:, :2])
  dx0 = t.randn(B, 16, D, H, W, generator=g)
  for accumulate in (False, True):
    dxg = dx0.clone().to(DEV)
    be.convt_par_dgrad(dy.to(DEV), 2, imgs["dgrad"][1], dxg, accumulate, host_table=imgs["dgrad"][2], resident=True)
    want = xr.grad + (dx0 if accumulate else 0)
    e = float((dxg.cpu() - want).abs().max() / want.abs().max())
    print(f"resident weights {dims} B {B} dgrad accumulate={accumulate}: max-abs-err/max = {e:.2e}")
    assert e <= 2e-5, ("dgrad", accumulate, e)
  if _SELF or dims != (64, 64, 64):
    return
  xg, dyg, bg = x.to(DEV), dy.to(DEV), bias.to(DEV)
  for nm, fn in (("fwd", lambda: be.convt_par_fwd(xg, trg, imgs["fwd"][1], bg, yg, 2, resident=True)),
                 ("dgrad", lambda: be.convt_par_dgrad(dyg, 2, imgs["dgrad"][1], dxg, False, resident=True))):
    fn(); fn()
    a, b_ = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    t.cuda.synchronize(); a.record()
    for _ in range(5): fn()
    b_.record(); t.cuda.synchronize()
    print(f"resident weights B {B} {nm}: {a.elapsed_time(b_) / 5 * 1e3:.0f} us")


@pytest.mark.parametrize("name,kind,wshape,pad,dims,B", BF3_CASES, ids=[c[0] for c in BF3_CASES])
def test_conv_bf16x3_fwd_dgrad(be, name, kind, wshape, pad, dims, B):
  """The split-bf16 MFMA engine (crn_conv_fwd_bf3) on the decoder layer shapes of stages 4-6
  (reconstruction_decoder.py:72-95), forward with the fused pre-ReLU affine transform + bias into a channel slice,
  and the data gradient (accumulating), against the same contract emulator as the fp32 engine.
  Error model: operands carry 16 mantissa bits (hi + lo bf16) and the lo*lo product is dropped, i.e. ~2^-16
  relative per product with random sign; the bar is 2e-5 of the output range (fp32 engine: 3e-5 on the same data is
  dominated by summation order), measured values are printed."""
  if _SELF:
    return
  from corenet_amd import views as V
  from corenet_amd.backend import Transform
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(hash(name) % 1000)
  w = t.randn(wshape, generator=g) / np.sqrt(np.prod(wshape[1:]))
  D, H, W = dims
  if kind == "conv":
    cin, cout = wshape[1], wshape[0]
    fwd, dgr = G.conv_fwd(wshape, pad), G.conv_dgrad(wshape, pad)
    odims = dims
  else:
    cin, cout = wshape[0], wshape[1]
    fwd, dgr = G.convt_fwd(wshape, pad), G.convt_dgrad(wshape, pad)
    odims = (2 * D, 2 * H, 2 * W)
  x = t.randn((B, cin) + dims, generator=g)
  scale = t.rand(cin, generator=g) + 0.5; shift = t.randn(cin, generator=g) * 0.3
  bias = t.randn(cout, generator=g)
  y = t.zeros((B, cout + 3) + odims)
  wf = EMU_pack(w, fwd); wd = EMU_pack(w, dgr)
  bpack = EMU_pack(bias, None, G.bias_index(cout, 8 if kind == "convT" else 1, fwd.npad, parity_major=(kind == "convT")))

  def yview(tens):
    v = V.view_of(tens).channels(0, cout)
    return V.space_to_depth_view(v, (2, 2, 2), parity_major=True) if kind == "convT" else v

  xg, yg = x.to(DEV), y.to(DEV)
  trc = Transform(scale, shift, pre_relu=True); trg = Transform(scale.to(DEV), shift.to(DEV), pre_relu=True)
  EMU.conv_fwd(V.view_of(x), trc, wf, fwd.npad, bpack, 0, yview(y), fwd.window, fwd.pad_lo)
  be.conv_fwd(V.view_of(xg), trg, wf.to(DEV), fwd.npad, bpack.to(DEV), 0, yview(yg), fwd.window, fwd.pad_lo, 0,
              boxes=(fwd.n_boxes, fwd.c_boxes), math="bf16x3")
  e = float((yg.cpu() - y).abs().max() / y.abs().max())
  print(f"bf16x3 {name} fwd: max-abs-err/max = {e:.2e}")
  assert e <= 2e-5, (name, "fwd", e)
  assert float(yg[:, cout:].abs().max()) == 0.0
  # the same launches with the weights pre-arranged as slab images (crn_bf3_operands + crn_conv_fwd_bf3_slabs):
  # the images equal the emulator's bit for bit, and so do the results of the two forms of the kernel
  nsf, nsd = G.slab_entries(fwd), G.slab_entries(dgr)
  desc, blocks = G.operand_table([(0, 0, fwd, True), (wf.numel(), nsf, dgr, True)])
  slabs = t.zeros((nsf + nsd) * 32, dtype=t.uint8)
  EMU.bf3_operands(t.cat([wf, wd]), (t.as_tensor(desc), blocks), slabs)
  slabs_g = t.zeros_like(slabs).to(DEV)
  be.bf3_operands(t.cat([wf, wd]).to(DEV), (t.as_tensor(desc).to(DEV), blocks), slabs_g)
  assert t.equal(slabs_g.cpu(), slabs), name
  yg2 = t.zeros_like(yg)
  be.conv_fwd(V.view_of(xg), trg, None, fwd.npad, bpack.to(DEV), 0, yview(yg2), fwd.window, fwd.pad_lo, 0,
              boxes=(fwd.n_boxes, fwd.c_boxes), math="bf16x3", wslab=slabs_g[:nsf * 32])
  assert t.equal(yg2, yg), (name, "fwd slabs")
  ye = t.zeros_like(y)
  EMU.conv_fwd(V.view_of(x), trc, None, fwd.npad, bpack, 0, yview(ye), fwd.window, fwd.pad_lo, wslab=slabs[:nsf * 32])
  assert float((ye - y).abs().max() / y.abs().max()) < 2e-5
  dy = t.randn((B, cout) + odims, generator=g)
  dyb = t.zeros_like(y); dyb[:, :cout] = dy
  dx = t.randn(x.shape, generator=g); dxg = dx.to(DEV)
  dyg = dyb.to(DEV)
  EMU.conv_fwd(yview(dyb), None, wd, dgr.npad, None, 0, V.view_of(dx), dgr.window, dgr.pad_lo, accumulate=True)
  dxg2 = dxg.clone()
  be.conv_fwd(yview(dyg), None, wd.to(DEV), dgr.npad, None, 0, V.view_of(dxg), dgr.window, dgr.pad_lo, 0, True,
              boxes=(dgr.n_boxes, dgr.c_boxes), math="bf16x3")
  e = float((dxg.cpu() - dx).abs().max() / dx.abs().max())
  print(f"bf16x3 {name} dgrad: max-abs-err/max = {e:.2e}")
  assert e <= 2e-5, (name, "dgrad", e)
  be.conv_fwd(yview(dyg), None, None, dgr.npad, None, 0, V.view_of(dxg2), dgr.window, dgr.pad_lo, 0, True,
              boxes=(dgr.n_boxes, dgr.c_boxes), math="bf16x3", wslab=slabs_g[nsf * 32:])
  assert t.equal(dxg2, dxg), (name, "dgrad slabs")
  # weight gradient (crn_conv_wgrad_bf3): real entries of the packed gradient against the contract, and the
  # un-packed gradient against autograd of torch's own op
  dw = t.zeros(wf.numel()); dwg = t.full((wf.numel(),), 7.0, device=DEV)
  EMU.conv_wgrad(V.view_of(x), trc, yview(dyb), dw, fwd.npad, fwd.window, fwd.pad_lo, True)
  be.conv_wgrad(V.view_of(xg), trg, yview(dyg), dwg, fwd.npad, fwd.window, fwd.pad_lo, True,
                boxes=(fwd.n_boxes, fwd.c_boxes), math="bf16x3")
  real = t.as_tensor(fwd.index) >= 0
  got = t.where(real, dwg.cpu(), t.zeros(())); want = t.where(real, dw, t.zeros(()))
  e = float((got - want).abs().max() / want.abs().max())
  print(f"bf16x3 {name} wgrad: max-abs-err/max = {e:.2e}")
  assert e <= 2e-5, (name, "wgrad", e)
  # ... un-packed (the scatter of corenet_amd/model/conv_geometry index), against autograd of torch's own op on
  # the transformed input: the contract emulator is not in this comparison
  xt = x.relu() * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
  wref = w.clone().requires_grad_(True)
  if kind == "conv":
    t.nn.functional.conv3d(xt, wref, None, padding=pad).backward(dy)
  else:
    t.nn.functional.conv_transpose3d(xt, wref, None, stride=2, padding=pad, output_padding=1).backward(dy)
  gw = t.zeros(w.numel())
  EMU.scatter(dwg.cpu(), t.as_tensor(fwd.index), gw)
  e = float((gw.view(wshape) - wref.grad).abs().max() / wref.grad.abs().max())
  print(f"bf16x3 {name} wgrad vs torch autograd: max-abs-err/max = {e:.2e}")
  assert e <= 5e-5, (name, "wgrad-vs-autograd", e)
  # accumulate into an existing gradient (zero_first = False)
  dwg2 = dwg.clone()
  be.conv_wgrad(V.view_of(xg), trg, yview(dyg), dwg2, fwd.npad, fwd.window, fwd.pad_lo, False,
                boxes=(fwd.n_boxes, fwd.c_boxes), math="bf16x3")
  assert float((t.where(real, dwg2.cpu(), t.zeros(())) - 2 * want).abs().max() / want.abs().max()) <= 4e-5


def EMU_pack(w, geom, index=None):
  idx = t.as_tensor(geom.index if index is None else index)
  out = t.zeros(idx.numel())
  EMU.gather(w.reshape(-1), idx, out)
  return out


@pytest.mark.parametrize("key", ["s6c1", "s5c1", "s3c1"])
def test_conv_fwd_leaves_the_statistics_of_the_norm_behind_it(be, key):
  """crn_conv_fwd_bf3_slabs_stats: the same y as crn_conv_fwd_bf3_slabs, and finalize(partial sums of the launch) == the statistics
  pass over max(y, 0) (scale / shift / saved / running statistics); a launch that splits its reduction reports 0 parts."""
  from corenet_amd import views as V
  from corenet_amd.backend import Transform
  from corenet_amd.model import conv_geometry as G
  # (s5c1 at B = 4: 256 tiles, the launch does not split -- the wave-specialised kernel; at B = 2 it would)
  shapes = {"s6c1": ((16, 28, 5, 5, 5), (32, 32, 64), 2), "s5c1": ((32, 56, 5, 5, 5), (32, 32, 32), 4), "s3c1": ((128, 224, 5, 5, 5), (8, 8, 8), 2)}
  wshape, dims, B = shapes[key]
  cout, cin = wshape[0], wshape[1]
  g = t.Generator().manual_seed(17)
  x = t.randn((B, cin) + dims, generator=g).to(DEV); w = t.randn(wshape, generator=g) * 0.05; bias = t.randn(cout, generator=g)
  fwd = G.conv_fwd(wshape, 2)
  wf = EMU_pack(w, fwd).to(DEV); bp = EMU_pack(bias, None, G.bias_index(cout, 1, fwd.npad)).to(DEV)
  ns = G.slab_entries(fwd)
  desc, blocks = G.operand_table([(0, 0, fwd, True)])
  slab = t.zeros(ns * 32, dtype=t.uint8, device=DEV)
  be.bf3_operands(wf, (t.as_tensor(desc).to(DEV), blocks), slab)
  sc, sh = (t.rand(cin, generator=g) + 0.5).to(DEV), t.randn(cin, generator=g).to(DEV)
  tr = Transform(sc, sh, pre_relu=True)
  S = dims[0] * dims[1] * dims[2]
  y0 = t.zeros((B, cout) + dims, device=DEV); y1 = t.full((B, cout) + dims, float("nan"), device=DEV)
  be.conv_fwd(V.view_of(x), tr, wf, fwd.npad, bp, 0, V.view_of(y0), fwd.window, fwd.pad_lo, 0, math="bf16x3", wslab=slab)
  def bn_state():
    gg = t.Generator().manual_seed(5)
    return [v.to(DEV) for v in (t.rand(cout, generator=gg) + 0.5, t.randn(cout, generator=gg), t.randn(cout, generator=gg),
                                t.rand(cout, generator=gg) + 0.5)] + [t.tensor([20000], dtype=t.int64, device=DEV)]
  gamma, beta, rm, rv, nbt = bn_state()
  s1, h1, v1 = t.zeros(cout, device=DEV), t.zeros(cout, device=DEV), t.zeros(4 * cout, device=DEV)
  parts = be.conv_fwd_stats(V.view_of(x), tr, slab, fwd.npad, bp, 0, V.view_of(y1), fwd.window, fwd.pad_lo, None, cout, True)
  assert t.equal(y0, y1)
  if key == "s3c1":
    assert parts == 0                                # 8^3 maps: the launch splits its reduction
    return
  assert parts > 0
  be.bn_finalize(parts, cout, B * S, gamma, beta, rm, rv, nbt, 1e-3, 0.01, s1, h1, v1)
  gamma2, beta2, rm2, rv2, nbt2 = bn_state()
  s2, h2, v2 = t.zeros(cout, device=DEV), t.zeros(cout, device=DEV), t.zeros(4 * cout, device=DEV)
  be.bn_stats(y0, B, cout, S, cout * S, True, gamma2, beta2, rm2, rv2, nbt2, 1e-3, 0.01, True, s2, h2, v2)
  for a, c, nm in ((s1, s2, "scale"), (h1, h2, "shift"), (v1, v2, "saved"), (rm, rm2, "running_mean"), (rv, rv2, "running_var")):
    close(a, c, 3e-6, f"{key} statistics {nm}")


def test_conv_stem_and_strided(be):
  """7x7/2 stem on the space-to-depth view and the stride-2 1x1 convs (views)."""
  from corenet_amd import views as V
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(3)
  x = t.randn(2, 3, 64, 64, generator=g); w = t.randn(64, 3, 7, 7, generator=g) * 0.1; b = t.randn(64, generator=g)
  geo = G.stem_fwd(w.shape, 3)
  wf = EMU_pack(w, geo); bp = EMU_pack(b, None, G.bias_index(64, 1, geo.npad))
  yg = t.zeros(2, 64, 32, 32, device=DEV); xg = x.to(DEV)
  be.conv_fwd(V.space_to_depth_view(V.view_of(xg), (1, 2, 2)), None, wf.to(DEV), geo.npad, bp.to(DEV), 0,
              V.view_of(yg), geo.window, geo.pad_lo, 0)
  ref = t.nn.functional.conv2d(t.nn.functional.pad(x, [3, 3, 3, 3]), w, b, stride=2)
  close(yg, ref, 2e-5, "stem")
  # stride-2 1x1: forward through a strided input view, dgrad through a strided output view
  x = t.randn(2, 96, 16, 16, generator=g); w = t.randn(40, 96, 1, 1, generator=g) * 0.1
  geo, dgeo = G.conv_fwd(w.shape, 0), G.conv_dgrad(w.shape, 0)
  wf, wd = EMU_pack(w, geo), EMU_pack(w, dgeo)
  xg = x.to(DEV); yg = t.zeros(2, 40, 8, 8, device=DEV)
  be.conv_fwd(V.strided_view(V.view_of(xg), (1, 2, 2)), None, wf.to(DEV), geo.npad, None, 0, V.view_of(yg),
              geo.window, geo.pad_lo, 0)
  ref = t.nn.functional.conv2d(x, w, None, stride=2)
  close(yg, ref, 2e-5, "1x1 s2 fwd")
  dy = t.randn(ref.shape, generator=g)
  dxg = t.zeros(2, 96, 16, 16, device=DEV)
  be.conv_fwd(V.view_of(dy.to(DEV)), None, wd.to(DEV), dgeo.npad, None, 0,
              V.strided_view(V.view_of(dxg), (1, 2, 2)), dgeo.window, dgeo.pad_lo, 0, True)
  xr = x.clone().requires_grad_(True)
  t.nn.functional.conv2d(xr, w, None, stride=2).backward(dy)
  close(dxg, xr.grad, 2e-5, "1x1 s2 dgrad")
  dwg = t.zeros(wf.numel(), device=DEV)
  be.conv_wgrad(V.strided_view(V.view_of(xg), (1, 2, 2)), None, V.view_of(dy.to(DEV)), dwg, geo.npad, geo.window,
                geo.pad_lo, True)
  wr = w.clone().requires_grad_(True)
  t.nn.functional.conv2d(x, wr, None, stride=2).backward(dy)
  gw = t.zeros(w.numel()); EMU.scatter(dwg.cpu(), t.as_tensor(geo.index), gw)
  close(gw.view(w.shape), wr.grad, 3e-5, "1x1 s2 wgrad")


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (4, 256, 256), (1, 56, 72), (3, 36, 40)])
def test_stem_conv_own_kernels(be, B, H, W):
  """csrc/stem_conv.hip (ZeroPad2d(3) + Conv2d 7x7 / 2, resnet50.py:122-124): forward against the CPU fp32 convolution, the
  BatchRenorm partial sums it leaves against the statistics pass (scale / shift / saved / running statistics), the weight
  gradient in the packed layout against autograd -- tiles cut by the image border included (56 x 72, 36 x 40)."""
  from corenet_amd import views as V
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(11)
  x = t.randn(B, 3, H, W, generator=g) * 50.0; w = t.randn(64, 3, 7, 7, generator=g) * 0.1; b = t.randn(64, generator=g)
  geo = G.stem_fwd(w.shape, 3)
  wf = EMU_pack(w, geo).to(DEV); bp = EMU_pack(b, None, G.bias_index(64, 1, geo.npad)).to(DEV)
  xg = x.to(DEV); H1, W1 = H // 2, W // 2
  assert int(be.lib.crn_stem_conv_parts(B, H, W)) == B * ((H1 + 3) // 4) * ((W1 + 31) // 32)
  assert int(be.lib.crn_stem_conv_parts(B, H + 1, W)) == 0 and int(be.lib.crn_stem_conv_parts(B, H, W + 4)) == 0
  yg = t.full((B, 64, H1, W1), float("nan"), device=DEV)
  parts = be.stem_conv_fwd(xg, wf, bp, yg, True)
  ref = t.nn.functional.conv2d(t.nn.functional.pad(x, [3, 3, 3, 3]), w, b, stride=2)
  close(yg, ref, 2e-5, "stem fwd")
  # the generic engine on the space-to-depth view computes the same thing
  y2 = t.zeros_like(yg)
  be.conv_fwd(V.space_to_depth_view(V.view_of(xg), (1, 2, 2)), None, wf, geo.npad, bp, 0, V.view_of(y2), geo.window, geo.pad_lo, 0)
  close(yg, y2, 2e-5, "stem fwd vs generic engine")
  # statistics: finalize from the launch's partial sums == the statistics pass over y
  def bn_state():
    gg = t.Generator().manual_seed(5)
    return [v.to(DEV) for v in (t.rand(64, generator=gg) + 0.5, t.randn(64, generator=gg), t.randn(64, generator=gg),
                                t.rand(64, generator=gg) + 0.5)] + [t.tensor([20000], dtype=t.int64, device=DEV)]
  outs = []
  for fused in (True, False):
    gamma, beta, rm, rv, nbt = bn_state()
    sc, sh, sv = t.zeros(64, device=DEV), t.zeros(64, device=DEV), t.zeros(4 * 64, device=DEV)
    if fused:
      parts = be.stem_conv_fwd(xg, wf, bp, yg, True)
      be.bn_finalize(parts, 64, B * H1 * W1, gamma, beta, rm, rv, nbt, 1e-5, 0.01, sc, sh, sv)
    else:
      be.bn_stats(yg, B, 64, H1 * W1, 64 * H1 * W1, False, gamma, beta, rm, rv, nbt, 1e-5, 0.01, True, sc, sh, sv)
    outs.append((sc, sh, sv, rm, rv))
  for a, c, nm in zip(outs[0], outs[1], ("scale", "shift", "saved", "running_mean", "running_var")):
    close(a, c.cpu(), 2e-6, "stem statistics " + nm)
  # without statistics (eval): same y
  y3 = t.zeros_like(yg)
  assert be.stem_conv_fwd(xg, wf, bp, y3, False) == 0
  assert t.equal(y3, yg)
  # weight gradient, added to the packed gradient
  dy = t.randn(ref.shape, generator=g)
  dwg = t.zeros(wf.numel(), device=DEV)
  be.conv_wgrad(V.space_to_depth_view(V.view_of(xg), (1, 2, 2)), None, V.view_of(dy.to(DEV)), dwg, geo.npad, geo.window,
                geo.pad_lo, False, math="stem")
  wr = w.clone().requires_grad_(True)
  t.nn.functional.conv2d(t.nn.functional.pad(x, [3, 3, 3, 3]), wr, None, stride=2).backward(dy)
  gw = t.zeros(w.numel()); EMU.scatter(dwg.cpu(), t.as_tensor(geo.index), gw)
  close(gw.view(w.shape), wr.grad, 3e-5, "stem wgrad")
  # every packed slot that is no real tap stays zero (the own kernel; in deterministic mode the call falls back to the generic
  # engine, which also fills the window slots of the space-to-depth form that hold no tap -- the un-pack ignores them)
  if os.environ.get("CRN_DETERMINISTIC", "0") != "1":
    mask = t.as_tensor(geo.index) < 0
    assert float(dwg.cpu()[mask].abs().max()) == 0.0
  # a second call accumulates
  be.conv_wgrad(V.space_to_depth_view(V.view_of(xg), (1, 2, 2)), None, V.view_of(dy.to(DEV)), dwg, geo.npad, geo.window,
                geo.pad_lo, False, math="stem")
  gw2 = t.zeros(w.numel()); EMU.scatter(dwg.cpu(), t.as_tensor(geo.index), gw2)
  close(gw2.view(w.shape), 2 * wr.grad, 3e-5, "stem wgrad accumulates")


def test_conv_1to4(be):
  from corenet_amd import views as V
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(4)
  B = 4
  x = t.randn(B, 67, generator=g); w = t.randn(67, 256, 4, 4, 4, generator=g) * 0.1; b = t.randn(256, generator=g)
  geo, dgeo = G.convt_1to4_fwd(w.shape), G.convt_1to4_dgrad(w.shape)
  wf, wd = EMU_pack(w, geo), EMU_pack(w, dgeo)
  bp = EMU_pack(b, None, G.bias_index(256, 64, geo.npad))
  yg = t.zeros(B, 256, 4, 4, 4, device=DEV); xg = x.to(DEV)
  be.conv_fwd(V.view_of(xg.view(B, 67, 1, 1, 1)), None, wf.to(DEV), geo.npad, bp.to(DEV), 0,
              V.flat_channel_view(V.view_of(yg)), geo.window, geo.pad_lo, 0)
  ref = t.nn.functional.conv_transpose3d(x.view(B, 67, 1, 1, 1), w, b, stride=4)
  close(yg, ref, 2e-5, "1->4 fwd")
  dy = t.randn(ref.shape, generator=g); dyg = dy.to(DEV)
  dxg = t.zeros(B, 67, device=DEV)
  be.conv_fwd(V.flat_channel_view(V.view_of(dyg)), None, wd.to(DEV), dgeo.npad, None, 0,
              V.view_of(dxg.view(B, 67, 1, 1, 1)), dgeo.window, dgeo.pad_lo, 0)
  xr = x.clone().requires_grad_(True)
  t.nn.functional.conv_transpose3d(xr.view(B, 67, 1, 1, 1), w, None, stride=4).backward(dy)
  close(dxg, xr.grad, 2e-5, "1->4 dgrad")
  # weight gradient with the fused pre-ReLU affine input transform, accumulating into an existing gradient
  from corenet_amd.backend import Transform
  sc, sh = t.rand(67, generator=g) + 0.5, t.randn(67, generator=g) * 0.3
  xt = (x.relu() * sc + sh)
  wr = w.clone().requires_grad_(True)
  t.nn.functional.conv_transpose3d(xt.view(B, 67, 1, 1, 1), wr, None, stride=4).backward(dy)
  prev = t.randn(wf.numel(), generator=g)
  dwg = prev.to(DEV).clone()
  be.conv_wgrad(V.view_of(xg.view(B, 67, 1, 1, 1)), Transform(sc.to(DEV), sh.to(DEV), pre_relu=True),
                V.flat_channel_view(V.view_of(dyg)), dwg, geo.npad, geo.window, geo.pad_lo, False)
  gw = t.zeros(w.numel()); EMU.scatter(dwg.cpu() - prev, t.as_tensor(geo.index), gw)
  close(gw.view(w.shape), wr.grad, 2e-5, "1->4 wgrad")
  # forward with the same transform and accumulation
  y2 = t.randn(B, 256, 4, 4, 4, generator=g); y2g = y2.to(DEV)
  be.conv_fwd(V.view_of(xg.view(B, 67, 1, 1, 1)), Transform(sc.to(DEV), sh.to(DEV), pre_relu=True), wf.to(DEV), geo.npad,
              bp.to(DEV), 0, V.flat_channel_view(V.view_of(y2g)), geo.window, geo.pad_lo, 0, True)
  close(y2g, y2 + t.nn.functional.conv_transpose3d(xt.view(B, 67, 1, 1, 1), w, b, stride=4), 2e-5, "1->4 fwd transform+accumulate")


# ------------------------------------------------------------------ BatchRenorm & elementwise
@pytest.mark.parametrize("B,C,S,pre,post,nbt", [(3, 5, 24, False, False, 0), (2, 28, 4096, True, False, 30000),
                                               (4, 64, 1024, False, True, 12000), (4, 67, 1, True, False, 0),
                                               (2, 7, 33, True, False, 7000),
                                               # channel-owner kernels (C >= 64, B*S <= 65536)
                                               (4, 128, 1024, False, True, 12000), (4, 256, 64, True, False, 30000),
                                               (2, 130, 37, False, False, 0), (4, 512, 256, False, True, 6000)])
def test_batch_renorm(be, B, C, S, pre, post, nbt):
  g = t.Generator().manual_seed(B * 100 + C)
  Ct = C + 2
  x = t.randn(B, Ct, S, generator=g) * 2 + 0.4           # channel slice of a wider buffer
  gy = t.randn(B, C, S, generator=g)
  gamma = t.rand(C, generator=g) + 0.5; beta = t.randn(C, generator=g)
  rm0 = t.randn(C, generator=g); rv0 = t.rand(C, generator=g) * 3 + 0.1
  nb = t.tensor([nbt], dtype=t.int64)
  res = []
  for dev, bk in (("cpu", EMU), (DEV, be)):
    xx = x.to(dev); rm, rv = rm0.clone().to(dev), rv0.clone().to(dev)
    sc, sh, sv = t.zeros(C, device=dev), t.zeros(C, device=dev), t.zeros(4 * C, device=dev)
    bk.bn_stats(xx, B, C, S, Ct * S, pre, gamma.to(dev), beta.to(dev), rm, rv, nb.to(dev), 1e-3, 0.01, True,
                sc, sh, sv)
    dx = t.zeros(B, C, S, device=dev); dg = t.zeros(C, device=dev); db = t.zeros(C, device=dev)
    nds = max(1, C - 2)                                   # fused sum(dx) for the first nds channels
    ds = t.full((C,), 7.0, device=dev)                    # stale content must be overwritten, tail untouched
    bk.bn_bwd(xx, Ct * S, gy.to(dev), C * S, B, C, S, pre, post, gamma.to(dev), sc, sh, sv, dx, C * S, dg, db,
              dsum=ds, ndsum=nds)
    res.append([v.cpu() for v in (sc, sh, sv, rm, rv, dx, dg, db)])
    dsum_ref = dx.double().sum((0, 2)).float().cpu()
    got = ds.cpu()
    scale_ = float(dx.abs().sum((0, 2)).max()) + 1e-6     # sum(dx) is ~0 without pre_relu: absolute check
    assert float((got[:nds] - dsum_ref[:nds]).abs().max()) <= 2e-5 * scale_, (dev, got[:nds], dsum_ref[:nds])
    assert bool((got[nds:] == 7.0).all())
  for a, b, nm in zip(res[1], res[0], ["scale", "shift", "saved", "rmean", "rvar", "dx", "dgamma", "dbeta"]):
    close(a, b, 2e-5, nm)
  # and against autograd of the oracle
  sd = {"weight": gamma.clone().requires_grad_(True), "bias": beta.clone().requires_grad_(True),
        "running_mean": rm0.clone(), "running_var": rv0.clone(), "num_batches_tracked": t.tensor(nbt)}
  xi = x[:, :C].clone().requires_grad_(True)
  y = O.batch_renorm(xi.relu() if pre else xi, sd, "", True)
  if post: y = y.relu()
  y.backward(gy)
  close(res[1][5], xi.grad, 1e-4, "dx vs autograd")
  close(res[1][6], sd["weight"].grad, 1e-4, "dgamma vs autograd")
  close(res[1][4], sd["running_var"], 1e-5, "running_var vs oracle")


@pytest.mark.parametrize("B,C,S,shortcut,second,nbt", [
    (4, 256, 4096, "identity", False, 12000),      # encoder stage 2 (16 float4 per thread: the widest register form)
    (4, 1024, 256, "affine", True, 30000),          # down-sampling block of stage 4, feature map also stored pre-ReLU
    (2, 130, 36, "identity", True, 0),              # ragged: the last float4 slots of a workgroup are empty
    (2, 64, 16384, "affine", False, 7000)])         # too large for the register form: the two-launch fallback
def test_bottleneck_tail_fused_equals_two_launches(be, B, C, S, shortcut, second, nbt):
  """crn_batch_renorm_stats_tail (statistics of a bottleneck's last norm + norm, shortcut and ReLU in the same launch)
  and crn_batch_renorm_bwd_head (d pre formed by the backward launch of that norm) against the two-launch sequences they
  replace -- crn_batch_renorm_stats + crn_affine_add_relu, crn_relu_bwd_add + crn_batch_renorm_bwd -- bit for bit, and the
  forward against the contract emulator."""
  g = t.Generator().manual_seed(C + S)
  dev = DEV
  x = (t.randn(B, C, S, generator=g) * 1.5 + 0.2).to(dev)
  r = t.randn(B, C, S, generator=g).to(dev)
  rsc, rsh = ((t.rand(C, generator=g) + 0.5).to(dev), t.randn(C, generator=g).to(dev)) if shortcut == "affine" else (None, None)
  gamma, beta = (t.rand(C, generator=g) + 0.5).to(dev), t.randn(C, generator=g).to(dev)
  rm0, rv0 = t.randn(C, generator=g).to(dev), (t.rand(C, generator=g) * 3 + 0.1).to(dev)
  nb = t.tensor([nbt], dtype=t.int64, device=dev)
  outs = []
  for fused in (False, True):
    rm, rv = rm0.clone(), rv0.clone()
    sc, sh, sv = t.zeros(C, device=dev), t.zeros(C, device=dev), t.zeros(4 * C, device=dev)
    y = t.full((B, C, S), -7.0, device=dev)
    ypre = t.full((B, C + 3, S), -7.0, device=dev) if second else None       # a wider buffer, like the skip feature maps
    if fused:
      be.bn_stats_tail(x, B, C, S, C * S, gamma, beta, rm, rv, nb, 1e-3, 0.01, True, sc, sh, sv, r, rsc, rsh, C * S,
                       ypre, (C + 3) * S if second else 0, y, C * S, True)
    else:
      be.bn_stats(x, B, C, S, C * S, False, gamma, beta, rm, rv, nb, 1e-3, 0.01, True, sc, sh, sv)
      be.affine_add_relu(x, sc, sh, r, rsc, rsh, B, C, S, C * S, C * S, ypre, (C + 3) * S if second else 0, y, C * S, True)
    # backward: gradient of y, activation = y (or the pre-ReLU copy), optional second gradient
    gy = t.randn(B, C, S, generator=t.Generator().manual_seed(5)).to(dev)
    g2 = t.randn(B, C + 3, S, generator=t.Generator().manual_seed(6)).to(dev) if second else None
    act, sBa = (ypre, (C + 3) * S) if second else (y, C * S)
    dpre = t.full((B, C, S), 3.0, device=dev); dx = t.zeros(B, C, S, device=dev)
    dg, db, ds = t.zeros(C, device=dev), t.zeros(C, device=dev), t.zeros(C, device=dev)
    if fused:
      be.bn_bwd_head(x, C * S, dpre, C * S, gy, C * S, act, sBa, g2, (C + 3) * S if second else 0, B, C, S, gamma, sc, sh,
                     sv, dx, C * S, dg, db, dsum=ds, ndsum=C)
    else:
      be.relu_bwd_add(gy, act, g2, B, C, S, C * S, sBa, (C + 3) * S if second else 0, dpre, C * S)
      be.bn_bwd(x, C * S, dpre, C * S, B, C, S, False, False, gamma, sc, sh, sv, dx, C * S, dg, db, dsum=ds, ndsum=C)
    t.cuda.synchronize()
    outs.append([v.clone() for v in (sc, sh, sv, rm, rv, y, dpre, dx, dg, db, ds)] + ([ypre.clone()] if second else []))
  names = ["scale", "shift", "saved", "rmean", "rvar", "y", "d pre", "dx", "dgamma", "dbeta", "dsum", "y_pre"]
  for a, b, nm in zip(outs[0], outs[1], names):
    if nm == "dsum" and B * S > 16384:      # (the two-pass form adds sum(dx) up with atomics: order differs from run to run)
      assert float((a - b).abs().max()) <= 1e-4 * (float(outs[0][7].abs().sum((0, 2)).max()) + 1e-6)
      continue
    assert t.equal(a, b), (nm, float((a - b).abs().max()))
  if second:
    assert float(outs[1][-1][:, C:].min()) == -7.0 and float(outs[1][-1][:, C:].max()) == -7.0      # the extra channels
  # forward against the emulator of the two contracts
  xc, rc = x.cpu(), r.cpu()
  sc, sh, sv = t.zeros(C), t.zeros(C), t.zeros(4 * C)
  EMU.bn_stats(xc, B, C, S, C * S, False, gamma.cpu(), beta.cpu(), rm0.cpu().clone(), rv0.cpu().clone(), nb.cpu(), 1e-3, 0.01,
               True, sc, sh, sv)
  want = t.zeros(B, C, S)
  EMU.affine_add_relu(xc, sc, sh, rc, rsc.cpu() if rsc is not None else None, rsh.cpu() if rsh is not None else None,
                      B, C, S, C * S, C * S, None, 0, want, C * S, True)
  close(outs[1][5], want, 2e-5, "tail vs emulator")
  # the gradient g in the COMPACT form of a stride-2 data gradient (g_compact argument of crn_batch_renorm_bwd_head): equals the call
  # on crn_stride2_scatter(g_compact), whichever path expands it (the register kernel on the fly, or a scatter launch into `g`)
  for W in ([64, 16] if S in (4096, 256) else [6, 4] if S == 36 else [128]):
    if S % W:
      continue
    H = S // W
    gcmp = t.randn(B, C, (H + 1) // 2, (W + 1) // 2, generator=t.Generator().manual_seed(9)).to(dev)
    gfull = t.zeros(B, C, H, W, device=dev); gfull[:, :, ::2, ::2] = gcmp
    act, sBa = outs[1][5], C * S
    res = []
    for compact in (False, True):
      dpre = t.full((B, C, S), 3.0, device=dev); dx = t.zeros(B, C, S, device=dev)
      dg, db, ds = t.zeros(C, device=dev), t.zeros(C, device=dev), t.zeros(C, device=dev)
      gbuf = gfull.clone() if not compact else t.full((B, C, H, W), 5.0, device=dev)      # (compact: a buffer the call may expand into)
      be.bn_bwd_head(x, C * S, dpre, C * S, gbuf, C * S, act, sBa, None, 0, B, C, S, gamma, outs[1][0], outs[1][1], outs[1][2],
                     dx, C * S, dg, db, dsum=ds, ndsum=C, g_compact=gcmp if compact else None, W=W)
      res.append((dpre, dx, dg, db))
    for a, b_, nm in zip(res[0], res[1], ("d pre", "dx", "dgamma", "dbeta")):
      assert t.equal(a, b_), (W, nm, float((a - b_).abs().max()))
  # the stride-2 compaction of y for the down-sampling block that follows (y2 argument): equals crn_stride2_gather of y, whichever
  # path wrote it (the tail launch itself for planes with W % 4 == 0, a gather launch otherwise)
  for W in ([64, 16] if S in (4096, 256) else [6, 4] if S == 36 else [128]):
    W = W if S % W == 0 else None
    if W is None:
      continue
    H = S // W
    rm, rv = rm0.clone(), rv0.clone()
    sc2, sh2, sv2 = t.zeros(C, device=dev), t.zeros(C, device=dev), t.zeros(4 * C, device=dev)
    y = t.full((B, C, S), -7.0, device=dev)
    y2 = t.full((B, C, (H + 1) // 2, (W + 1) // 2), -7.0, device=dev)
    be.bn_stats_tail(x, B, C, S, C * S, gamma, beta, rm, rv, nb, 1e-3, 0.01, True, sc2, sh2, sv2, r, rsc, rsh, C * S,
                     None, 0, y, C * S, True, y2=y2, W=W)
    assert t.equal(y, outs[1][5]), W
    assert t.equal(y2, y.view(B, C, H, W)[:, :, ::2, ::2]), (W, float((y2 - y.view(B, C, H, W)[:, :, ::2, ::2]).abs().max()))


@pytest.mark.parametrize("m,B,C,dims", [(1, 3, 2, (5, 6, 7)), (2, 2, 14, (4, 5, 6)), (3, 1, 5, (3, 4, 5)), (2, 1, 2, (16, 16, 16))])
def test_softmax_superres(be, m, B, C, dims):
  g = t.Generator().manual_seed(m * 10 + C)
  D, H, W = dims
  logits = t.randn(m ** 3, B, C, D, H, W, generator=g) * 4
  want = t.empty(B, C, m * D, m * H, m * W)
  EMU.softmax_superres(logits, m, B, C, D, H, W, want)
  got = t.full((B, C, m * D, m * H, m * W), -1.0, device=DEV)
  be.softmax_superres(logits.to(DEV), m, B, C, D, H, W, got)
  close(got, want, 2e-6, "softmax_superres")
  assert float((got.sum(1) - 1).abs().max()) < 1e-5


def test_elementwise(be):
  g = t.Generator().manual_seed(9)
  B, C, S = 2, 6, 100
  x = t.randn(B, C, S, generator=g); r = t.randn(B, C, S, generator=g)
  sc, sh, rsc, rsh = [t.randn(C, generator=g) for _ in range(4)]
  out = {}
  for dev, bk in (("cpu", EMU), (DEV, be)):
    pre = t.zeros(B, C + 3, S, device=dev); y = t.zeros(B, C, S, device=dev)
    bk.affine_add_relu(x.to(dev), sc.to(dev), sh.to(dev), r.to(dev), rsc.to(dev), rsh.to(dev), B, C, S, C * S, C * S,
                       pre, (C + 3) * S, y, C * S, True)
    dx = t.zeros(B, C, S, device=dev)
    bk.relu_bwd_add(r.to(dev), pre, x.to(dev), B, C, S, C * S, (C + 3) * S, C * S, dx, C * S)
    db = t.zeros(C, device=dev)
    bk.bias_grad(x.to(dev), B, C, S, C * S, db)
    out[dev] = [v.cpu() for v in (pre, y, dx, db)]
  for a, b in zip(out[DEV], out["cpu"]):
    close(a, b, 1e-6)


def test_encoder_misc(be):
  g = t.Generator().manual_seed(10)
  B, C, H = 2, 8, 16
  img = t.randint(0, 256, (B, 3, 12, 20), generator=g, dtype=t.uint8)
  x = t.randn(B, C, H, H, generator=g); sc = t.rand(C, generator=g) + 0.5; sh = t.randn(C, generator=g) * 0.2
  dy = t.randn(B, C, H // 2, H // 2, generator=g)
  w = t.randn(5, 32, generator=g); bias = t.randn(5, generator=g); xl = t.randn(B, 32, generator=g)
  off = t.rand(B, 3, generator=g)
  out = {}
  for dev, bk in (("cpu", EMU), (DEV, be)):
    f = t.zeros(B, 3, 12, 20, device=dev); bk.preprocess(img.to(dev), f)
    y = t.zeros(B, C, H // 2, H // 2, device=dev); am = t.zeros(B, C, H // 2, H // 2, dtype=t.int32, device=dev)
    bk.maxpool_fwd(x.to(dev), sc.to(dev), sh.to(dev), B, C, H, H, y, am)
    dx = t.zeros(B, C, H, H, device=dev); bk.maxpool_bwd(dy.to(dev), am, B, C, H, H, dx)
    avg = t.zeros(B, C, device=dev); bk.relu_mean_fwd(x.to(dev), B, C, H * H, C * H * H, avg)
    dm = t.ones(B, C, H * H, device=dev); bk.relu_mean_bwd(x.to(dev), avg, B, C, H * H, C * H * H, dm, C * H * H, True)
    yl = t.zeros(B, 8, device=dev); bk.linear_fwd(xl.to(dev), w.to(dev), bias.to(dev), B, 32, 5, yl, 8)
    dxl = t.zeros(B, 32, device=dev); dw = t.zeros(5, 32, device=dev); dbb = t.zeros(5, device=dev)
    bk.linear_bwd(xl.to(dev), w.to(dev), yl, 8, B, 32, 5, dxl, dw, dbb)
    fo = t.zeros(B, 7, 9, device=dev); bk.fill_offset_channels(fo, B, 63, 9, 4, off.to(dev))
    out[dev] = [v.cpu().float() for v in (f, y, am, dx, avg, dm, yl, dxl, dw, dbb, fo)]
  for i, (a, b) in enumerate(zip(out[DEV], out["cpu"])):
    close(a, b, 1e-6, f"misc[{i}]")
  ref = t.nn.functional.max_pool2d(t.nn.functional.pad((x * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1)).relu(), [1] * 4), 3, 2)
  close(out[DEV][1], ref, 1e-6, "maxpool vs torch")


def test_stride2_gather_scatter(be):
  """crn_stride2_gather / crn_stride2_scatter (the compacted input of the ResNet downscale blocks' stride-2 1x1
  convs and the adjoint): exact copies; the scatter writes every element (zeros between the samples)."""
  g = t.Generator().manual_seed(2)
  # (B, C, input h, input w): even extents with even / odd output widths (224 x 224 images reach 14 -> 7), odd input extents
  for B, C, hin, win in ((2, 5, 16, 16), (4, 256, 64, 64), (1, 3, 12, 20), (2, 7, 14, 14), (2, 3, 7, 10), (1, 2, 9, 5)):
    h, w = (hin + 1) // 2, (win + 1) // 2
    x = t.randn(B, C, hin, win, generator=g)
    y = t.full((B, C, h, w), 9.0, device=DEV)
    be.stride2_gather(x.to(DEV), y)
    assert t.equal(y.cpu(), x[:, :, ::2, ::2])
    dy = t.randn(B, C, h, w, generator=g)
    dx = t.full((B, C, hin, win), 9.0, device=DEV)
    be.stride2_scatter(dy.to(DEV), dx)
    want = t.zeros(B, C, hin, win); want[:, :, ::2, ::2] = dy
    assert t.equal(dx.cpu(), want)


def test_decoder_inputs_one_launch(be):
  """crn_decoder_inputs: the four layer matrices v2s . scale(128 / r) (reconstruction_decoder.py:111-116) and the offset copy of a
  call in one launch -- bit for bit what the torch expression of the contract gives (a column scaling is exact)."""
  g = t.Generator().manual_seed(3)
  for B in (1, 4, 7):
    v2s, off = t.randn(B, 4, 4, generator=g), t.rand(B, 3, generator=g)
    scales = [8.0, 4.0, 2.0, 1.0]
    want_m, want_o = t.zeros(4, B, 16), t.zeros(B, 3)
    EMU.decoder_inputs(v2s, off, scales, want_m, want_o)
    lm, oo = t.full((4, B, 16), 9.0, device=DEV), t.full((B, 3), 9.0, device=DEV)
    be.decoder_inputs(v2s.to(DEV), off.to(DEV), scales, lm, oo)
    assert t.equal(lm.cpu(), want_m) and t.equal(oo.cpu(), want_o)
    assert t.equal(want_m[1].view(B, 4, 4)[:, :, 3], v2s[:, :, 3]) and t.equal(want_m[1].view(B, 4, 4)[:, :, :3], v2s[:, :, :3] * 4.0)


# ------------------------------------------------------------------ ray-traced skip
def test_ray_sample_golden_and_edge(be, golden_dir):
  import os
  z = np.load(os.path.join(golden_dir, "sample_grid2d.npz"))
  src, w, b = t.tensor(z["src"]), t.tensor(z["weight"]), t.tensor(z["bias"])
  mats, off = t.tensor(z["mats"]), t.tensor(z["off"])
  cmap = t.nn.functional.conv2d(src, w, b)
  B, C = cmap.shape[:2]
  out = t.full((B, C + 2, 16, 16, 16), 7.0, device=DEV)
  be.ray_sample_fwd(cmap.to(DEV), C * 256, B, C, 16, 16, mats.reshape(B, 16).to(DEV), off.to(DEV),
                    out[:, 2:], (C + 2) * 4096, 16, 16, 16)
  assert int((out[:, 2:].cpu() != t.tensor(z["y"])).sum()) == 0        # bit-exact incl. edge-case camera
  assert float((out[:, :2] - 7.0).abs().max()) == 0
  if C % 4 == 0:   # channel-last map [B][h][w][C] (dwordx4 gathers), same result bit for bit
    out2 = t.full((B, C + 2, 16, 16, 16), 7.0, device=DEV)
    be.ray_sample_fwd(cmap.permute(0, 2, 3, 1).contiguous().to(DEV), C * 256, B, C, 16, 16, mats.reshape(B, 16).to(DEV),
                      off.to(DEV), out2[:, 2:], (C + 2) * 4096, 16, 16, 16, map_sC=1, map_sP=C)
    assert t.equal(out2, out)
  gy = t.tensor(z["gy"])
  dmap = t.zeros(B, C, 16, 16, device=DEV)
  be.ray_sample_bwd(gy.to(DEV), C * 4096, B, C, 16, 16, 16, mats.reshape(B, 16).to(DEV), off.to(DEV), dmap, C * 256,
                    16, 16, True)
  cm = cmap.clone().requires_grad_(True)
  O.ray_sample(cm, mats, off, (16, 16, 16)).backward(gy)
  close(dmap, cm.grad, 1e-5, "ray bwd")


@pytest.mark.parametrize("res,C", [(8, 96), (16, 48), (32, 24), (64, 12)])
def test_ray_sample_decoder_scales(be, res, C):
  """Index parity (mismatch count 0) at the four decoder scales, canonical camera, B=2."""
  g = t.Generator().manual_seed(res)
  B = 2
  cmap = t.randn(B, C, res, res, generator=g)
  m = (O.canonical_camera() @ O.scale([1.0 / 128] * 3) @ O.scale([128.0 / res] * 3))[None].expand(B, 4, 4).contiguous()
  off = t.tensor([[0.5, 0.5, 0.5], [0.25, 0.5, 0.75]])
  out = t.zeros(B, C, res, res, res, device=DEV)
  be.ray_sample_fwd(cmap.to(DEV), C * res * res, B, C, res, res, m.reshape(B, 16).to(DEV), off.to(DEV), out,
                    C * res ** 3, res, res, res)
  ref = O.ray_sample(cmap, m, off, (res,) * 3)
  assert int((out.cpu() != ref).sum()) == 0
  # the layout the model uses: channel-last map written by the 1x1 compress conv
  out2 = t.zeros(B, C, res, res, res, device=DEV)
  be.ray_sample_fwd(cmap.permute(0, 2, 3, 1).contiguous().to(DEV), C * res * res, B, C, res, res,
                    m.reshape(B, 16).to(DEV), off.to(DEV), out2, C * res ** 3, res, res, res, map_sC=1, map_sP=C)
  assert int((out2.cpu() != ref).sum()) == 0


@pytest.mark.parametrize("res,C", [(8, 96), (16, 48), (32, 24), (64, 12), (16, 10)])
def test_ray_sample_backward_cameras_and_accumulate(be, res, C):
  """crn_ray_sample_bwd against the oracle's autograd (index_put_ accumulate) at the four decoder scales:
  sample 0 canonical camera, sample 1 shifted so that part of the grid leaves the image and a slab is behind the
  camera, sample 2 rolled 20 degrees about the optical axis (general
  matrix), incl. accumulation into an existing gradient (zero_first = False)."""
  g = t.Generator().manual_seed(res + C)
  B = 3
  base = O.canonical_camera() @ O.scale([1.0 / 128] * 3) @ O.scale([128.0 / res] * 3)
  shift = O.translate([0.9, -0.4, -0.95]) @ base
  a = np.deg2rad(20.0)
  roll = t.tensor([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=t.float32) @ base
  m = t.stack([base, shift, roll])
  off = t.tensor([[0.5, 0.5, 0.5], [0.25, 0.5, 0.75], [0.5, 0.5, 0.5]])
  cmap = t.randn(B, C, res, res, generator=g).requires_grad_(True)
  gy = t.randn(B, C, res, res, res, generator=g)
  y = O.ray_sample(cmap, m, off, (res,) * 3)
  y.backward(gy)
  dmap = t.full((B, C, res, res), 5.0, device=DEV)
  be.ray_sample_bwd(gy.to(DEV), C * res ** 3, B, C, res, res, res, m.reshape(B, 16).to(DEV), off.to(DEV), dmap,
                    C * res * res, res, res, True)
  for b in range(B):
    close(dmap[b], cmap.grad[b], 2e-5, f"ray bwd sample {b}")
  assert float(cmap.grad[1].abs().sum()) > 0 and float((y[1] == 0).float().mean()) > 0.05     # the edge case is live
  prev = t.randn(B, C, res, res, generator=g)
  dm2 = prev.to(DEV).clone()
  be.ray_sample_bwd(gy.to(DEV), C * res ** 3, B, C, res, res, res, m.reshape(B, 16).to(DEV), off.to(DEV), dm2,
                    C * res * res, res, res, False)
  close(dm2, prev + cmap.grad, 2e-5, "ray bwd accumulate")


def _ray_cameras(res):
  base = O.canonical_camera() @ O.scale([1.0 / 128] * 3) @ O.scale([128.0 / res] * 3)
  shift = O.translate([0.9, -0.4, -0.95]) @ base
  a = np.deg2rad(20.0)
  roll = t.tensor([[np.cos(a), -np.sin(a), 0, 0], [np.sin(a), np.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=t.float32) @ base
  return t.stack([base, shift, roll]), t.tensor([[0.5, 0.5, 0.5], [0.25, 0.5, 0.75], [0.5, 0.5, 0.5]])


@pytest.mark.parametrize("res,C,hw", [(8, 96, 8), (16, 48, 16), (32, 24, 32), (64, 12, 64), (16, 10, 16), (32, 12, 80), (20, 8, 12)])
def test_ray_saved_index_tensor_and_scatter(be, res, C, hw):
  """The saved index tensor (crn_ray_sample_fwd_idx / crn_ray_project) against the oracle's index arithmetic, bit for bit
  (ray_traced_skip_connection.py:118-133; 0xFFFF = outside value), and crn_ray_sample_bwd_idx against the oracle's autograd
  (index_put_ accumulate, :135) on three cameras, incl. maps that are larger / smaller than the grid, a grid whose width is
  not a multiple of the tile (20) and accumulation into an existing gradient."""
  g = t.Generator().manual_seed(res * 7 + C)
  B = 3
  m, off = _ray_cameras(res)
  cmap = t.randn(B, C, hw, hw, generator=g).requires_grad_(True)
  gy = t.randn(B, C, res, res, res, generator=g)
  y = O.ray_sample(cmap, m, off, (res,) * 3)
  y.backward(gy)
  want = EmuBackend.ray_indices_u16(m.reshape(B, 16), off, B, res, res, res, hw, hw)
  md, od = m.reshape(B, 16).to(DEV), off.to(DEV)
  idx = t.full((B, res ** 3), 77, dtype=t.int16, device=DEV)
  be.ray_project(md, od, B, res, res, res, hw, hw, idx)
  assert t.equal(idx.cpu().to(t.int64).view(B, res, res, res) & 0xFFFF, want)
  idx2 = t.full((B, res ** 3), 77, dtype=t.int16, device=DEV)
  out = t.zeros(B, C, res, res, res, device=DEV)
  be.ray_sample_fwd_idx(cmap.detach().permute(0, 2, 3, 1).contiguous().to(DEV), C * hw * hw, B, C, hw, hw, md, od, out, C * res ** 3,
                        res, res, res, idx2, map_sC=1, map_sP=C)
  assert t.equal(idx2, idx) and int((out.cpu() != y.detach()).sum()) == 0
  dmap = t.full((B, C, hw, hw), 5.0, device=DEV)
  be.ray_sample_bwd_idx(gy.to(DEV), C * res ** 3, B, C, res, res, res, idx, dmap, C * hw * hw, hw, hw, True)
  for b in range(B):
    close(dmap[b], cmap.grad[b], 2e-5, f"ray bwd (saved indices) sample {b}")
  prev = t.randn(B, C, hw, hw, generator=g)
  dm2 = prev.to(DEV).clone()
  be.ray_sample_bwd_idx(gy.to(DEV), C * res ** 3, B, C, res, res, res, idx, dm2, C * hw * hw, hw, hw, False)
  close(dm2, prev + cmap.grad, 2e-5, "ray bwd (saved indices) accumulate")
  # integer-valued gradients: every sum is exact in fp32 whatever the order -> bit-exact against index_put_
  gi = t.randint(-8, 9, gy.shape, generator=g).float()
  cm2 = cmap.detach().clone().requires_grad_(True)
  O.ray_sample(cm2, m, off, (res,) * 3).backward(gi)
  dm3 = t.zeros(B, C, hw, hw, device=DEV)
  be.ray_sample_bwd_idx(gi.to(DEV), C * res ** 3, B, C, res, res, res, idx, dm3, C * hw * hw, hw, hw, True)
  assert t.equal(dm3.cpu(), cm2.grad), "integer gradients: the scatter must be exact"


def test_ray_index_entry_points_reject_large_maps(be):
  """h*w >= 65535 does not fit the 16-bit index tensor: the _idx entry points say so (CRN_EINVAL), the plain backward takes 32-bit
  indices in its own scratch and still matches the oracle."""
  from corenet_amd._lib import HipError
  B, C, res, hw = 1, 4, 8, 256
  m, off = _ray_cameras(res)
  m, off = m[:1], off[:1]
  idx = t.zeros(B, res ** 3, dtype=t.int16, device=DEV)
  with pytest.raises(HipError):
    be.ray_project(m.reshape(B, 16).to(DEV), off.to(DEV), B, res, res, res, hw, hw, idx)
  g = t.Generator().manual_seed(1)
  cmap = t.randn(B, C, hw, hw, generator=g).requires_grad_(True)
  gy = t.randn(B, C, res, res, res, generator=g)
  O.ray_sample(cmap, m, off, (res,) * 3).backward(gy)
  dmap = t.zeros(B, C, hw, hw, device=DEV)
  be.ray_sample_bwd(gy.to(DEV), C * res ** 3, B, C, res, res, res, m.reshape(B, 16).to(DEV), off.to(DEV), dmap, C * hw * hw, hw, hw, True)
  close(dmap, cmap.grad, 2e-5, "ray bwd 32-bit indices")


def _probe_lib():
  import ctypes
  from corenet_amd import _lib
  if not os.path.exists(_lib.PROBE_LIB_PATH):
    pytest.skip("tools/_build/libcrn_probe.so not built (python -m corenet_amd.build --tools)")
  return ctypes.CDLL(_lib.PROBE_LIB_PATH)


def _conv_aggressor(be, key, direction):
  """One bf16x3 launch of the plan on its real shape at B = 2 (the layer keys of tools/bench_conv.py), as a callable."""
  from corenet_amd.model import conv_geometry as G
  from corenet_amd import views as V
  from corenet_amd.backend import Transform
  B = 2
  g = t.Generator().manual_seed(5)
  if key == "s6t1par":        # the parity-walk kernels of the 14-class logits layer (csrc/convt_par.hip)
    w = t.randn(16, 14, 7, 7, 7, generator=g) * 0.05
    x = t.randn(B, 16, 64, 64, 64, generator=g).to(DEV); y = t.randn(B, 14, 128, 128, 128, generator=g).to(DEV)
    fn = G.convt_par_fwd_table if direction == "fwd" else G.convt_par_dgrad_table
    tab, nbytes = fn(tuple(w.shape), 0)
    img = t.zeros(nbytes, dtype=t.uint8, device=DEV)
    be.bf3_gather_image(w.reshape(-1).to(DEV), t.as_tensor(tab).to(DEV), img)
    tr = Transform((t.rand(16, generator=g) + 0.5).to(DEV), t.randn(16, generator=g).to(DEV), pre_relu=True)
    if direction == "fwd":
      return lambda: be.convt_par_fwd(x, tr, img, None, y, 14)
    return lambda: be.convt_par_dgrad(y, 14, img, x, False)
  kind, wshape, pad, dims = {"s6c1": ("conv", (16, 28, 5, 5, 5), 2, (64, 64, 64)), "s6t1": ("convT", (16, 2, 7, 7, 7), 3, (64, 64, 64)),
                             "s5t1": ("convT", (32, 16, 7, 7, 7), 3, (32, 32, 32))}[key]
  if kind == "conv":
    cin, cout = wshape[1], wshape[0]; fwd, dgr = G.conv_fwd(wshape, pad), G.conv_dgrad(wshape, pad); odims = dims
  else:
    cin, cout = wshape[0], wshape[1]; fwd, dgr = G.convt_fwd(wshape, pad), G.convt_dgrad(wshape, pad); odims = tuple(2 * d for d in dims)
  x = t.randn((B, cin) + dims, generator=g).to(DEV); y = t.randn((B, cout) + odims, generator=g).to(DEV)
  w = t.randn(wshape, generator=g) * 0.05
  wf, wd = EMU_pack(w, fwd).to(DEV), EMU_pack(w, dgr).to(DEV)
  tr = Transform((t.rand(cin, generator=g) + 0.5).to(DEV), t.randn(cin, generator=g).to(DEV), pre_relu=True)
  yv = V.space_to_depth_view(V.view_of(y), (2, 2, 2), parity_major=True) if kind == "convT" else V.view_of(y)
  if direction == "fwd":
    return lambda: be.conv_fwd(V.view_of(x), tr, wf, fwd.npad, None, 0, yv, fwd.window, fwd.pad_lo, 0, boxes=(fwd.n_boxes, fwd.c_boxes), math="bf16x3")
  return lambda: be.conv_fwd(yv, None, wd, dgr.npad, None, 0, V.view_of(x), dgr.window, dgr.pad_lo, 0, boxes=(dgr.n_boxes, dgr.c_boxes), math="bf16x3")


@pytest.mark.parametrize("neighbour", ["probe 1", "probe 44", "probe 48", "fwd s6c1", "fwd s6t1", "dgrad s6t1", "fwd s5t1", "dgrad s5t1",
                                       "fwd s6t1par", "dgrad s6t1par"])
def test_ray_scatter_beside_mfma_neighbours(be, neighbour):
  """The 64^3 ray-sample scatter on a side stream BESIDE an MFMA-dense neighbour on the main stream, 30 runs per neighbour and
  start delay, each compared with the oracle's index_put_ gradient (ray_traced_skip_connection.py:135) to 2e-5.  Round 4 found that
  dependent-MFMA chains of a neighbour wave (the probe's modes 1, 44-48; the split-bf16 convolutions before their products became
  one block of adjacent MFMAs) made loop-invariant projection products of the round 1-4 scatter go missing in 28-29 of 30 runs
  (DESIGN section 3e).  The scatter no longer projects (saved index tensor, integer compares + float adds only): this test holds
  the victim side; the plain entry point (projection launch + scatter) and the gather + index tensor run beside the same
  neighbours."""
  import ctypes
  from corenet_amd import _lib
  B, Cs, res = 2, 12, 64
  g = t.Generator().manual_seed(0)
  gu = (t.randn(B, 28, res, res, res, generator=g) * 1e-6)
  # sample 0: the canonical camera; sample 1: a rolled, shifted one -- a DENSE matrix.  Round 6 found that the glitch only shows
  # where a lost product is not a product with one of the canonical camera's zeros (tools/project_glitch.py: crn_ray_project beside
  # probe modes 1 / 44 / 48 is wrong in 20 of 20 runs, lanes 48-63, sample 1 only): a victim test on the canonical camera alone is blind
  cams, offs = _ray_cameras(res)
  m = t.stack([cams[0], cams[2] @ O.translate([0.02, -0.03, 0.01])]).contiguous()
  off = t.stack([offs[0], offs[1]])
  cmap = t.randn(B, Cs, res, res, generator=g).requires_grad_(True)
  yref = O.ray_sample(cmap, m, off, (res,) * 3)
  yref.backward(gu[:, 16:])
  want, scale = cmap.grad, float(cmap.grad.abs().max())
  want_idx = EmuBackend.ray_indices_u16(m.reshape(B, 16), off, B, res, res, res, res, res)
  gud, md, od = gu.to(DEV), m.reshape(B, 16).to(DEV), off.to(DEV)
  cl = cmap.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
  idx = t.zeros(B, res ** 3, dtype=t.int16, device=DEV)
  be.ray_project(md, od, B, res, res, res, res, res, idx)
  gmap, gmap2 = t.zeros(B, Cs, res, res, device=DEV), t.zeros(B, Cs, res, res, device=DEV)
  out = t.zeros(B, Cs, res, res, res, device=DEV)
  idx2 = t.zeros_like(idx)
  if neighbour.startswith("probe"):
    probe, mode = _probe_lib(), int(neighbour.split()[1])
    sink = t.zeros(16, device=DEV)
    def aggressor():
      assert probe.crn_mfma_probe(mode, 3000, 256, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(_lib.stream())) == 0
  else:
    aggressor = _conv_aggressor(be, neighbour.split()[1], neighbour.split()[0])
  side = t.cuda.Stream()
  bad = []
  idx3 = t.zeros_like(idx)
  probe_neighbour = neighbour.startswith("probe")
  for delay in (0, 20000):
    for i in range(30):
      gmap.fill_(3.0); gmap2.fill_(3.0); idx2.zero_(); out.zero_(); idx3.zero_()
      t.cuda.synchronize()
      ev = t.cuda.Event(); ev.record()
      aggressor()
      with t.cuda.stream(side), _lib.pinned_stream(side):
        side.wait_event(ev)
        if delay: t.cuda._sleep(delay)
        be.ray_sample_bwd_idx(gud[:, 16:], gud.stride(0), B, Cs, res, res, res, idx, gmap, gmap.stride(0), res, res, True)
        be.ray_sample_bwd(gud[:, 16:], gud.stride(0), B, Cs, res, res, res, md, od, gmap2, gmap2.stride(0), res, res, True)
        be.ray_sample_fwd_idx(cl, cl.stride(0), B, Cs, res, res, md, od, out, out.stride(0), res, res, res, idx2, map_sC=1, map_sP=Cs)
        be.ray_project(md, od, B, res, res, res, res, res, idx3)
      t.cuda.synchronize()
      e1 = float((gmap.cpu() - want).abs().max()) / scale
      e2 = float((gmap2.cpu() - want).abs().max()) / scale
      ok3 = t.equal(idx2.cpu().to(t.int64).view(B, res, res, res) & 0xFFFF, want_idx) and int((out.cpu() != yref.detach()).sum()) == 0
      ok4 = t.equal(idx3.cpu().to(t.int64).view(B, res, res, res) & 0xFFFF, want_idx)
      # The scatter from saved indices (e1) has no projection and must be exact beside EVERYTHING.  The kernels that project
      # (e2 = projection launch + scatter, ok3 = gather + index tensor, ok4 = crn_ray_project) must be exact beside every
      # launch of the LIBRARY -- whose MFMA streams are the shapes that never disturb a neighbour (mfma3 blocks, round-robin
      # accumulators) --; beside the probe's deliberately hostile chains they show the hardware's behaviour and are reported
      if e1 > 2e-5 or (not probe_neighbour and (e2 > 2e-5 or not ok3 or not ok4)):
        bad.append((delay, i, e1, e2, ok3, ok4))
      elif probe_neighbour and (e2 > 2e-5 or not ok3 or not ok4):
        print(f"(hostile probe neighbour {neighbour}, delay {delay}, run {i}: projecting kernels off -- e2 {e2:.1e} gather {ok3} project {ok4})")
  assert not bad, (neighbour, len(bad), bad[:5])


def _scatter_vs_index_put(be, gy, m, off, res, C, hw, tag):
  """crn_ray_sample_bwd_idx against the oracle's autograd index_put_ (ray_traced_skip_connection.py:135) on gradients that hold
  inf / NaN / zeros / extreme magnitudes: non-finite POSITIONS equal (and of the same kind), finite elements within 2e-5 of
  what the pixel's contributions sum to in absolute value (index_put_ of |gy|: the conditioning-aware scale of each pixel)."""
  B = gy.shape[0]
  cmap = t.zeros(B, C, hw, hw).requires_grad_(True)
  with np.errstate(all="ignore"):
    O.ray_sample(cmap, m, off, (res,) * 3).backward(gy)
    want = cmap.grad.clone()
    fin = t.where(t.isfinite(gy), gy.abs(), t.zeros_like(gy))
    cm2 = t.zeros(B, C, hw, hw).requires_grad_(True)
    O.ray_sample(cm2, m, off, (res,) * 3).backward(fin)
    wabs = cm2.grad
  md, od = m.reshape(B, 16).to(DEV), off.to(DEV)
  idx = t.zeros(B, res ** 3, dtype=t.int16, device=DEV)
  be.ray_project(md, od, B, res, res, res, hw, hw, idx)
  for zero_first in (True, False):
    dmap = t.full((B, C, hw, hw), 0.0 if not zero_first else 9.0, device=DEV)
    be.ray_sample_bwd_idx(gy.to(DEV), C * res ** 3, B, C, res, res, res, idx, dmap, C * hw * hw, hw, hw, zero_first)
    got = dmap.cpu()
    assert t.equal(t.isnan(got), t.isnan(want)), (tag, int(t.isnan(got).sum()), int(t.isnan(want).sum()))
    assert t.equal(t.isposinf(got), t.isposinf(want)) and t.equal(t.isneginf(got), t.isneginf(want)), tag
    f = t.isfinite(want)
    err = (got[f].double() - want[f].double()).abs()
    bar = 2e-5 * wabs[f].double() + 1e-44
    assert bool((err <= bar).all()), (tag, zero_first, float((err / bar).max()))
  return want


def test_ray_scatter_fallback_branches(be):
  """The branches of ray_scatter_kernel beside its fixed-point window (csrc/ray_sample.hip): a workgroup whose largest |gradient| is
  inf / NaN, or below 2^-83, or zero, adds to HBM in float -- the reference's autograd `index_put_(accumulate=True)`
  (ray_traced_skip_connection.py:135) propagates non-finite values and loses nothing of tiny ones.  (i) one +inf, one -inf and one
  NaN element (in different tiles, and a pixel that receives +inf and -inf), (ii) an all-zero 64^3 gradient, (iii) gradients of
  magnitude 1e-30, (iv) mixed tiles: one 32 x 8-column tile of magnitude 1e+20 among tiles of 1e-20, (v) one huge tile among zeros;
  64^3 x 12 (the tiled form) and a 16^3 x 48 map (8 x 8 tiles), canonical and shifted cameras."""
  for res, C, hw in ((64, 12, 64), (16, 48, 16)):
    g = t.Generator().manual_seed(res)
    B = 2
    cams, offs = _ray_cameras(res)
    m, off = cams[:B], offs[:B]
    shape = (B, C, res, res, res)
    base = t.randn(shape, generator=g)
    # (i) non-finite elements
    gy = base.clone()
    gy[0, 1, 3, 5, 7] = float("inf"); gy[0, 2, res - 2, res - 3, res - 4] = float("-inf"); gy[1, 0, res // 2, 9, 11] = float("nan")
    gy[1, 3, 2, 4, 4] = float("inf"); gy[1, 3, 3, 4, 4] = float("-inf")      # neighbours in z: the same pixel or adjacent ones
    want = _scatter_vs_index_put(be, gy, m, off, res, C, hw, f"non-finite {res}")
    assert int((~t.isfinite(want)).sum()) >= 3
    # (ii) all zeros (M = 0: nothing to scale by)
    w0 = _scatter_vs_index_put(be, t.zeros(shape), m, off, res, C, hw, f"zeros {res}")
    assert float(w0.abs().max()) == 0.0
    # (iii) tiny gradients: 1e-30 is below the window's 2^-83 floor, nothing may be flushed to zero
    wt = _scatter_vs_index_put(be, base * 1e-30, m, off, res, C, hw, f"tiny {res}")
    assert float(wt.abs().max()) > 1e-30
    # (iv) / (v) one tile huge, its neighbours tiny / zero: per-workgroup scales differ by 2^133
    tx, ty = (32, 8) if res >= 64 else (8, 8)
    for other in (1e-20, 0.0):
      gm = base * other
      gm[:, :, :8, ty:2 * ty, tx % res:(tx % res) + tx] = base[:, :, :8, ty:2 * ty, tx % res:(tx % res) + tx] * 1e20
      _scatter_vs_index_put(be, gm, m, off, res, C, hw, f"mixed {other} {res}")


def test_ray_scatter_width_one_map(be):
  """A skip map of width 1 (the stage-5 map of a 64 x 32 image, which check_image_hw accepts): ceil(2^32 / w) does not fit the
  16-bit path's reciprocal; the host scatters the [h][1] map as the [1][h] map it is in memory.  Gather indices bit
  for bit, scatter against index_put_, for (h, w) = (2, 1), (8, 1), (1, 1) and (1, 8)."""
  res, C, B = 8, 96, 2
  cams, offs = _ray_cameras(res)
  m, off = cams[:B], offs[:B]
  g = t.Generator().manual_seed(3)
  for h, w in ((2, 1), (8, 1), (1, 1), (1, 8)):
    cmap = t.randn(B, C, h, w, generator=g).requires_grad_(True)
    gy = t.randn(B, C, res, res, res, generator=g)
    O.ray_sample(cmap, m, off, (res,) * 3).backward(gy)
    md, od = m.reshape(B, 16).to(DEV), off.to(DEV)
    idx = t.zeros(B, res ** 3, dtype=t.int16, device=DEV)
    be.ray_project(md, od, B, res, res, res, h, w, idx)
    assert t.equal(idx.cpu().to(t.int64).view(B, res, res, res) & 0xFFFF, EmuBackend.ray_indices_u16(m.reshape(B, 16), off, B, res, res, res, h, w))
    dmap = t.full((B, C, h, w), 4.0, device=DEV)
    be.ray_sample_bwd_idx(gy.to(DEV), C * res ** 3, B, C, res, res, res, idx, dmap, C * h * w, h, w, True)
    close(dmap, cmap.grad, 2e-5, f"scatter into a {h} x {w} map")
    dm2 = t.zeros(B, C, h, w, device=DEV)
    be.ray_sample_bwd(gy.to(DEV), C * res ** 3, B, C, res, res, res, md, od, dm2, C * h * w, h, w, True)
    close(dm2, cmap.grad, 2e-5, f"plain backward into a {h} x {w} map")


@pytest.mark.parametrize("mode", [1, 44, 48])
def test_side_stream_victims_beside_mfma_probe(be, mode):
  """Round 4's glitch made VALU results of a wave go missing when dependent-MFMA chains ran on the same SIMD; the
  only victim ever seen was the old scatter's projection.  The other kernels the plan runs on the side stream beside split-bf16
  convolutions get the same treatment here: the two-pass BatchRenorm backward (bn_bwd_partial / bn_bwd_apply_kernel), the loss's
  second pass (loss_pass2_kernel), Adam from device scalars (adam_hyper_kernel) and crn_ray_project (which still evaluates the
  loop-free projection), each on a side stream BESIDE the probe's failing family on the main stream, 30 runs, against the CPU
  contract (kernel_contract_emu) computed once: BatchRenorm / loss gradients 2e-5, Adam 1e-6, indices bit for bit."""
  if _SELF:
    return
  import ctypes
  from corenet_amd import _lib
  probe = _probe_lib()
  sink = t.zeros(16, device=DEV)
  g = t.Generator().manual_seed(mode)
  # BatchRenorm backward, decoder shape (two-pass form): [2, 16, 64^3]
  B, C, S = 2, 16, 64 ** 3
  x = t.randn(B, C, S, generator=g) * 2 + 0.4; gy = t.randn(B, C, S, generator=g)
  gamma, beta = t.rand(C, generator=g) + 0.5, t.randn(C, generator=g)
  rm0, rv0 = t.randn(C, generator=g), t.rand(C, generator=g) * 3 + 0.1
  nb = t.tensor([30000], dtype=t.int64)
  def bn_state(dev, bk):
    sc, sh, sv = t.zeros(C, device=dev), t.zeros(C, device=dev), t.zeros(4 * C, device=dev)
    bk.bn_stats(x.to(dev), B, C, S, C * S, True, gamma.to(dev), beta.to(dev), rm0.clone().to(dev), rv0.clone().to(dev), nb.to(dev),
                1e-3, 0.01, True, sc, sh, sv)
    return sc, sh, sv
  sc, sh, sv = bn_state("cpu", EMU)
  dx_w, dg_w, db_w = t.zeros(B, C, S), t.zeros(C), t.zeros(C)
  EMU.bn_bwd(x, C * S, gy, C * S, B, C, S, True, False, gamma, sc, sh, sv, dx_w, C * S, dg_w, db_w)
  xd, gyd, gammad = x.to(DEV), gy.to(DEV), gamma.to(DEV)
  scd, shd, svd = bn_state(DEV, be)
  dx, dg, db = t.zeros(B, C, S, device=DEV), t.zeros(C, device=DEV), t.zeros(C, device=DEV)
  # loss: iou_fgbg and xent_times_iou_agnostic on [2, C, 64^3]
  Sl = 64 ** 3
  loss_cases = []
  for Cn, kind in ((2, 0), (14, 4)):
    logits = t.randn(2, Cn, 64, 64, 64, generator=g) * 2; gt = t.randint(0, Cn, (2, 64, 64, 64), generator=g)
    lw, dlw = t.zeros(1), t.zeros(2, Cn, 64, 64, 64)
    EMU.loss_fwd_bwd(kind, logits, gt.to(t.int32), 2, Cn, Sl, lw, dlw, 1.0)
    loss_cases.append((kind, Cn, logits.to(DEV), gt.to(t.int32).to(DEV), float(lw), dlw, t.zeros(1, device=DEV),
                       t.zeros(2, Cn, 64, 64, 64, device=DEV)))
  # Adam from device scalars
  n = 1 << 22
  p0, gr = t.randn(n, generator=g), t.randn(n, generator=g)
  m0, v0 = t.randn(n, generator=g) * 0.1, t.rand(n, generator=g) * 0.01
  pw, mw, vw, hy = p0.clone(), m0.clone(), v0.clone(), t.zeros(8)
  EMU.adam_set_hyper(hy, 4e-4, 0.9, 0.999, 1e-4, 0.5, 7)
  EMU.adam_step_hyper(pw, gr, mw, vw, n, hy)
  hyd = t.zeros(8, device=DEV)
  be.adam_set_hyper(hyd, 4e-4, 0.9, 0.999, 1e-4, 0.5, 7)
  grd = gr.to(DEV)
  # projection
  res = 64
  cams, offs = _ray_cameras(res)
  mc, oc = cams[:2], offs[:2]
  want_idx = EmuBackend.ray_indices_u16(mc.reshape(2, 16), oc, 2, res, res, res, res, res)
  mcd, ocd = mc.reshape(2, 16).to(DEV), oc.to(DEV)
  idx = t.zeros(2, res ** 3, dtype=t.int16, device=DEV)
  side = t.cuda.Stream()
  bad = []
  proj_off = 0
  for i in range(30):
    dx.zero_(); dg.zero_(); db.zero_(); idx.zero_()
    pd, md_, vd = p0.to(DEV), m0.to(DEV), v0.to(DEV)
    for c in loss_cases:
      c[6].zero_(); c[7].zero_()
    t.cuda.synchronize()
    ev = t.cuda.Event(); ev.record()
    assert probe.crn_mfma_probe(mode, 6000, 256, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(_lib.stream())) == 0
    with t.cuda.stream(side), _lib.pinned_stream(side):
      side.wait_event(ev)
      if i % 2: t.cuda._sleep(20000)
      be.ray_project(mcd, ocd, 2, res, res, res, res, res, idx)
      be.bn_bwd(xd, C * S, gyd, C * S, B, C, S, True, False, gammad, scd, shd, svd, dx, C * S, dg, db)
      for kind, Cn, lg, gtd, _, _, lossd, dld in loss_cases:
        be.loss_fwd_bwd(kind, lg, gtd, 2, Cn, Sl, lossd, dld, 1.0)
      be.adam_step_hyper(pd, grd, md_, vd, n, hyd)
    t.cuda.synchronize()
    def rel(a, b):
      return float((a.cpu().double() - b.double()).abs().max()) / (float(b.abs().max()) + 1e-30)
    errs = {"bn dx": rel(dx, dx_w), "bn dgamma": rel(dg, dg_w), "bn dbeta": rel(db, db_w), "adam p": rel(pd, pw), "adam m": rel(md_, mw),
            "adam v": rel(vd, vw)}
    for kind, Cn, _, _, lv, dlw, lossd, dld in loss_cases:
      errs[f"loss{kind} dl"] = rel(dld, dlw)
      errs[f"loss{kind} value"] = abs(float(lossd) - lv) / max(1.0, abs(lv)) * 2.0     # (bar 1e-5: scaled onto the 2e-5 bar)
    bars = {k: (1e-6 if k.startswith("adam") else 2e-5) for k in errs}
    off_ = {k: v for k, v in errs.items() if not v <= bars[k]}
    if not t.equal(idx.cpu().to(t.int64).view(2, res, res, res) & 0xFFFF, want_idx):
      proj_off += 1
    if off_:
      bad.append((i, off_))
  # crn_ray_project is the one kernel that DOES go wrong beside these hostile chains (20 of 20 runs, lanes 48-63, the dense
  # camera only: tools/project_glitch.py, profiles/r06_project_glitch.txt) -- the hardware's behaviour, reproduced at will since
  # round 6; what the product relies on is test_ray_scatter_beside_mfma_neighbours: exact beside every launch of the library
  print(f"probe mode {mode}: crn_ray_project off in {proj_off} of 30 runs beside the hostile chain (reported, not asserted)")
  assert not bad, (mode, len(bad), bad[:3])


@pytest.mark.parametrize("Cin,N,hw", [(2048, 96, 8), (256, 12, 64), (96, 20, 16)])
def test_pointwise_conv_channel_last_output(be, Cin, N, hw):
  """1x1 conv writing a channel-last view (compress_channels -> skip map), with and without split-K, against
  the same conv into a plain view."""
  from corenet_amd.model import conv_geometry as G
  from corenet_amd import views as V
  g = t.Generator().manual_seed(Cin)
  B = 2
  x = t.randn(B, Cin, hw, hw, generator=g).to(DEV)
  w = t.randn(N, Cin, 1, 1, generator=g) / np.sqrt(Cin)
  geo = G.conv_fwd(tuple(w.shape), 0)
  wp = EMU_pack(w, geo)
  b_ref = t.randn(N, generator=g)
  bias = EMU_pack(b_ref, None, G.bias_index(N, 1, geo.npad, parity_major=False))
  y0 = t.zeros(B, N, hw, hw, device=DEV); y1 = t.full((B, hw, hw, N), 3.0, device=DEV)
  for y in (V.view_of(y0), V.view_of(y1.permute(0, 3, 1, 2))):
    be.conv_fwd(V.view_of(x), None, wp.to(DEV), geo.npad, bias.to(DEV), 0, y, geo.window, geo.pad_lo, 0, False,
                boxes=(geo.n_boxes, geo.c_boxes))
  ref = t.nn.functional.conv2d(x.cpu(), w, b_ref)
  close(y0, ref, 2e-5, "pointwise plain")
  assert t.equal(y1.permute(0, 3, 1, 2), y0)


# ------------------------------------------------------------------ losses / metrics / adam
@pytest.mark.parametrize("C,kind", [(2, 0), (5, 0), (5, 1), (14, 1), (5, 2), (5, 3), (5, 4)])
def test_losses(be, C, kind):
  g = t.Generator().manual_seed(C * 10 + kind)
  B, dims = 2, (6, 7, 8)
  S = int(np.prod(dims))
  logits = t.randn(B, C, *dims, generator=g) * 2
  gt = t.randint(0, C, (B,) + dims, generator=g)
  name = EmuBackend.LOSSES[kind]
  l = logits.clone().requires_grad_(True)
  v = getattr(O, name)(gt, l); v.backward()
  loss = t.zeros(1, device=DEV); dl = t.zeros(B, C, *dims, device=DEV)
  be.loss_fwd_bwd(kind, logits.to(DEV), gt.to(t.int32).to(DEV), B, C, S, loss, dl, 1.0)
  assert abs(float(loss) - float(v)) <= 1e-5 * max(1.0, abs(float(v)))
  close(dl, l.grad, 2e-5, name)


@pytest.mark.parametrize("C,kind", [(2, 0), (5, 1), (14, 1), (5, 2), (5, 3), (5, 4)])
def test_losses_weighted(be, C, kind):
  """Per-voxel loss weights (losses.py:47-49,99-102,134-136): value and gradient against the oracle's autograd."""
  g = t.Generator().manual_seed(C * 10 + kind + 100)
  B, dims = 3, (5, 9, 8)
  S = int(np.prod(dims))
  logits = t.randn(B, C, *dims, generator=g) * 2
  gt = t.randint(0, C, (B,) + dims, generator=g)
  w = t.rand((B,) + dims, generator=g) * 1.5
  w[0, 0] = 0                                   # weight 0 switches voxels off
  name = EmuBackend.LOSSES[kind]
  l = logits.clone().requires_grad_(True)
  v = getattr(O, name)(gt, l, w); v.backward()
  loss = t.zeros(1, device=DEV); dl = t.zeros(B, C, *dims, device=DEV)
  be.loss_fwd_bwd(kind, logits.to(DEV), gt.to(t.int32).to(DEV), B, C, S, loss, dl, 1.0, weights=w.to(DEV))
  assert abs(float(loss) - float(v)) <= 1e-5 * max(1.0, abs(float(v)))
  close(dl, l.grad, 2e-5, name)


def test_losses_golden_from_reference(be):
  """tests/golden/losses.npz: values and gradients produced by the reference's own losses (oracle/gen_golden.py),
  with and without per-voxel weights, all five loss kinds."""
  z = np.load(os.path.join(os.path.dirname(__file__), "golden", "losses.npz"))
  logits, gt, w = t.tensor(z["logits"]), t.tensor(z["gt"]).to(t.int32), t.tensor(z["weights"])
  B, C = logits.shape[:2]
  S = logits[0, 0].numel()
  for kind, name in EmuBackend.LOSSES.items():
    for suffix, ww in (("", None), ("_w", w)):
      loss = t.zeros(1, device=DEV); dl = t.zeros(logits.shape, device=DEV)
      be.loss_fwd_bwd(kind, logits.to(DEV), gt.to(DEV), B, C, S, loss, dl, 1.0, weights=None if ww is None else ww.to(DEV))
      np.testing.assert_allclose(float(loss), float(z[name + suffix]), rtol=1e-5, atol=1e-6)
      close(dl, t.tensor(z[name + suffix + "_grad"]), 2e-5, name + suffix)


def test_losses_reference_known_answers(be):
  """All six known answers of the reference's test/losses_test.py:25-88 (three of them weighted), through the
  C ABI and through the drop-in functions of corenet_amd.model.losses (autograd, int64 labels)."""
  from reference_known_answers import LOSS_LOGITS, LOSS_GT, LOSS_WEIGHTS
  logits = t.tensor(LOSS_LOGITS).permute(0, 4, 1, 2, 3).contiguous()
  gt = t.tensor(LOSS_GT, dtype=t.int32)
  w = t.tensor(LOSS_WEIGHTS)
  cases = ((2, None, 0.8060565), (0, None, 0.3579613), (3, None, 1.4547757),
           (2, w, 0.8174121), (0, w, 0.4265449), (3, w, 0.7043564))
  for kind, ww, want in cases:
    loss = t.zeros(1, device=DEV)
    be.loss_fwd_bwd(kind, logits.to(DEV), gt.to(DEV), 2, 4, 12, loss, None, 1.0,
                    weights=None if ww is None else ww.to(DEV))
    np.testing.assert_allclose(float(loss), want, rtol=1e-5, atol=1e-6)
  if _SELF:
    return
  from corenet_amd.model import losses
  for kind, ww, want in cases:
    fn = getattr(losses, EmuBackend.LOSSES[kind])
    l = logits.to(DEV).requires_grad_(True)
    v = fn(gt.long().to(DEV), l, None if ww is None else ww.to(DEV))
    np.testing.assert_allclose(float(v), want, rtol=1e-5, atol=1e-6)
    (2 * v).backward()
    lo = logits.clone().requires_grad_(True)
    (2 * getattr(O, EmuBackend.LOSSES[kind])(gt.long(), lo, ww)).backward()
    close(l.grad, lo.grad, 2e-5, EmuBackend.LOSSES[kind])
  # labels outside [0, C): the reference raises inside F.one_hot; here opt-in (needs a read-back)
  bad = gt.long().clone(); bad[0, 0, 0, 0] = 4
  os.environ["CRN_CHECK_LABELS"] = "1"
  try:
    with pytest.raises(ValueError):
      losses.iou_fgbg(bad.to(DEV), logits.to(DEV))
    losses.iou_fgbg(gt.long().to(DEV), logits.to(DEV))
  finally:
    del os.environ["CRN_CHECK_LABELS"]
  assert np.isfinite(float(losses.xent(bad.to(DEV), logits.to(DEV))))


def test_argmax_confusion_and_adam(be):
  g = t.Generator().manual_seed(5)
  B, C, S = 2, 5, 1000
  logits = t.randn(B, C, S, generator=g); gt = t.randint(0, C, (B, S), generator=g).to(t.int32)
  lab = t.zeros(B, S, dtype=t.int32, device=DEV); cm = t.zeros(C * C, dtype=t.int64, device=DEV)
  be.argmax_confusion(logits.to(DEV), gt.to(DEV), B, C, S, lab, cm)
  assert t.equal(lab.cpu().long(), logits.argmax(1))
  assert t.equal(cm.cpu().view(C, C), O.confusion_matrix(gt, logits.argmax(1), C))
  n = 1003
  p = t.randn(n, generator=g); gr = t.randn(n, generator=g)
  pt = p.clone().requires_grad_(True); opt = t.optim.Adam([pt], lr=4e-4, eps=1e-4)
  pg, m, v = p.to(DEV), t.zeros(n, device=DEV), t.zeros(n, device=DEV)
  for step in range(1, 4):
    pt.grad = gr * step; opt.step()
    be.adam_step(pg, (gr * step * 2).to(DEV), m, v, n, 4e-4, 0.9, 0.999, 1e-4, 0.5, step)
  close(pg, pt.detach(), 1e-6, "adam")


# ------------------------------------------------------------------ ground-truth side
def _shells(N, R, radii, dtype=np.float32):
  zz, yy, xx = np.meshgrid(np.arange(R), np.arange(R), np.arange(R), indexing="ij")
  g = np.zeros((N, R, R, R), dtype)
  for n in range(N):
    r = radii[n % len(radii)]
    d = np.sqrt((xx - R / 2 + 0.5) ** 2 + (yy - R / 2 + 0.5) ** 2 + (zz - R / 2 + 0.5) ** 2)
    g[n][(d <= r) & (d > r - 1.5)] = 1
  return g


def test_fill_known_answers(be):
  from reference_known_answers import fill_grids
  from corenet_amd.cc import fill_voxels
  g1, g2, e1, e2 = fill_grids()
  for dt in (t.float32, t.uint8, t.int32, t.float64, t.int64):
    grid = t.tensor(np.stack([g1, g2])).to(dt).to(DEV)
    out = fill_voxels.fill_inside_voxels_gpu(grid, inplace=False)
    assert out.data_ptr() != grid.data_ptr() and out.dtype == dt
    np.testing.assert_array_equal(out.cpu().numpy(), np.stack([e1, e2]).astype(out.cpu().numpy().dtype))
    out2 = fill_voxels.fill_inside_voxels_gpu(grid, inplace=True)
    assert out2.data_ptr() == grid.data_ptr()
    np.testing.assert_array_equal(grid.cpu().numpy(), np.stack([e1, e2]).astype(out.cpu().numpy().dtype))
  with pytest.raises(ValueError):
    fill_voxels.fill_inside_voxels_gpu(t.zeros(2, 2, 2, 2), inplace=False)          # CPU tensor
  with pytest.raises(ValueError):
    fill_voxels.fill_inside_voxels_gpu(t.zeros(2, 2, 2, device=DEV), inplace=False)  # rank 3


@pytest.mark.parametrize("shape,seed", [((3, 5, 6, 7), 0), ((2, 33, 31, 65), 1), ((1, 9, 9, 130), 2),
                                        ((4, 64, 64, 64), 3), ((2, 7, 7, 7), 4)])
def test_fill_random_bit_exact(be, shape, seed):
  import fill_oracle_c
  rng = np.random.RandomState(seed)
  for dens in (0.2, 0.45, 0.7):
    g = (rng.rand(*shape) < dens).astype(np.float32)
    g[0].flat[::7] = -3.0                    # non-positive values count as empty (data > 0)
    out = t.empty(shape, device=DEV)
    be.fill_voxels(t.tensor(g).to(DEV), out)
    np.testing.assert_array_equal(out.cpu().numpy(), fill_oracle_c.fill(g))
    if shape[-1] % 64 == 0:                  # the 16-byte load path also serves int32 grids
      gi = t.tensor(g).to(t.int32).to(DEV); oi = t.empty_like(gi)
      be.fill_voxels(gi, oi)
      np.testing.assert_array_equal(oi.cpu().numpy(), fill_oracle_c.fill(g).astype(np.int32))


def test_fill_y1_sub_grid_65(be):
  """The y1 eval configuration (generate_configs.py:205-208: 32^3 grid, sub-grid sampling): conservative
  sub-grid voxelization into (2*32+1)^3 = 65^3 grids, fill on the 65^3 grids (odd sizes, W not a multiple of 64),
  centres extracted: HIP chain == oracle chain on every voxel, and the fill alone bit-exact on random 65^3 grids."""
  import fill_oracle_c
  from corenet_amd.cc import fill_voxels
  from corenet_amd.data import batched_example
  from corenet_amd.geometry import voxelization
  rng = np.random.RandomState(65)
  g = (rng.rand(3, 65, 65, 65) < 0.35).astype(np.float32)
  out = t.empty(g.shape, device=DEV)
  be.fill_voxels(t.tensor(g).to(DEV), out)
  np.testing.assert_array_equal(out.cpu().numpy(), fill_oracle_c.fill(g))
  if _SELF:
    return
  R = 32
  tris = np.concatenate([_uv_sphere(24, 48, np.array([0.45, 0.5, 0.5]), 0.25), _uv_sphere(24, 48, np.array([0.6, 0.55, 0.5]), 0.2)])
  nt = [24 * 48 * 2] * 2
  v2v = batched_example.view2voxel_matrices(t.full((1, 3), 0.5), (R, R, R))[0]
  kw = dict(sub_grid_sampling=True, image_resolution_multiplier=9, conservative_rasterization=True)
  gg = voxelization.voxelize_mesh(t.tensor(tris), nt, (R, R, R), v2v, **kw)
  assert gg.shape == (2, 65, 65, 65)
  ref = O.voxelize_mesh(tris, nt, (R, R, R), v2v.numpy(), **kw)
  np.testing.assert_array_equal(gg.cpu().numpy(), ref)
  filled = fill_voxels.fill_inside_voxels_gpu(gg)
  np.testing.assert_array_equal(filled.cpu().numpy(), fill_oracle_c.fill(ref))
  c = voxelization.get_sub_grid_centers(filled).cpu().numpy()
  np.testing.assert_array_equal(c, O.get_sub_grid_centers(fill_oracle_c.fill(ref)))
  assert c.shape == (2, 32, 32, 32) and c.sum() > 1000
  labels = batched_example.voxelize_labels(t.tensor(tris), [t.tensor(nt, dtype=t.int32)], [[2, 7]], t.full((1, 3), 0.5), (R, R, R),
                                           **kw).cpu().numpy()
  np.testing.assert_array_equal(labels, O.merge_labels(O.get_sub_grid_centers(fill_oracle_c.fill(ref)), [2], [[2, 7]]))


def test_fill_full_size_shells(be):
  """BASELINE sizes: 12 x 128^3 hollow shells -> solid balls, low-face semantics, idempotence."""
  import fill_oracle_c
  g = _shells(12, 128, (10, 30, 50))
  g[5, 100:, 100:, 100:] = 0; g[5, 99, 99:, 99:] = 1; g[5, 99:, 99, 99:] = 1; g[5, 99:, 99:, 99] = 1   # high-face pocket
  grid = t.tensor(g).to(DEV)
  out = t.empty_like(grid)
  be.fill_voxels(grid, out)
  o = out.cpu().numpy()
  np.testing.assert_array_equal(o[:6], fill_oracle_c.fill(g[:6]))
  assert o[5, 100:, 100:, 100:].min() == 1
  out2 = t.empty_like(grid); be.fill_voxels(out, out2)
  assert t.equal(out, out2)                                  # idempotent
  assert set(np.unique(o)) <= {0.0, 1.0}


def test_fill_serpentine_rescue_and_multi_launch_paths(be):
  """A corridor that snakes up and down z behind every wall needs hundreds of slab exchanges: the single-launch
  kernel runs them on its ring of self-cleaning control words (bit-exact, also in place); with the round limit
  forced down (CRN_FILL_MAXROUNDS) a workgroup gives up, raises the launch's flag, and the last workgroup to leave
  the launch redoes its grids alone (no second launch, no host involvement): same answer.
  The rescue path and the any-size path (one persistent workgroup per grid on global bitmaps, which replaced the
  multi-launch sweeps and their host-side convergence check) are also run on their own (CRN_FILL_RESCUE /
  CRN_FILL_MULTI) in fresh processes, on random grids of three dtypes."""
  import fill_oracle_c, subprocess, sys, os
  D, H, W = 64, 6, 64
  g = np.zeros((2, D, H, W), np.float32)
  g[:, 0] = 1; g[:, :, 0] = 1                  # close the z=0 and y=0 faces: seeds only on x=0
  for i, x in enumerate(range(2, W - 1, 2)):
    g[:, :, :, x] = 1
    g[:, (D - 1) if i % 2 == 0 else 1, 1:, x] = 0      # gap alternates between the top and the bottom
  g[1, 30:40, 2:5, 50:60] = 1                  # plus a solid block and an enclosed pocket
  g[1, 33:36, 3, 53:56] = 0
  want = fill_oracle_c.fill(g)
  out = t.empty(g.shape, device=DEV)
  be.fill_voxels(t.tensor(g).to(DEV), out)
  np.testing.assert_array_equal(out.cpu().numpy(), want)
  inpl = t.tensor(g).to(DEV)
  be.fill_voxels(inpl, inpl)
  np.testing.assert_array_equal(inpl.cpu().numpy(), want)
  if _SELF:
    return
  code = ("import sys, numpy as np, torch as t; sys.path.insert(0, %r); sys.path.insert(0, %r);"
          "import fill_oracle_c; from corenet_amd.backend import HipBackend; be = HipBackend();"
          "rng = np.random.RandomState(5);\n"
          "for shape in ((3, 40, 33, 70), (2, 128, 128, 128), (5, 7, 9, 11)):\n"
          "  g = (rng.rand(*shape) < 0.4).astype(np.float32); want = fill_oracle_c.fill(g)\n"
          "  for dt in (t.float32, t.uint8, t.int64):\n"
          "    x = t.tensor(g).to(dt).cuda(); out = t.empty_like(x); be.fill_voxels(x, out)\n"
          "    assert (out.cpu().numpy() == want.astype(out.cpu().numpy().dtype)).all(), (shape, dt)\n"
          "    be.fill_voxels(x, x); assert t.equal(x, out)\n"
          "print('path ok')"
          % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
  # CRN_FILL_MAXROUNDS=3: most of these grids need more exchange rounds than that -- a workgroup gives up, the last one out
  # redoes the grids and clears the self-cleaning control blocks for the calls that follow in the same process
  for var, val in (("CRN_FILL_MULTI", "1"), ("CRN_FILL_RESCUE", "1"), ("CRN_FILL_MAXROUNDS", "3")):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{var: val}), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "path ok" in r.stdout, (var, r.stderr[-2000:])


@pytest.mark.parametrize("shape", [(2, 5, 9, 640), (1, 3, 600, 600), (1, 2, 70, 1100), (3, 20, 20, 513)])
def test_fill_any_size_bit_exact(be, shape):
  """The reference op has no size limit (fill_voxels_gpu.cu:136-171): rows wider than 512 voxels and planes whose
  bitmaps exceed the LDS budget go to the one-workgroup-per-grid kernel on bitmaps in the workspace (round 2 returned
  CRN_EINVAL for W > 512).  Random grids at three densities plus a serpentine corridor across the wide axis, in place
  and out of place, two dtypes: bit-exact against the C oracle."""
  import fill_oracle_c
  rng = np.random.RandomState(sum(shape))
  N, D, H, W = shape
  grids = [(rng.rand(*shape) < d).astype(np.float32) for d in (0.25, 0.45, 0.65)]
  snake = np.zeros(shape, np.float32)
  snake[:, 0] = 1; snake[:, :, 0] = 1                      # seeds only on x = 0
  for i, x in enumerate(range(2, W - 1, 2)):
    snake[:, :, :, x] = 1
    snake[:, :, (H - 1) if i % 2 == 0 else 1, x] = 0       # the gap alternates between the two ends of y
  for g in grids + [snake]:
    want = fill_oracle_c.fill(g)
    out = t.empty(shape, device=DEV)
    be.fill_voxels(t.tensor(g).to(DEV), out)
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    gi = t.tensor(g).to(t.uint8).to(DEV)
    be.fill_voxels(gi, gi)
    np.testing.assert_array_equal(gi.cpu().numpy(), want.astype(np.uint8))


def test_fill_strided_views_like_packed_accessors(be):
  """The reference op reads and writes through packed accessors (fill_voxels_gpu.cu:146-163), so a non-contiguous tensor is
  filled where it lies: `crn_fill_voxels_strided` (no `.contiguous()` copy).  Permuted, sliced and step-2 views, out of place
  (fresh contiguous result, input untouched) and in place (the caller's view is mutated, the elements between its strides are
  not), through `get_module()` like cc/fill_voxels.py:102-107; 128-wide rows (the contiguous twin would take the 16-byte
  path), a grid that needs the any-size kernel (W > 512) and the in-launch rescue path (subprocess, CRN_FILL_MAXROUNDS)."""
  if _SELF:
    return
  import fill_oracle_c, subprocess, sys, os
  from corenet_amd.cc import fill_voxels as fv
  mod = fv.get_module()
  rng = np.random.RandomState(11)
  base = (rng.rand(3, 40, 36, 140) < 0.42).astype(np.float32)
  views = [lambda x: x.permute(0, 1, 3, 2), lambda x: x.permute(0, 3, 2, 1), lambda x: x[:, ::2, 1:, ::3],
           lambda x: x[1:, :, :, 5:133], lambda x: x.permute(1, 0, 2, 3)]
  for dt in (t.float32, t.uint8, t.int64):
    for mk in views:
      full = t.tensor(base).to(dt).to(DEV)
      v = mk(full)
      assert not v.is_contiguous()
      want = fill_oracle_c.fill(v.cpu().contiguous().numpy().astype(np.float32))
      keep = full.clone()
      out = mod.fill_inside_voxels_gpu(v, False)
      assert out.is_contiguous() and out.data_ptr() != full.data_ptr() and t.equal(full, keep)
      np.testing.assert_array_equal(out.cpu().numpy(), want.astype(out.cpu().numpy().dtype))
      r = mod.fill_inside_voxels_gpu(v, True)
      assert r.data_ptr() == v.data_ptr() and r.stride() == v.stride()
      np.testing.assert_array_equal(v.cpu().numpy(), want.astype(out.cpu().numpy().dtype))
      # the elements the view does not cover are untouched
      mask = t.zeros_like(full, dtype=t.bool); mk(mask)[...] = True
      assert t.equal(full[~mask], keep[~mask])
  wide = (rng.rand(2, 4, 520, 6) < 0.4).astype(np.float32)          # viewed as W = 520: the any-size kernel
  wv = t.tensor(wide).to(DEV).permute(0, 1, 3, 2)
  np.testing.assert_array_equal(fv.fill_inside_voxels_gpu(wv).cpu().numpy(), fill_oracle_c.fill(wv.cpu().contiguous().numpy()))
  ex = t.tensor(base[:1, :1]).to(DEV).expand(2, 40, 36, 140)          # stride-0 dimensions: readable, not writable in place
  np.testing.assert_array_equal(fv.fill_inside_voxels_gpu(ex).cpu().numpy(), fill_oracle_c.fill(ex.cpu().contiguous().numpy()))
  with pytest.raises(ValueError):
    fv.fill_inside_voxels_gpu(ex, inplace=True)
  if _SELF:
    return
  code = ("import sys, numpy as np, torch as t; sys.path.insert(0, %r); sys.path.insert(0, %r);"
          "import fill_oracle_c; from corenet_amd.cc import fill_voxels as fv;"
          "rng = np.random.RandomState(5);\n"
          "for shape in ((3, 40, 70, 33), (2, 64, 128, 128)):\n"
          "  g = (rng.rand(*shape) < 0.4).astype(np.float32); x = t.tensor(g).cuda().permute(0, 1, 3, 2)\n"
          "  want = fill_oracle_c.fill(x.cpu().contiguous().numpy())\n"
          "  assert (fv.fill_inside_voxels_gpu(x).cpu().numpy() == want).all(), shape\n"
          "  fv.fill_inside_voxels_gpu(x, inplace=True); assert (x.cpu().numpy() == want).all(), shape\n"
          "print('path ok')"
          % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))))
  for var, val in (("CRN_FILL_MULTI", "1"), ("CRN_FILL_RESCUE", "1"), ("CRN_FILL_MAXROUNDS", "3")):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{var: val}), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "path ok" in r.stdout, (var, r.stderr[-2000:])


def test_fill_is_asynchronous_graph_capturable(be):
  """crn_fill_voxels never waits for the GPU (the reference op does not either, fill_voxels_gpu.cu:158-165): the
  whole call -- one kernel launch -- is captured into a HIP graph (a host-side
  synchronisation inside the call would abort the capture) and replayed on new contents of the same buffers,
  including a serpentine grid that needs hundreds of exchange rounds."""
  if _SELF:
    return
  import fill_oracle_c
  rng = np.random.RandomState(9)
  shape = (4, 64, 64, 64)
  x = t.zeros(shape, device=DEV); out = t.zeros(shape, device=DEV)
  be.fill_voxels(x, out)                       # warm-up outside the capture (workspace allocation, attributes)
  t.cuda.synchronize()
  side = t.cuda.Stream()
  graph = t.cuda.CUDAGraph()
  with t.cuda.stream(side):
    with t.cuda.graph(graph, stream=side):
      be.fill_voxels(x, out)
  t.cuda.synchronize()
  grids = [(rng.rand(*shape) < d).astype(np.float32) for d in (0.3, 0.5)]
  snake = np.zeros(shape, np.float32)
  snake[:, 0] = 1; snake[:, :, 0] = 1
  for i, xx in enumerate(range(2, 63, 2)):
    snake[:, :, :, xx] = 1
    snake[:, 63 if i % 2 == 0 else 1, 1:, xx] = 0
  for g in grids + [snake]:
    x.copy_(t.tensor(g).to(DEV))
    graph.replay()
    t.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), fill_oracle_c.fill(g))


def test_voxelizer_known_answers(be):
  from corenet_amd.cc import fill_voxels
  from corenet_amd.geometry import voxelization
  from test_oracle_cpu import _cube
  quad = t.tensor([[[0, 0, 0], [1, 0, 1], [0, 1, 0]], [[1, 0, 1], [0, 1, 0], [1, 1, 1]]], dtype=t.float32)
  grid = voxelization.voxelize_mesh(quad, [2], (4, 4, 4), O.scale([4, 4, 4]), image_resolution_multiplier=16)
  fill_voxels.fill_inside_voxels_gpu(grid, inplace=True)
  e = np.zeros((4, 4, 4), np.float32)
  for i in range(4): e[i, :, i] = 1
  np.testing.assert_array_equal(grid.cpu().numpy(), e[None])
  cube = t.tensor(_cube(0.99))
  g = voxelization.voxelize_mesh(cube, [12], (3, 3, 3), t.eye(4), image_resolution_multiplier=1)
  e = np.zeros((3, 3, 3), np.float32); e[1, 1, [0, 2]] = e[1, [0, 2], 1] = e[[0, 2], 1, 1] = 1
  np.testing.assert_array_equal(g.cpu().numpy(), e[None])
  g = voxelization.voxelize_mesh(cube, [12], (3, 3, 3), t.eye(4), image_resolution_multiplier=1,
                                 conservative_rasterization=True)
  e = np.ones((3, 3, 3), np.float32); e[1, 1, 1] = 0
  np.testing.assert_array_equal(g.cpu().numpy(), e[None])
  g = voxelization.voxelize_mesh(cube, [12], (3, 3, 3), t.eye(4), sub_grid_sampling=True,
                                 image_resolution_multiplier=9, conservative_rasterization=True)
  g = fill_voxels.fill_inside_voxels_gpu(g, inplace=False)
  e = np.zeros((1, 7, 7, 7), np.float32); e[0, 2:5, 2:5, 2:5] = 1
  np.testing.assert_array_equal(g.cpu().numpy(), e)
  cubes = t.cat([cube, cube - 0.5])
  tr = t.stack([O.translate([-0.5, 0, 0]), O.translate([0.5, 1, 1])])
  g = voxelization.voxelize_mesh(cubes, [12, 12], (3, 3, 3), tr, sub_grid_sampling=True,
                                 image_resolution_multiplier=9, conservative_rasterization=True)
  c = voxelization.get_sub_grid_centers(fill_voxels.fill_inside_voxels_gpu(g)).cpu().numpy()
  e1 = np.zeros((3, 3, 3)); e1[1, 1, [0, 1]] = 1
  e2 = np.zeros((3, 3, 3)); e2[1, [1, 2], 1] = e2[2, [1, 2], 1] = 1
  np.testing.assert_array_equal(c[0], e1); np.testing.assert_array_equal(c[1], e2)
  with pytest.raises(ValueError):
    voxelization.voxelize_mesh(cube, [12], (3, 3, 3), t.eye(4), sub_grid_sampling=True, image_resolution_multiplier=8)


def _uv_sphere(n_lat, n_lon, center, radius):
  th = np.linspace(0, np.pi, n_lat + 1); ph = np.linspace(0, 2 * np.pi, n_lon + 1)
  p = lambda a, b: center + radius * np.array([np.sin(a) * np.cos(b), np.sin(a) * np.sin(b), np.cos(a)])
  tris = []
  for i in range(n_lat):
    for j in range(n_lon):
      a, b, c, d = p(th[i], ph[j]), p(th[i + 1], ph[j]), p(th[i + 1], ph[j + 1]), p(th[i], ph[j + 1])
      tris += [[a, b, c], [a, c, d]]
  return np.array(tris, np.float32)


def test_voxelizer_sphere_vs_oracle_and_labels(be):
  """Sphere meshes at 32^3: the HIP rasterizer and the oracle perform the same fp32 operations in the same
  order (oracle.voxelize_mesh docstring), so they must agree on EVERY voxel; after fill: a solid ball; label
  merge: larger class id wins on overlap (Q11), bit-exact."""
  from corenet_amd.data import batched_example
  R = 32
  tris = np.concatenate([_uv_sphere(24, 48, np.array([0.45, 0.5, 0.5]), 0.25),
                         _uv_sphere(24, 48, np.array([0.6, 0.5, 0.5]), 0.2)])
  nt = [24 * 48 * 2, 24 * 48 * 2]
  off = t.full((1, 3), 0.5)
  v2v = batched_example.view2voxel_matrices(off, (R, R, R))
  np.testing.assert_allclose(v2v.numpy(), O.view2voxel_matrices(off, (R, R, R)).numpy())
  from corenet_amd.geometry import voxelization
  for kw in (dict(image_resolution_multiplier=8), dict(image_resolution_multiplier=4, conservative_rasterization=True),
             dict(image_resolution_multiplier=5, sub_grid_sampling=True, conservative_rasterization=True),
             dict(image_resolution_multiplier=3, projection_depth_multiplier=2)):
    g = voxelization.voxelize_mesh(t.tensor(tris), nt, (R, R, R), v2v[0], **kw).cpu().numpy()
    ref = O.voxelize_mesh(tris, nt, (R, R, R), v2v[0].numpy(), **kw)
    assert ref.sum() > 1000 and int((g != ref).sum()) == 0, (kw, int((g != ref).sum()))
  ref = O.voxelize_mesh(tris, nt, (R, R, R), v2v[0].numpy(), image_resolution_multiplier=8)
  labels = batched_example.voxelize_labels(t.tensor(tris), [t.tensor(nt, dtype=t.int32)], [[3, 5]], off, (R, R, R),
                                    image_resolution_multiplier=8).cpu().numpy()
  filled = O.fill_inside_voxels(ref)
  want = O.merge_labels(filled, [2], [[3, 5]])
  np.testing.assert_array_equal(labels, want)
  zz, yy, xx = np.meshgrid(*[np.arange(R) + 0.5] * 3, indexing="ij")
  inside2 = ((xx / R - 0.6) ** 2 + (yy / R - 0.5) ** 2 + (zz / R - 0.5) ** 2) < 0.17 ** 2
  assert (labels[0][inside2] == 5).all()


def test_voxelizer_full_size_bit_exact(be):
  """The BASELINE ground-truth workload (tools/bench_voxelize.py; h7.json5:54): 128^3, multiplier 8 (a 1024^2
  raster), UV spheres of 20 k triangles each, per-mesh view2voxel matrices with different sampling offsets:
  0 differing voxels against the oracle; then fill + label merge of the same scene, bit-exact against
  oracle rasterizer -> C fill oracle -> oracle merge."""
  from corenet_amd.data import batched_example
  from corenet_amd.geometry import voxelization
  import fill_oracle_c
  R = 128
  spheres = [((0.45, 0.5, 0.5), 0.25), ((0.6, 0.5, 0.5), 0.2), ((0.5, 0.42, 0.55), 0.3), ((0.3, 0.7, 0.4), 0.12)]
  tris = np.concatenate([_uv_sphere(100, 100, np.array(c), r) for c, r in spheres])
  nt = [20000] * 4
  off = t.tensor([[0.5, 0.5, 0.5], [0.25, 0.75, 0.5]])
  v2v = batched_example.view2voxel_matrices(off, (R, R, R))
  mesh_v2v = t.stack([v2v[0], v2v[0], v2v[1], v2v[1]])
  g = voxelization.voxelize_mesh(t.tensor(tris), nt, (R, R, R), mesh_v2v, image_resolution_multiplier=8).cpu().numpy()
  ref = O.voxelize_mesh(tris, nt, (R, R, R), mesh_v2v.numpy(), image_resolution_multiplier=8)
  assert ref.sum() > 50000 and int((g != ref).sum()) == 0, int((g != ref).sum())
  # sub-grid mode of generate_configs-style eval data (odd multiplier, conservative): (2R+1)^3 grids
  R2 = 64
  g = voxelization.voxelize_mesh(t.tensor(tris[:40000]), nt[:2], (R2, R2, R2), v2v[0] @ t.diag(t.tensor([.5, .5, .5, 1.])),
                                 sub_grid_sampling=True, image_resolution_multiplier=9,
                                 conservative_rasterization=True).cpu().numpy()
  ref2 = O.voxelize_mesh(tris[:40000], nt[:2], (R2, R2, R2), (v2v[0] @ t.diag(t.tensor([.5, .5, .5, 1.]))).numpy(),
                         sub_grid_sampling=True, image_resolution_multiplier=9, conservative_rasterization=True)
  assert g.shape == (2, 129, 129, 129) and ref2.sum() > 20000 and int((g != ref2).sum()) == 0
  labels = batched_example.voxelize_labels(t.tensor(tris), [t.tensor(nt[:2], dtype=t.int32), t.tensor(nt[2:], dtype=t.int32)],
                                           [[3, 5], [7, 2]], off, (R, R, R), image_resolution_multiplier=8).cpu().numpy()
  want = O.merge_labels(fill_oracle_c.fill(ref), [2, 2], [[3, 5], [7, 2]])
  np.testing.assert_array_equal(labels, want)
  assert set(np.unique(labels)) == {0, 2, 3, 5, 7}


def test_merge_labels_bit_exact(be):
  """crn_merge_labels alone (batched_example.py:186-196 + voxelization.get_sub_grid_centers :167-182): per scene
  max over meshes of label * occupancy -> int32; on overlap the larger class id wins (Q11); sub-grid mode reads
  the odd centres of the (2D+1)(2H+1)(2W+1) grids; a scene without meshes is all void."""
  g = t.Generator().manual_seed(11)
  for (D, H, W), sub in (((8, 6, 10), False), ((16, 16, 16), False), ((7, 5, 6), True), ((33, 20, 17), False)):
    num = [3, 1, 0, 4]
    M = sum(num)
    shape = (M, 2 * D + 1, 2 * H + 1, 2 * W + 1) if sub else (M, D, H, W)
    grids = (t.rand(shape, generator=g) < 0.4).float()
    labels = [[4, 9, 2], [1], [], [13, 13, 6, 1]]
    start = t.tensor(np.concatenate([[0], np.cumsum(num)]).astype(np.int32))
    flat = t.tensor([l for ls in labels for l in ls], dtype=t.int32)
    out = t.full((len(num), D, H, W), -7, dtype=t.int32, device=DEV)
    be.merge_labels(grids.to(DEV), start.to(DEV), flat.to(DEV), len(num), D, H, W, sub, out)
    centers = O.get_sub_grid_centers(grids.numpy()) if sub else grids.numpy()
    off = 0
    for b, n in enumerate(num):
      if n == 0:
        assert int(out[b].abs().sum()) == 0
      else:
        want = O.merge_labels(centers[off:off + n], [n], [labels[b]])[0]
        np.testing.assert_array_equal(out[b].cpu().numpy(), want)
      off += n
    assert EMU is be or t.equal(out.cpu(), _emu_merge(grids, start, flat, len(num), D, H, W, sub))


def _emu_merge(grids, start, flat, B, D, H, W, sub):
  out = t.zeros((B, D, H, W), dtype=t.int32)
  EMU.merge_labels(grids, start, flat, B, D, H, W, sub, out)
  return out


def test_data_path_batch_and_voxelize(be):
  """N2: fixture dataset -> reader -> `batch` (crn_transform_meshes) -> `voxelize` on the GPU.  View-space vertices
  against the reference's own collate output (tests/golden/data_path.npz; fp32, summation order of a 4-term dot
  product: <= 2e-6 relative), label grid against the oracle pipeline run on the golden vertices."""
  from corenet_amd.data import batched_example as B, dataset as D
  G_ = os.path.join(os.path.dirname(__file__), "golden")
  root = os.path.join(G_, "n2_dataset")
  z = np.load(os.path.join(G_, "data_path.npz"))
  ds = D.CoReNetDatasetImpl(os.path.join(root, "dataset.json"), os.path.join(root, "meshes"))
  ex = B.batch([ds[0], ds[1]])
  assert ex.vertices.is_cuda and ex.input_image.is_cuda and ex.vertices.shape == (90, 3, 3)
  got, want = ex.vertices.cpu().numpy(), z["vertices"]
  assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
  assert t.equal(ex.input_image.cpu(), t.as_tensor(z["hr_input_image"]))
  assert t.equal(ex.view_transform.cpu(), t.as_tensor(z["view_transform"]))
  # in place over the input is allowed by the ABI
  raw = t.as_tensor(z["raw_vertices"]).to(DEV)
  from corenet_amd.geometry import voxelization
  tm = voxelization.dynamic_tile(t.as_tensor(z["mesh_num_tri"])).to(DEV)
  mats = t.cat([t.matmul(e.view_transform[None], e.o2w_transforms) for e in (ds[0], ds[1])]).to(DEV)
  be.transform_meshes(raw, tm, mats, raw)
  assert t.equal(raw, ex.vertices)
  R = 32
  out = B.voxelize(ex, (R, R, R), B.VoxelContentSemanticLabel(ex.mesh_labels), image_resolution_multiplier=8)
  assert out.grid.shape == (2, R, R, R) and out.grid.dtype == t.int32 and out.grid.is_cuda
  assert t.equal(out.v2x_transform.cpu(), t.diag(t.tensor([32.0, 32, 32, 1])).expand(2, 4, 4))
  v2v = O.view2voxel_matrices(t.full((2, 3), 0.5), (R, R, R)).numpy()
  nm = list(z["num_meshes"])
  mesh_v2v = np.concatenate([np.repeat(v2v[b:b + 1], n, 0) for b, n in enumerate(nm)])
  # the oracle rasterizer is bit-defined (same fp32 operations as the kernel): fed the vertices the GPU produced,
  # the whole chain rasterizer -> fill -> merge must agree exactly
  ref = O.voxelize_mesh(got, list(z["mesh_num_tri"]), (R, R, R), mesh_v2v, image_resolution_multiplier=8)
  labels = np.split(z["mesh_labels"], np.cumsum(nm)[:-1])
  wantg = O.merge_labels(O.fill_inside_voxels(ref), nm, labels)
  assert wantg.max() >= 2
  np.testing.assert_array_equal(out.grid.cpu().numpy(), wantg)
  plain = B.voxelize(ex, (R, R, R), image_resolution_multiplier=8).grid.cpu().numpy()
  assert set(np.unique(plain)) <= {0, 1, 2, 3} and np.array_equal(plain > 0, wantg > 0)


def test_copy_tiles_pack_unpack(be):
  """crn_copy_tiles_f32 against the flat-index gather / scatter it replaces, on every kind of pack."""
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(3)
  geoms = [(G.conv_fwd((64, 32, 3, 3), 1), 0), (G.conv_dgrad((64, 32, 3, 3), 1), 9), (G.conv_dgrad((24, 515, 1, 1), 0), 0),
           (G.conv_fwd((16, 28, 5, 5, 5), 2), 0), (G.conv_dgrad((16, 28, 5, 5, 5), 2), 125),
           (G.convt_fwd((16, 14, 7, 7, 7), 3), 0), (G.convt_dgrad((16, 14, 7, 7, 7), 3), 64), (G.stem_fwd(), 0)]
  nparam = max(int(ge.index.max()) for ge, _ in geoms) + 1
  src = t.randn(nparam, generator=g)
  parts, off = [], 0
  for ge, grp in geoms:
    parts.append((off, ge.index, ge.npad, grp)); off += ge.index.size
  flat = t.as_tensor(np.concatenate([ge.index for ge, _ in geoms]).astype(np.int64))
  tiles = G.tile_index(parts)
  tiles_dev = (t.as_tensor(tiles[0]).to(DEV), t.as_tensor(tiles[1].view(np.int64)).to(DEV), t.as_tensor(tiles[2]).to(DEV))
  want = t.where(flat >= 0, src[flat.clamp(min=0)], t.zeros(()))
  got = t.zeros(off, device=DEV)
  be.copy_tiles(src.to(DEV), got, tiles_dev)
  assert t.equal(got.cpu(), want)
  # un-pack: every parameter of the forward-geometry packs comes back exactly once
  gi = t.as_tensor(geoms[3][0].index.astype(np.int64))
  tl = G.tile_index([(0, geoms[3][0].index, geoms[3][0].npad, 0)])
  tl_dev = (t.as_tensor(tl[0]).to(DEV), t.as_tensor(tl[1].view(np.int64)).to(DEV), t.as_tensor(tl[2] if tl[2].size else np.zeros(1, np.int32)).to(DEV))
  packed = t.randn(gi.numel(), generator=g)
  back = t.full((int(gi.max()) + 1,), -5.0, device=DEV)
  be.copy_tiles(packed.to(DEV), back, tl_dev, reverse=True)
  ref = t.full((int(gi.max()) + 1,), -5.0); ref[gi[gi >= 0]] = packed[gi >= 0]
  assert t.equal(back.cpu(), ref)


def test_copy_mats_pack_unpack(be):
  """crn_copy_mats_f32 (LDS-staged block copies of the plain convolutions' packs, conv_geometry.mat_index) + the 8x8
  tiles of what is left, against the flat-index gather: forward / data-gradient layouts of 1x1, 3x3, 3x3x3 and 5x5x5
  layers incl. ragged channel counts, plain biases; transposed convolutions, the stem and repeated biases must stay on
  the tiles.  Bit-exact both ways."""
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(5)
  layers = [((64, 32, 3, 3), 1), ((24, 515, 1, 1), 0), ((16, 28, 5, 5, 5), 2), ((128, 224, 5, 5, 5), 2), ((256, 256, 3, 3, 3), 1),
            ((2048, 512, 1, 1), 0), ((67, 130, 3, 3), 1)]
  for shape, p in layers:
    fw, dg = G.conv_fwd(shape, p), G.conv_dgrad(shape, p)
    n = int(np.prod(shape))
    src = t.randn(n, generator=g)
    parts = [(3, fw.index, fw.npad, 0), (3 + fw.index.size + 8, dg.index, dg.npad, dg.taps if dg.taps > 1 else 0)]
    total = parts[1][0] + dg.index.size
    mats, rest = G.mat_index(parts)
    assert mats.shape[0] and not rest, shape
    tl = G.tile_index(rest)
    dev = (t.as_tensor(tl[0]).to(DEV), t.as_tensor(tl[1].view(np.int64)).to(DEV), t.zeros(1, dtype=t.int32, device=DEV),
           t.as_tensor(mats).to(DEV))
    flat = t.full((total,), -1, dtype=t.int64)
    flat[3:3 + fw.index.size] = t.as_tensor(fw.index.astype(np.int64))
    flat[parts[1][0]:] = t.as_tensor(dg.index.astype(np.int64))
    want = t.where(flat >= 0, src[flat.clamp(min=0)], t.zeros(()))
    got = t.zeros(total, device=DEV)
    be.copy_tiles(src.to(DEV), got, dev)
    assert t.equal(got.cpu(), want), shape
    # un-pack of the forward layout (the gradient's): every parameter exactly once
    m1, r1 = G.mat_index(parts[:1])
    dev1 = (t.zeros((0, 6), dtype=t.int32, device=DEV), t.zeros(0, dtype=t.int64, device=DEV), t.zeros(1, dtype=t.int32, device=DEV),
            t.as_tensor(m1).to(DEV))
    packed = t.randn(3 + fw.index.size, generator=g)
    back = t.full((n,), -5.0, device=DEV)
    be.copy_tiles(packed.to(DEV), back, dev1, reverse=True)
    gi = t.as_tensor(fw.index.astype(np.int64))
    ref = t.full((n,), -5.0); ref[gi[gi >= 0]] = packed[3:][gi >= 0]
    assert t.equal(back.cpu(), ref), shape
  for part in ((0, G.convt_fwd((16, 14, 7, 7, 7), 3).index, G.convt_fwd((16, 14, 7, 7, 7), 3).npad, 0),
               (0, G.stem_fwd().index, G.stem_fwd().npad, 0), (0, G.bias_index(16, 8, 128, True).astype(np.int64), 128, 0)):
    assert G.mat_index([part])[0].shape[0] == 0
  assert G.mat_index([(0, G.bias_index(64, 1, 64).astype(np.int64), 64, 0)])[0].shape[0] == 1


BNF_CASES = [("conv3d_k5_64", "conv", (16, 28, 5, 5, 5), 2, (64, 64, 64), 1, True),      # stage_6.c1: 32-column blocks
             ("conv3d_k5_32", "conv", (32, 56, 5, 5, 5), 2, (32, 32, 32), 4, True),      # stage_5.c1 at the bench batch
             ("convT_k7_32_c16", "convT", (32, 16, 7, 7, 7), 3, (32, 32, 32), 4, True),  # stage_5.t1: stride-2 input view
             ("convT_k7_64_c2", "convT", (16, 2, 7, 7, 7), 3, (64, 64, 64), 1, True),    # stage_6.t1
             ("conv3d_k5_16_c112", "conv", (64, 112, 5, 5, 5), 2, (16, 16, 16), 2, False)]  # splits its reduction: no fusion


@pytest.mark.parametrize("name,kind,wshape,pad,dims,B,fuses", BNF_CASES, ids=[c[0] for c in BNF_CASES])
def test_conv_dgrad_fused_bn_bwd_sums(be, name, kind, wshape, pad, dims, B, fuses):
  """A decoder data gradient that also leaves the two sums of the BatchRenorm backward behind it
  (crn_conv_fwd_bf3_slabs_bnbwd + crn_batch_renorm_bwd_apply; reconstruction_decoder.py:56-60 is ReLU -> norm -> conv,
  batch_renorm.py:41-47) against the two separate calls (crn_conv_fwd_bf3_slabs, crn_batch_renorm_bwd): the data gradient
  bit for bit, dx / dgamma / dbeta / the conv bias gradient within 2e-5 (only the order of the two sums differs), and dx
  against the contract emulator.  A launch that splits its reduction reports that it did not fuse and the caller's
  ordinary norm backward gives the same result."""
  if _SELF:
    return
  from corenet_amd import views as V
  from corenet_amd.model import conv_geometry as G
  g = t.Generator().manual_seed(hash(name) % 1000 + 5)
  w = t.randn(wshape, generator=g) / np.sqrt(np.prod(wshape[1:]))
  if kind == "conv":
    cin, cout = wshape[1], wshape[0]
    fwd, dgr = G.conv_fwd(wshape, pad), G.conv_dgrad(wshape, pad)
    odims = dims
  else:
    cin, cout = wshape[0], wshape[1]
    fwd, dgr = G.convt_fwd(wshape, pad), G.convt_dgrad(wshape, pad)
    odims = tuple(2 * v for v in dims)
  S = int(np.prod(dims))
  wd = EMU_pack(w, dgr)
  nsd = G.slab_entries(dgr)
  desc, blocks = G.operand_table([(0, 0, dgr, True)])
  slab = t.zeros(nsd * 32, dtype=t.uint8, device=DEV)
  be.bf3_operands(wd.to(DEV), (t.as_tensor(desc).to(DEV), blocks), slab)
  x = t.randn((B, cin) + dims, generator=g)                       # input of the norm (before its ReLU)
  dy = t.randn((B, cout) + odims, generator=g)                    # gradient of the conv's output
  gamma = t.rand(cin, generator=g) + 0.5
  mu = t.rand(cin, generator=g) * 0.4; rstd = 1.0 / (t.rand(cin, generator=g) * 0.5 + 0.3)
  r = t.rand(cin, generator=g) * 0.5 + 0.75; dd = t.randn(cin, generator=g) * 0.1
  saved = t.cat([mu, rstd, r, dd]).contiguous()
  scale = gamma * r * rstd; shift = t.randn(cin, generator=g) * 0.1

  def dyview(tens):
    v = V.view_of(tens)
    return V.space_to_depth_view(v, (2, 2, 2), parity_major=True) if kind == "convT" else v

  xg, dyg = x.to(DEV), dy.to(DEV)
  gm, sc, sh, sv = gamma.to(DEV), scale.to(DEV), shift.to(DEV), saved.to(DEV)
  ndsum = max(1, cin - 3)                                          # (a concat buffer: the producing conv has fewer channels)
  out = {}
  for mode in ("separate", "fused"):
    gq = t.full((B, cin) + dims, 3.0, device=DEV)
    dx = t.full((B, cin) + dims, 5.0, device=DEV)
    dg, db, ds = t.full((cin,), 7.0, device=DEV), t.full((cin,), 7.0, device=DEV), t.full((cin,), 7.0, device=DEV)
    if mode == "separate":
      be.conv_fwd(dyview(dyg), None, None, dgr.npad, None, 0, V.view_of(gq), dgr.window, dgr.pad_lo, 0, False,
                  boxes=(dgr.n_boxes, dgr.c_boxes), math="bf16x3", wslab=slab)
      did = False
    else:
      be.splitk_defer()
      did = be.conv_dgrad_bn_bwd(dyview(dyg), slab, dgr.npad, V.view_of(gq), gq, dgr.window, dgr.pad_lo,
                                 (dgr.n_boxes, dgr.c_boxes), xg, cin * S, B, cin, S, True, gm, sc, sh, sv, dx, cin * S, dg, db,
                                 dsum=ds, ndsum=ndsum)
      assert did == fuses, (name, did)
    if not did:
      be.bn_bwd(xg, cin * S, gq, cin * S, B, cin, S, True, False, gm, sc, sh, sv, dx, cin * S, dg, db, dsum=ds, ndsum=ndsum)
    out[mode] = [v.cpu() for v in (gq, dx, dg, db, ds)]
  a, b = out["separate"], out["fused"]
  if fuses:
    assert t.equal(a[0], b[0]), (name, "data gradient")
  # (a launch that splits, armed by crn_splitk_defer: the sum is left to the norm's backward, its only reader, and the
  # data gradient itself is never materialised)
  for u, v, nm in zip(a[1:], b[1:], ("dx", "dgamma", "dbeta", "dsum")):
    if nm == "dsum":
      u, v = u[:ndsum], v[:ndsum]
      assert float(a[4][ndsum:].min()) == 7.0 and float(b[4][ndsum:].max()) == 7.0
    e = float((u - v).abs().max() / u.abs().max())
    print(f"{name} fused vs separate {nm}: {e:.2e}")
    assert e <= 2e-5, (name, nm, e)
  # ... and against the contract (emulator) on the fused launch's own data gradient
  dxe, dge, dbe = t.zeros_like(x), t.zeros(cin), t.zeros(cin)
  EMU.bn_bwd(x, cin * S, a[0], cin * S, B, cin, S, True, False, gamma, scale, shift, saved, dxe, cin * S, dge, dbe)
  e = float((b[1] - dxe).abs().max() / dxe.abs().max())
  print(f"{name} fused dx vs contract: {e:.2e}")
  assert e <= 2e-5, (name, "dx vs contract", e)
  assert float((b[2] - dge).abs().max() / dge.abs().max()) <= 2e-5 and float((b[3] - dbe).abs().max() / dbe.abs().max()) <= 2e-5
