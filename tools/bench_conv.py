#!/usr/bin/env python
"""Micro-benchmark of one conv layer through the C ABI (for rocprofv3 --pmc runs).
usage: bench_conv.py <fwd|dgrad|wgrad> <layer-key> [iters] [B] [fp32|bf16x3]
The stamp switches (CRN_BF3_STAMPS, CRN_PW_STAMPS) need the tools build of the library (tools/_build/libcorenet_hip_tools.so,
python -m corenet_amd.build --tools), which this script selects when one of them is set."""
import os, sys, time
if os.environ.get("CRN_BF3_STAMPS") or os.environ.get("CRN_PW_STAMPS"):
  os.environ.setdefault("CRN_TOOLS_LIB", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from corenet_amd import views as V
from corenet_amd.backend import HipBackend, Transform
from corenet_amd.model import conv_geometry as G

LAYERS = {  # name: (kind, wshape, pad, in dims)
    "s6c1": ("conv", (16, 28, 5, 5, 5), 2, (64, 64, 64)),
    "s5c1": ("conv", (32, 56, 5, 5, 5), 2, (32, 32, 32)),
    "s6t1": ("convT", (16, 2, 7, 7, 7), 3, (64, 64, 64)),
    "s5t1": ("convT", (32, 16, 7, 7, 7), 3, (32, 32, 32)),
    "s6t1c14": ("convT", (16, 14, 7, 7, 7), 3, (64, 64, 64)),     # m7/m9: 14 classes
    "e2c": ("conv", (256, 64, 1, 1), 0, (1, 64, 64)),
    "e2a": ("conv", (64, 256, 1, 1), 0, (1, 64, 64)), "e2a0": ("conv", (64, 64, 1, 1), 0, (1, 64, 64)),
    "e3c": ("conv", (512, 128, 1, 1), 0, (1, 32, 32)), "e3a": ("conv", (128, 512, 1, 1), 0, (1, 32, 32)),
    "e4c": ("conv", (1024, 256, 1, 1), 0, (1, 16, 16)), "e4a": ("conv", (256, 1024, 1, 1), 0, (1, 16, 16)),
    "e5c": ("conv", (2048, 512, 1, 1), 0, (1, 8, 8)), "e5a": ("conv", (512, 2048, 1, 1), 0, (1, 8, 8)),
    "e3b": ("conv", (128, 128, 3, 3), 1, (1, 32, 32)),
    "e2b": ("conv", (64, 64, 3, 3), 1, (1, 64, 64)),
    "e4b": ("conv", (256, 256, 3, 3), 1, (1, 16, 16)),
    "e5b": ("conv", (512, 512, 3, 3), 1, (1, 8, 8)),
    "s4c1": ("conv", (64, 112, 5, 5, 5), 2, (16, 16, 16)),
    "s4t1": ("convT", (64, 32, 7, 7, 7), 3, (16, 16, 16)),
    "s3c1": ("conv", (128, 224, 5, 5, 5), 2, (8, 8, 8)),
    "s3t1": ("convT", (128, 64, 7, 7, 7), 3, (8, 8, 8)),
    "s2c1": ("conv", (256, 259, 3, 3, 3), 1, (4, 4, 4)),
    "s2t1": ("convT", (256, 128, 3, 3, 3), 1, (4, 4, 4)),
}
mode, key = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
MATH = sys.argv[5] if len(sys.argv) > 5 else "fp32"
kind, wshape, pad, dims = LAYERS[key]
be = HipBackend()
g = t.Generator().manual_seed(0)
is2d = len(wshape) == 4
if kind == "conv":
  cin, cout = wshape[1], wshape[0]; fwd, dgr = G.conv_fwd(wshape, pad), G.conv_dgrad(wshape, pad); odims = dims
else:
  cin, cout = wshape[0], wshape[1]; fwd, dgr = G.convt_fwd(wshape, pad), G.convt_dgrad(wshape, pad)
  odims = tuple(2 * d for d in dims)
sh = lambda c, d: (B, c) + (d[1:] if is2d else d)
x = t.randn(sh(cin, dims), generator=g).cuda(); y = t.zeros(sh(cout, odims)).cuda()
w = t.randn(wshape, generator=g) * 0.05
pk = lambda idx: t.where(t.as_tensor(idx) >= 0, w.reshape(-1)[t.as_tensor(idx).clamp(min=0).long()], t.zeros(())).cuda()
wf, wd = pk(fwd.index), pk(dgr.index)
sc, shf = (t.rand(cin) + 0.5).cuda(), t.randn(cin).cuda()
tr = Transform(sc, shf, pre_relu=True)
yv = V.space_to_depth_view(V.view_of(y), (2, 2, 2), parity_major=True) if kind == "convT" else V.view_of(y)
dw = t.zeros(wf.numel()).cuda()
sf = sd = None
if MATH == "bf16x3" and os.environ.get("CRN_BF3_SLABS", "1") != "0":      # weights pre-arranged as slab images
  nsf, nsd = G.slab_entries(fwd), G.slab_entries(dgr)
  desc, blocks = G.operand_table([(0, 0, fwd, True), (wf.numel(), nsf, dgr, True)])
  slabs = t.zeros((nsf + nsd) * 32, dtype=t.uint8, device="cuda")
  be.bf3_operands(t.cat([wf, wd]), (t.as_tensor(desc).cuda(), blocks), slabs)
  sf, sd = slabs[:nsf * 32], slabs[nsf * 32:]
def run():
  if mode == "fwd": be.conv_fwd(V.view_of(x), tr, wf, fwd.npad, None, 0, yv, fwd.window, fwd.pad_lo, 0, boxes=(fwd.n_boxes, fwd.c_boxes), math=MATH, wslab=sf)
  elif mode == "dgrad": be.conv_fwd(yv, None, wd, dgr.npad, None, 0, V.view_of(x), dgr.window, dgr.pad_lo, 0, boxes=(dgr.n_boxes, dgr.c_boxes), math=MATH, wslab=sd)
  else: be.conv_wgrad(V.view_of(x), tr, yv, dw, fwd.npad, fwd.window, fwd.pad_lo, True, boxes=(fwd.n_boxes, fwd.c_boxes), math=MATH)
def timeit(fn):
  for _ in range(3): fn()
  t.cuda.synchronize(); a = t.cuda.Event(enable_timing=True); b = t.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(iters): fn()
  b.record(); t.cuda.synchronize()
  return a.elapsed_time(b) / iters
for _ in range(3): run()
t.cuda.synchronize(); a = t.cuda.Event(enable_timing=True); b = t.cuda.Event(enable_timing=True)
a.record()
for _ in range(iters): run()
b.record(); t.cuda.synchronize()
ms = a.elapsed_time(b) / iters
import numpy as np
flop = 2.0 * B * np.prod(dims) * cin * cout * np.prod(wshape[2:])
print(f"{mode} {key} B={B} {MATH}: {ms*1e3:.1f} us  {flop/ms/1e9:.1f} TFLOP/s (real flops)")

if os.environ.get("CRN_BF3_STAMPS") == "ws":  # wave-specialised kernel: consumer wave 0 / producer wave 8 of workgroup 0
  import ctypes
  st = (ctypes.c_longlong * 192)()
  be.lib.cdll.crn_bf3_debug_stamps(st)
  base = min(v for v in st if v > 0)
  print("iter | consumer: barrier-exit  +dma-issue  +mfma  +vmcnt0 | producer: barrier-exit  +issue  +wait  +commit   (cycles; exits relative to the first stamp)")
  for i in range(24):
    r = st[i * 8:i * 8 + 8]
    if r[0] == 0 and r[4] == 0: break
    c = f"{r[0]-base:8d} {r[1]-r[0]:6d} {r[2]-r[1]:6d} {r[3]-r[2]:6d}" if r[0] else " " * 29
    q = f"{r[4]-base:8d} {r[5]-r[4]:6d} {r[6]-r[5]:6d} {r[7]-r[6]:6d}" if r[4] else ""
    print(f"{i:4d} | {c} | {q}")
elif os.environ.get("CRN_BF3_STAMPS"):        # per-step phases of workgroup 0 (conv_bf3.hip, crn_bf3_debug_stamps)
  import ctypes
  st = (ctypes.c_longlong * 192)()
  be.lib.cdll.crn_bf3_debug_stamps(st)
  names = ["wait", "barrier", "commit", "barrier", "issue", "mfma"]
  tot = [0] * 6
  n = 0
  for i in range(24):
    r = st[i * 8:i * 8 + 7]
    if r[6] == 0: break
    d = [r[j + 1] - r[j] for j in range(6)]
    print(f"step {i:2d}: " + "  ".join(f"{nm} {v:6d}" for nm, v in zip(names, d)) + (f"   gap {r[0] - prev:6d}" if i else ""))
    prev = r[6]; n += 1
    for j in range(6): tot[j] += d[j]
  if n: print("mean   : " + "  ".join(f"{nm} {v // n:6d}" for nm, v in zip(names, tot)))
if os.environ.get("CRN_PW_STAMPS"):            # pointwise kernel (1x1 layers, fp32): workgroup (0,0,0), thread 0
  import ctypes
  st = (ctypes.c_longlong * 32)()
  if be.lib.cdll.crn_pw_debug_stamps(st) == 0:
    marks = [v for v in st[:30] if v > 0]
    if st[30] > 0: print("  shader clock during workgroup 0: %.2f GHz (%d cycles in %d ticks of 10 ns)" % (st[31] / st[30] / 10.0, st[31], st[30]))
    print("pointwise stamps (shader cycles from kernel entry): setup %d, loads issued %d; per chunk (wait, commit, mfma): %s; end %d"
          % (st[0], st[1], " ".join("(%d %d %d)" % (marks[i] - marks[i - 1], marks[i + 1] - marks[i], marks[i + 2] - marks[i + 1])
                                    for i in range(2, len(marks) - 2, 3)), st[31]))
