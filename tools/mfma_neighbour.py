#!/usr/bin/env python
"""Two kernels alone: a neighbour on the main stream and the ray-sample scatter (64^3, 12 channels) on a side stream, started
together; the scatter's result is compared with its serial run.  Round 4 found with this pair what made the scatter's sums differ
from run to run inside the training step: MFMA-dense neighbours on the same CU whose MFMAs wait for each
other (accumulator chains) make VALU results of OTHER waves go missing (lanes 48-63 keep the old register content).
  neighbour = crn_mfma_probe, a kernel of nothing but v_mfma_f32_16x16x32_bf16 in a fixed order (30 runs each, MI355X):
    mode 0   four independent accumulators                                              0 of 30 scatters wrong
    mode 1   one accumulator, four MFMAs per loop trip, 8 idle cycles after each       29 of 30
    mode 33  the same without the idle cycles                                           0 of 30
    modes 2, 4 (interleaved chains), 3, 10, 12-15, 20-28, 30, 31 (two-MFMA chains with 0 ... 48 idle cycles, shared A or B
    registers), 32 (the split-bf16 triple back to back)                                 0 of 30 each
    modes 40-43 (mode 1 with 1, 2, 4, 6 idle cycles)                                    0 of 30 each
    modes 44-47 (mode 1 with 12, 16, 24, 32 idle cycles), 48 (three MFMAs per trip, 8)  28-29 of 30 each
  -- three or more dependent MFMAs in a row with >= 8 idle cycles between them do it; back to back, or two per loop trip, do not.
  neighbour = a convolution of the library (tools/bench_conv.py layer keys): with the three products of an accumulator left to the
  compiler's scheduler (two adjacent MFMAs and a third behind LDS reads) `fwd s6c1`, `fwd s6t1`, `dgrad s6t1`, `fwd s5t1`, `dgrad s5t1`
  (bf16x3) made 29 / 20 / 5 / 20 / 28 of 30 scatters wrong; as one block of three adjacent MFMAs (mfma3, csrc/conv_bf3.hip, the
  library now) every layer and direction: 0 of 60.
usage: mfma_neighbour.py probe <mode>
       mfma_neighbour.py <fwd|dgrad|wgrad> <layer key of tools/bench_conv.py> [fp32|bf16x3]"""
import os, sys, runpy, io, contextlib, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode, key = sys.argv[1], sys.argv[2]
math = sys.argv[3] if len(sys.argv) > 3 else "bf16x3"
if mode == "probe":
  import torch as t0
  from corenet_amd.backend import HipBackend
  from corenet_amd import _lib as _l0
  be = HipBackend()
  probe = ctypes.CDLL(_l0.PROBE_LIB_PATH)          # tools/mfma_probe.hip (python -m corenet_amd.build --tools)
  sink = t0.zeros(16, device="cuda")
  iters = int(os.environ.get("PROBE_ITERS", "3000"))
  def run():
    rc = probe.crn_mfma_probe(int(key), iters, 256, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(_l0.stream()))
    assert rc == 0, rc
else:
  sys.argv = ["bench_conv.py", mode, key, "1", "2", math]
  with contextlib.redirect_stdout(io.StringIO()):
    G = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_conv.py"), run_name="aggressor")
  run, be = G["run"], G["be"]
import torch as t
from oracle import corenet_oracle as O
from corenet_amd import _lib
B, C, res = 2, 28, 64
g = t.Generator().manual_seed(0)
gu = (t.randn(B, C, res, res, res, generator=g) * 1e-6).cuda()
m = (O.canonical_camera() @ O.scale([1.0 / 128] * 3) @ O.scale([128.0 / res] * 3))[None].expand(B, 4, 4).reshape(B, 16).contiguous().cuda()
off = t.full((B, 3), 0.5).cuda()
gmap = t.zeros(B, 12, res, res).cuda()
def scatter():
  be.ray_sample_bwd(gu[:, 16:], gu.stride(0), B, 12, res, res, res, m, off, gmap, gmap.stride(0), res, res, True)
scatter(); t.cuda.synchronize(); ref = gmap.clone()
side = t.cuda.Stream()
for delay in (0, 20000, 60000):
  nbad, worst = 0, 0.0
  for i in range(30):
    t.cuda.synchronize()
    ev = t.cuda.Event(); ev.record()
    run()
    with t.cuda.stream(side), _lib.pinned_stream(side):
      side.wait_event(ev)
      if delay: t.cuda._sleep(delay)
      scatter()
    t.cuda.synchronize()
    e = float((gmap - ref).abs().max() / ref.abs().max())
    worst = max(worst, e); nbad += e > 1e-5
  print(f"neighbour {mode} {key}{'' if mode == 'probe' else ' ' + math}, scatter delayed by _sleep({delay}): {nbad} of 30 scatters off by more than 1e-5 (worst {worst:.1e})")
