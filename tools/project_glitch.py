#!/usr/bin/env python
"""crn_ray_project (the loop-free projection kernel) on a side stream beside the MFMA probe (tools/mfma_probe.hip) on the main
stream: how many of its indices differ from the CPU contract, where (lane inside the wave), per probe mode and start delay.
usage: project_glitch.py [runs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch as t
from oracle import corenet_oracle as O
from kernel_contract_emu import EmuBackend
from corenet_amd import _lib
from corenet_amd.backend import HipBackend

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
be = HipBackend()
probe = ctypes.CDLL(_lib.PROBE_LIB_PATH)
res, B = 64, 2
base = O.canonical_camera() @ O.scale([1.0 / 128] * 3) @ O.scale([128.0 / res] * 3)
shift = O.translate([0.9, -0.4, -0.95]) @ base
cams = t.stack([base, shift]); offs = t.tensor([[0.5, 0.5, 0.5], [0.25, 0.5, 0.75]])
want = EmuBackend.ray_indices_u16(cams.reshape(B, 16), offs, B, res, res, res, res, res)
md, od = cams.reshape(B, 16).cuda(), offs.cuda()
idx = t.zeros(B, res ** 3, dtype=t.int16, device="cuda")
sink = t.zeros(16, device="cuda")
side = t.cuda.Stream()
for mode in (None, 0, 32, 1, 44, 48):
  for delay in (0, 20000):
    nbad, tot, lanes, samples = 0, 0, np.zeros(64, np.int64), set()
    for i in range(runs):
      idx.zero_(); t.cuda.synchronize()
      ev = t.cuda.Event(); ev.record()
      if mode is not None:
        assert probe.crn_mfma_probe(mode, 6000, 256, ctypes.c_void_p(sink.data_ptr()), ctypes.c_void_p(_lib.stream())) == 0
      with t.cuda.stream(side), _lib.pinned_stream(side):
        side.wait_event(ev)
        if delay: t.cuda._sleep(delay)
        be.ray_project(md, od, B, res, res, res, res, res, idx)
      t.cuda.synchronize()
      got = idx.cpu().to(t.int64).view(B, res, res, res) & 0xFFFF
      bad = (got != want)
      n = int(bad.sum())
      if n:
        nbad += 1; tot += n
        flat = bad.view(B, -1).nonzero()
        # thread t of a sample handles voxels 4 t .. 4 t + 3: lane = (voxel // 4) % 64
        for bb, v in flat[:20000].tolist():
          lanes[(v // 4) % 64] += 1
          samples.add(bb)
    q = [int(lanes[16 * k:16 * k + 16].sum()) for k in range(4)]
    print(f"neighbour probe mode {mode} delay {delay}: {nbad} of {runs} runs off, {tot} wrong indices, by quarter-wave {q}, samples {sorted(samples)}")
