import os, sys, ctypes
sys.path.insert(0, "/root/repo")
import numpy as np
import torch as t
from oracle import corenet_oracle as O
from corenet_amd import _lib
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
B = 2
sd = O.make_state(0, 2, nbt=0)
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(B, 0, 2)]
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math="bf16x3")
m.load_state_dict(sd); m.train()
cd = _lib.lib().cdll
buf = (ctypes.c_ulonglong * (8 * 4096))()
cd.crn_ray_dbg_wg(buf, 1)
ref = None
log = (ctypes.c_ulonglong * (512 * 8))(); ln = ctypes.c_uint(0)
for dl in ():
  e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
  t.cuda.synchronize(); e0.record(); t.cuda._sleep(max(dl, 1)); e1.record(); t.cuda.synchronize()
  os.environ["CRN_DBG_DELAY"] = str(dl)
  os.environ["CRN_DBG_SKIP"] = "side"
  tot = 0; cans = [0] * 16
  for rep in range(8):
    m.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", lr=0.0, adam_eps=1e-4)
    t.cuda.synchronize()
    cd.crn_ray_dbg_wg(buf, 1)
    cd.crn_ray_dbg_log(log, ctypes.byref(ln)); tot += ln.value
    can = (ctypes.c_uint * 16)(); cd.crn_ray_canary_read(can); cans = [a + b for a, b in zip(cans, list(can))]
  print(f"scatter delayed by _sleep({dl}) = {e0.elapsed_time(e1) * 1e3:.1f} us (incl. launch): {tot} glitched threads in 8 steps; canaries at kernel start {cans[:4]}, right after the flush [mul+add, lcg, mul+add, u64 mad] {cans[8:12]}, after the third pass {cans[4:8]}")
os.environ["CRN_DBG_DELAY"] = "0"
for rep in range(8):
  m.engine.plan(B).ray_side = rep >= 2                      # (the first two steps: scatter on the main stream = the reference)
  m.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", lr=0.0, adam_eps=1e-4)
  t.cuda.synchronize()
  rc = cd.crn_ray_dbg_wg(buf, 1)
  a = np.frombuffer(buf, dtype=np.uint64).reshape(8, 4096).copy()
  if ref is None: ref = a
  names = ["pixel", "pixel(2nd)", "clip", "clip(2nd)", "quotient", "quotient(2nd)", "screen", "screen(2nd)"]
  line = ", ".join(f"{names[i]} {np.nonzero(a[i] != ref[i])[0].tolist()}" for i in range(8))
  print(f"rep {rep} ({'side' if rep >= 2 else 'main'}): rc {rc}; workgroups that differ from the reference: {line}")
  rc = cd.crn_ray_dbg_log(log, ctypes.byref(ln))
  L = np.frombuffer(log, dtype=np.uint64).reshape(512, 8)
  print(f"    log: {ln.value} threads (rc {rc})")
  for e in L[:min(ln.value, 40)]:
    wg, tid = int(e[0]) >> 32, int(e[0]) & 0xffffffff
    hwid, xcc = int(e[1]) & 0xffffffff, int(e[1]) >> 32
    import struct
    f = lambda u: struct.unpack("f", struct.pack("I", u & 0xffffffff))[0]
    print(f"      wg {wg} tid {tid} (wave {tid // 64} lane {tid % 64}): first!=second {(int(e[2]) >> 1) & 1}, second!=third {int(e[2]) & 1}; plane {(int(e[5]) >> 32) - 100}: py second {f(int(e[7]) >> 32):.6f} third {f(int(e[7])):.6f}; probe before the loops: m5*cy second {f(int(e[3]) >> 32):.6f} third {f(int(e[3])):.6f}, m5 second {f(int(e[4]) >> 32):.6f} third {f(int(e[4])):.6f}, cy {f(int(e[6]) >> 32)}")
  can = (ctypes.c_uint * 16)()
  rc = cd.crn_ray_canary_read(can)
  print(f"    in-kernel canaries [u32 lcg, u64 mad, divergent branches, mul+add]: threads whose two evaluations differ at the start of the scatter kernel {list(can)[:4]}, after its third pass {list(can)[4:8]}")
