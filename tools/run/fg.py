import os, sys
sys.path.insert(0, "/root/repo")
import torch as t
import bench
from corenet_amd import _lib
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
from oracle import corenet_oracle as O
det = os.environ.get("DET", "0") == "1"
if det: _lib.lib().cdll.crn_set_deterministic(1)
sd = O.make_state(0, 2, nbt=0)
image, v2s, off, grid = [x.cuda() for x in bench.synthetic_batch(4, 0, 2)]
grid = grid.to(t.int32)
res = {}
for mode in ("0", "fwd"):
  os.environ["CRN_GRAPH"] = mode
  m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math="bf16x3")
  m.load_state_dict(sd); m.train()
  losses = []
  for i in range(6):
    losses.append(float(m.train_step(image, v2s, off, grid, "iou_fgbg")))
  t.cuda.synchronize()
  res[mode] = (losses, m.engine.store.params.clone())
  a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
  for rep in range(2):
    a.record()
    for _ in range(30): m.train_step(image, v2s, off, grid, "iou_fgbg")
    b.record(); t.cuda.synchronize()
    print(f"CRN_GRAPH={mode}: {a.elapsed_time(b) / 30:.3f} ms/step")
  import time
  t0 = time.perf_counter()
  for _ in range(30): m.train_step(image, v2s, off, grid, "iou_fgbg")
  t1 = time.perf_counter(); t.cuda.synchronize()
  print(f"CRN_GRAPH={mode}: host enqueue {(t1 - t0) / 30 * 1e3:.2f} ms/step")
  del m
print("losses eager", res["0"][0]); print("losses fwd  ", res["fwd"][0])
d = (res["0"][1] - res["fwd"][1]).abs().max().item()
print("max |param diff| after 6 steps:", d, "(deterministic mode)" if det else "")
