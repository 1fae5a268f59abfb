import os, sys
sys.path.insert(0, os.getcwd())
import torch as t, bench
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math=os.environ.get("M", "bf16x3"))
image, v2s, off, grid = [x.cuda() for x in bench.synthetic_batch(4, 0, 2)]
m.eval()
with t.no_grad():
  for _ in range(3): m(image, v2s, off)
  t.cuda.synchronize()
  a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(20): m(image, v2s, off)
  b.record(); t.cuda.synchronize()
print(os.environ.get("M", "bf16x3"), os.environ.get("CRN_E2D"), os.environ.get("CRN_BF3_SLABS"), f"eval forward {a.elapsed_time(b)/20:.3f} ms/batch")
