"""Cost of the bucketed backward + overlapped RCCL exchange on ONE GPU (world_size 1, where the all-reduce
moves no data): step time of train_step with no exchange, with the one-slab exchange after backward, and
with the overlapped buckets.   python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import torch.distributed as dist
from corenet_amd import distributed as D
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
from oracle import corenet_oracle as O

dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29711", rank=0, world_size=1)
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda")
m.load_state_dict(O.make_state(0, 2, nbt=0)); m.train()
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(4, 0, 2)]
grid = grid.to(t.int32)
for name, sync in (("no exchange", None), ("one slab after backward", D.GradientSync(1, overlap=False, force=True)),
                   ("overlapped buckets", D.GradientSync(1, overlap=True, force=True))):
  for _ in range(3):
    m.train_step(image, v2s, off, grid, "iou_fgbg", all_reduce=sync)
  t.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(20):
    m.train_step(image, v2s, off, grid, "iou_fgbg", all_reduce=sync)
  t.cuda.synchronize()
  print(f"{name:28s} {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step")
dist.destroy_process_group()
