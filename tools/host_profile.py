import os, sys, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch as t, bench
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math="bf16x3")
m.train()
image, v2s, off, grid = [x.cuda() for x in bench.synthetic_batch(4, 0, 2)]
grid = grid.to(t.int32)
for _ in range(3): m.train_step(image, v2s, off, grid, "iou_fgbg")
t.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): m.train_step(image, v2s, off, grid, "iou_fgbg")
pr.disable()
t.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
