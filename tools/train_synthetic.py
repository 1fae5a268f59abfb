#!/usr/bin/env python
"""Does the whole path learn?  Trains the h7 configuration (random init) on ONE fixed synthetic batch (random
images, analytic ball as ground truth; SURVEY 8d inputs) with the fused train step and prints the loss and the
foreground IoU of the train-mode logits every 25 steps.  Not a reproduction of the paper's numbers (no dataset
here) -- a functional check of forward + loss + backward + Adam as one system.
usage: python tools/train_synthetic.py [steps] [classes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
from oracle import corenet_oracle as O            # synthetic inputs / deterministic weights only
from corenet_amd import voxel_metrics as VM
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
C = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 4
loss_name = "iou_fgbg" if C == 2 else "xent_times_iou_agnostic"
model = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), C, 2, 64, 0.75)), device="cuda")
model.load_state_dict(O.make_state(0, C, nbt=0)); model.train()
image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(B, seed=0, num_classes=C)]
grid = grid.to(t.int32)
plan = model.engine.plan(B)
t0 = time.time()
for s in range(steps + 1):
  loss = model.train_step(image, v2s, off, grid, loss_name, lr=4e-4, adam_eps=1e-4)
  if s % 25 == 0:
    _, cm = VM.argmax_confusion(plan.logits, grid, C)          # logits of this step's (train-mode) forward
    print(f"step {s:4d}  loss {float(loss):.4f}  mean IoU (non-void classes) {VM.mean_iou(cm):.4f}", flush=True)
t.cuda.synchronize()
print(f"{steps + 1} steps in {time.time() - t0:.1f} s")
