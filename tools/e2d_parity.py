import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch as t
from oracle import corenet_oracle as O
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
G = "tests/golden"
def relerr(a, b):
  b = t.as_tensor(b).to(a.device)
  return float((a - b).abs().max() / b.abs().max())
for tag, nc, nbt, B in (("h7_train_b1", 2, 0, 1), ("h7_train_b2_nbt30k", 2, 30000, 2), ("m9_train_b1", 14, 0, 1)):
  z = np.load(os.path.join(G, f"model_{tag}.npz"))
  m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), nc, 2, 64, 0.75)), device="cuda", decoder_math="bf16x3")
  m.load_state_dict(O.make_state(0, nc, nbt=nbt)); m.train()
  image, v2s, off, grid = O.synthetic_batch(B, 0, nc)
  with t.no_grad():
    logits = m(image.cuda(), v2s.cuda(), off.cuda())
  print(os.environ.get("CRN_E2D_KINDS"), os.environ.get("CRN_E2D_FWD_STAGES"), tag, f"{relerr(logits[:, :, ::16, ::16, ::16], z['logits_sub']):.2e}", flush=True)
