import os, sys, numpy as np, torch as t
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import corenet_oracle as O
from corenet_amd.model import losses
from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
z = np.load("/root/repo/tests/golden/model_h7_train_b2_nbt30k.npz")
m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda", decoder_math=sys.argv[1])
m.load_state_dict(O.make_state(0, 2, nbt=30000)); m.train()
image, v2s, off, grid = O.synthetic_batch(2, 0, 2)
losses.iou_fgbg(grid.cuda(), m(image.cuda(), v2s.cuda(), off.cuda())).backward()
gmax = max(float(z[k]) for k in z.files if k.startswith("gmax::"))
print("gmax", gmax)
for name in ["encoder.stage3.c.op_b.bn.bias", "encoder.stage4.e.op_b.bn.bias", "encoder.stage3.c.op_b.bn.weight", "encoder.stage3.c.op_b.conv.bias"]:
  g = m.get_parameter(name).grad.reshape(-1)
  st = max(1, -(-g.numel() // 512))
  got = g[::st].double().cpu().numpy(); w64 = z["g64sub::" + name].astype(np.float64); r32 = z["gsub::" + name].astype(np.float64)
  i = int(np.abs(got - w64).argmax())
  print(name, "tensor max", float(z["gmax::" + name]), "worst idx", i, "got", got[i], "fp64", w64[i], "ref32", r32[i], "| sorted |err|:", np.sort(np.abs(got - w64))[-4:], "ref32 err", np.sort(np.abs(r32 - w64))[-2:])
