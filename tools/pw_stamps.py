import os, sys, ctypes
os.environ.setdefault("CRN_TOOLS_LIB", "1")          # crn_pw_debug_stamps exists only in the tools build of the library
sys.path.insert(0, os.getcwd())
import torch as t
from corenet_amd import views as V
from corenet_amd.backend import HipBackend, Transform
from corenet_amd.model import conv_geometry as G
be = HipBackend()
for name, wshape, hw in [("s4_1024_256", (256, 1024, 1, 1), 16), ("s4_256_1024", (1024, 256, 1, 1), 16), ("s2_64_256", (256, 64, 1, 1), 64), ("s5_2048_512", (512, 2048, 1, 1), 8)]:
  cout, cin = wshape[0], wshape[1]
  fwd = G.conv_fwd(wshape, 0)
  w = t.randn(wshape) * 0.05
  idx = t.as_tensor(fwd.index)
  wf = t.where(idx >= 0, w.reshape(-1)[idx.clamp(min=0).long()], t.zeros(())).cuda()
  x = t.randn(4, cin, hw, hw).cuda(); y = t.zeros(4, cout, hw, hw).cuda()
  tr = Transform((t.rand(cin) + 0.5).cuda(), t.randn(cin).cuda(), post_relu=True)
  bias = t.randn(cout).cuda()
  for _ in range(3): be.conv_fwd(V.view_of(x), tr, wf, fwd.npad, bias, 0, V.view_of(y), fwd.window, fwd.pad_lo, 0)
  st = (ctypes.c_longlong * 32)()
  be.lib.cdll.crn_pw_debug_stamps(st)
  print(name, [st[i] for i in range(32) if st[i] > 0])
