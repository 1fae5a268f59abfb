"""World-size-2 gloo test of the data-parallel step logic (corenet_amd/distributed.py):
chunked all-reduce of the flat gradient slab, 1/world folded into Adam, buffer
broadcast from rank 0, confusion-matrix reduce."""
import os
import sys

import torch as t
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
  sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank))
  from corenet_amd import distributed as D
  from corenet_amd.model.engine import ParamStore, Engine
  from kernel_contract_emu import EmuBackend
  r, l, w = D.init_from_env("gloo")
  assert (r, w) == (rank, world)
  specs = [("a.weight", (7, 5, 3), "param"), ("a.bias", (7,), "param"), ("a.running_mean", (7,), "buffer"),
           ("a.num_batches_tracked", (), "nbt")]
  store = ParamStore(specs, "cpu")
  g = t.Generator().manual_seed(0)
  store.params.copy_(t.randn(store.params.shape, generator=g))
  store.buffers.fill_(float(rank + 1)); store.nbt.fill_(rank + 5)
  D.broadcast_buffers(store)
  assert float(store.buffers[0]) == 1.0 and int(store.nbt[0]) == 5
  store.grads.copy_(t.arange(store.grads.numel(), dtype=t.float32) * (rank + 1))
  D.GradientSync(world, chunks=3)(store.grads)
  # the overlapped form: buckets pushed from the top of the slab down, joined by wait()
  g2 = t.arange(store.grads.numel(), dtype=t.float32) * (rank + 1)
  sync = D.GradientSync(world)
  assert sync.overlap
  n = g2.numel()
  for lo, hi in ((n - 40, n), (16, n - 40), (0, 16)):
    sync.push(g2[lo:hi])
  sync.wait()
  assert t.equal(g2, store.grads) and sync.pushed == [40, n - 56, 16]
  be = EmuBackend()
  m, v = t.zeros_like(store.params), t.zeros_like(store.params)
  be.adam_step(store.params, store.grads, m, v, store.params.numel(), 4e-4, 0.9, 0.999, 1e-4, 1.0 / world, 1)
  cm = t.full((3, 3), float(rank + 1)); D.reduce_confusion_matrix(cm)
  t.save({"p": store.params.clone(), "g": store.grads.clone(), "cm": cm}, os.path.join(out, f"r{rank}.pt"))
  dist.barrier(); dist.destroy_process_group()


def test_two_rank_gradient_sync(tmp_path):
  world, port = 2, 29533 + os.getpid() % 200
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  a, b = t.load(tmp_path / "r0.pt"), t.load(tmp_path / "r1.pt")
  n = a["g"].numel()
  assert t.equal(a["g"], t.arange(n, dtype=t.float32) * 3) and t.equal(a["g"], b["g"])
  assert t.equal(a["p"], b["p"])                       # replicas stay bit-identical
  g = t.Generator().manual_seed(0); p0 = t.randn(n, generator=g)
  pt = p0.clone().requires_grad_(True); opt = t.optim.Adam([pt], lr=4e-4, eps=1e-4)
  pt.grad = t.arange(n, dtype=t.float32) * 1.5; opt.step()     # mean gradient over the two ranks
  assert float((a["p"] - pt.detach()).abs().max()) < 1e-6
  assert float(a["cm"][0, 0]) == 3.0                   # reduced onto rank 0
