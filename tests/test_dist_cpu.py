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


def _train_worker(rank, world, port, out):
  sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank))
  t.set_num_threads(max(1, (os.cpu_count() or 2) // world))
  from corenet_amd import distributed as D
  from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
  from kernel_contract_emu import EmuBackend
  from oracle import corenet_oracle as O
  D.init_from_env("gloo")
  model = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cpu", backend=EmuBackend())
  model.load_state_dict(O.make_state(0, 2, nbt=0)); model.train()
  if rank == 1:                                   # replicas start from rank 0's buffers (DDP broadcast_buffers)
    model.engine.store.buffers.add_(1.0)
  D.broadcast_buffers(model.engine.store)
  image, v2s, off, grid = O.synthetic_batch(1, seed=rank, num_classes=2)       # a different sample per rank
  # rank 0's BatchRenorm buffers ride on the first gradient bucket (GradientSync.attach): after the step every rank holds
  # the running statistics rank 0 computed from ITS sample, as DDP's broadcast before the next forward would deliver
  sync = D.GradientSync(world).attach(model.engine)
  before = model.engine.store.params.clone()
  buf0 = model.engine.store.buffers.clone()
  assert sync.needs_buffer_broadcast              # attach() is collective: every rank owes (and enters) the start-up broadcast
  loss = model.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", world_size=world, all_reduce=sync)
  assert not sync.needs_buffer_broadcast
  rec = {"p": model.engine.store.params.clone(), "g": model.engine.store.grads.clone(), "before": before,
         "buf": model.engine.store.buffers.clone(), "buf_before": buf0,
         "loss": float(loss), "pushed": list(sync.pushed), "buckets": [(lo, hi) for _, lo, hi in model.engine.grad_buckets]}
  # ONE rank reloads state after attach() (a checkpoint resumed on rank 0 only).  Nothing rank-local may gate a
  # collective: the next step must run its bucket all-reduces on both ranks (no hang), and rank 0's reloaded statistics --
  # stepped by its forward -- reach rank 1 with that step's first bucket.
  if rank == 0:
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    for k in sd:
      if k.endswith("running_mean"):
        sd[k] += 0.25
    model.load_state_dict(sd)
  assert not sync.needs_buffer_broadcast          # a rank-local reload does not raise the collective flag
  sync.pushed.clear()
  model.train_step(image, v2s, off, grid.to(t.int32), "iou_fgbg", world_size=world, all_reduce=sync)
  rec.update(buf2=model.engine.store.buffers.clone(), p2=model.engine.store.params.clone(), pushed2=list(sync.pushed))
  t.save(rec, os.path.join(out, f"t{rank}.pt"))
  dist.barrier(); dist.destroy_process_group()


def test_two_rank_overlapped_train_step(tmp_path):
  """Whole data-parallel step on two gloo ranks over the CPU contract emulator: forward, loss, bucketed backward
  with the gradient buckets all-reduced as they become final, Adam with 1/world.  Different samples per rank, so
  the local gradients differ; after the step the summed gradient slab and the parameters are bit-identical on both
  ranks, every bucket was exchanged, and the parameters moved."""
  world, port = 2, 29433 + os.getpid() % 200
  mp.spawn(_train_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  a, b = t.load(tmp_path / "t0.pt"), t.load(tmp_path / "t1.pt")
  assert a["loss"] != b["loss"] and all(map(lambda v: v == v, (a["loss"], b["loss"])))     # different samples, finite
  assert t.equal(a["g"], b["g"]) and t.equal(a["p"], b["p"])
  assert a["pushed"] == [hi - lo for lo, hi in a["buckets"]] and sum(a["pushed"]) == a["g"].numel()
  assert t.equal(a["buf"], b["buf"]) and not t.equal(a["buf"], a["buf_before"])         # rank 0's stepped statistics everywhere
  moved = (a["p"] - a["before"]).abs()
  assert float(moved.max()) > 1e-5 and float(moved.max()) <= 4e-4 * 1.01      # one Adam step of lr 4e-4
  # second step, after rank 0 alone reloaded its state: every bucket exchanged on both ranks, rank 0's (shifted, then stepped)
  # running statistics on both
  assert a["pushed2"] == a["pushed"] and b["pushed2"] == b["pushed"]
  assert t.equal(a["buf2"], b["buf2"]) and t.equal(a["p2"], b["p2"])
  assert float((a["buf2"] - a["buf"]).abs().max()) > 0.2

