"""Two data-parallel ranks on ONE GPU (gloo over CUDA tensors: a single-GPU box cannot host two RCCL ranks):
the real HIP train step with the bucketed backward, the side stream and asynchronous all-reduces of the slab
slices, world size 2.  RCCL itself is exercised by tests/test_model_gpu.py (one rank) and by bench.py on a node."""
import os
import sys

import pytest
import torch as t
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
  sys.path.insert(0, ROOT)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK="0")
  from corenet_amd import distributed as D
  from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
  from oracle import corenet_oracle as O
  t.cuda.set_device(0)
  D.init_from_env("gloo")
  model = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device="cuda:0")
  model.load_state_dict(O.make_state(0, 2, nbt=0)); model.train()
  image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(1, seed=rank, num_classes=2)]
  grid = grid.to(t.int32)
  sync = D.GradientSync(world).attach(model.engine)      # rank 0's BatchRenorm buffers ride on the first bucket
  assert sync.overlap and model.engine.plan(1).side is not None
  D.broadcast_buffers(model.engine.store)
  losses = []
  for _ in range(3):
    losses.append(float(model.train_step(image, v2s, off, grid, "iou_fgbg", world_size=world, all_reduce=sync)))
  t.cuda.synchronize()
  t.save({"p": model.engine.store.params.cpu(), "g": model.engine.store.grads.cpu(), "losses": losses,
          "buf": model.engine.store.buffers.cpu(),
          "pushed": list(sync.pushed), "n": model.engine.store.grads.numel()}, os.path.join(out, f"g{rank}.pt"))
  dist.barrier(); dist.destroy_process_group()


def test_two_ranks_one_gpu_overlapped_train_steps(tmp_path):
  world, port = 2, 29333 + os.getpid() % 200
  mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  a, b = t.load(tmp_path / "g0.pt"), t.load(tmp_path / "g1.pt")
  assert t.equal(a["g"], b["g"]) and t.equal(a["p"], b["p"])          # same summed gradients, same parameters
  assert sum(a["pushed"]) == a["n"] and len(a["pushed"]) == 7
  assert t.equal(a["buf"], b["buf"])                                   # ... and the same running statistics (rank 0's)
  assert a["losses"][0] != b["losses"][0]                              # different samples per rank
  assert a["losses"][-1] < a["losses"][0] and b["losses"][-1] < b["losses"][0]


def test_bench_two_rank_launch_path_dry_run():
  """The driver's N > 1 command line, `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
  127.0.0.1 --master-port P bench.py --gpus 2 ...`, on this one-GPU box with CRN_DIST_BACKEND=gloo (two ranks
  share the GPU): rank 0 prints ONE JSON line with the contract's keys, n_gpus 2, whole-job throughput, the
  exchange description, and no CPU baseline on N > 1."""
  import json, subprocess
  env = dict(os.environ, CRN_DIST_BACKEND="gloo")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
         "127.0.0.1", "--master-port", str(29600 + os.getpid() % 300), os.path.join(ROOT, "bench.py"), "--gpus", "2",
         "--steps", "2", "--warmup", "1"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout[-2000:]
  d = json.loads(lines[0])
  for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"):
    assert k in d, k
  assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8
  assert abs(d["value"] - 8 * 128 ** 3 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
  assert d["rccl"]["ranks"] == 2 and d["rccl"]["backend"] == "gloo" and len(d["rccl"]["buckets_mb"]) == 7
  assert d["rccl"]["transport"] == "torch.distributed/gloo" and d["rccl"]["buffers_on_first_bucket"] is True
  assert len(d["rccl"]["exposed_ms_per_bucket"]) == 7 and all(v >= 0 for v in d["rccl"]["exposed_ms_per_bucket"])
  assert "NCCL_ALGO" in d["rccl"] and d["rccl"]["exposed_exchange_ms"] >= 0
  # the replicas hold the same parameters and buffers after the timed steps (MIN == MAX of their checksums over the
  # communicator), and the line names the communicator's own size
  assert d["rccl"]["replicas"] == {"equal": True, "comm_ranks": 2, "checksum_spread": 0.0}
  assert "cpu_baseline" not in d
