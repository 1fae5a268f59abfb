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
  model.load_state_dict(O.make_state(0, 2, nbt=30000)); model.train()       # (nbt 30000: the r / d clamps read the running statistics)
  image, v2s, off, grid = [x.cuda() for x in O.synthetic_batch(1, seed=rank, num_classes=2)]
  grid = grid.to(t.int32)
  sync = D.GradientSync(world).attach(model.engine)      # rank 0's BatchRenorm buffers ride on the first bucket
  assert sync.overlap and model.engine.plan(1).side is not None
  # rank 1 starts from DIFFERENT running statistics (a checkpoint resumed on one rank, a load_state_dict on another): the exchange owes
  # one broadcast of rank 0's buffers before the first forward (DDP broadcast_buffers=True, pipeline.py:199) -- paid by train_step
  if rank == 1:
    model.engine.store.buffers.add_(0.3)
  assert sync.needs_buffer_broadcast
  losses = []
  for _ in range(3):
    losses.append(float(model.train_step(image, v2s, off, grid, "iou_fgbg", world_size=world, all_reduce=sync)))
  assert not sync.needs_buffer_broadcast
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
  assert sum(a["pushed"]) == a["n"] and len(a["pushed"]) == 8
  assert t.equal(a["buf"], b["buf"])                                   # ... and the same running statistics (rank 0's)
  assert a["losses"][0] != b["losses"][0]                              # different samples per rank
  assert a["losses"][-1] < a["losses"][0] and b["losses"][-1] < b["losses"][0]
  # rank 1's first forward ran on rank 0's running statistics, not on its own perturbed ones: its first loss is the oracle's loss
  # on its sample with the clean state (perturbed statistics move it by ~1e-1 through the r / d clamps)
  from oracle import corenet_oracle as O
  image, v2s, off, grid = O.synthetic_batch(1, seed=1, num_classes=2)
  want = float(O.iou_fgbg(grid, O.corenet_forward(O.make_state(0, 2, nbt=30000), image, v2s, off, training=True)))
  assert abs(b["losses"][0] - want) < 2e-4 * abs(want), (b["losses"][0], want)


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
  assert d["rccl"]["ranks"] == 2 and d["rccl"]["backend"] == "gloo" and len(d["rccl"]["buckets_mb"]) == 8
  assert d["rccl"]["transport"] == "torch.distributed/gloo" and d["rccl"]["buffers_on_first_bucket"] is True
  assert len(d["rccl"]["exposed_ms_per_bucket"]) == 8 and all(v >= 0 for v in d["rccl"]["exposed_ms_per_bucket"])
  assert "NCCL_ALGO" in d["rccl"] and d["rccl"]["exposed_exchange_ms"] >= 0
  # the replicas hold the same parameters and buffers after the timed steps (MIN == MAX of their checksums over the
  # communicator), and the line names the communicator's own size
  assert d["rccl"]["replicas"] == {"equal": True, "comm_ranks": 2, "checksum_spread": 0.0}
  assert "cpu_baseline" not in d


def _rccl_worker(rank, world, port, out):
  """One rank per GPU over real RCCL: three overlapped training steps with torch.distributed's transport, then three more with the
  library's own communicator (crn_allreduce_f32) on a second model that starts from the same state."""
  sys.path.insert(0, ROOT)
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                    LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
  from corenet_amd import distributed as D
  from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
  from oracle import corenet_oracle as O
  D.init_from_env("nccl")
  dev = f"cuda:{rank}"
  image, v2s, off, grid = [x.to(dev) for x in O.synthetic_batch(1, seed=rank, num_classes=2)]
  grid = grid.to(t.int32)
  res = {}
  for name, native in (("torch", False), ("native", True)):
    model = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device=dev)
    model.load_state_dict(O.make_state(0, 2, nbt=30000)); model.train()
    if rank == 1:
      model.engine.store.buffers.add_(0.3)
    sync = D.GradientSync(world, native=native).attach(model.engine)
    assert (sync.native is not None) == native
    losses = [float(model.train_step(image, v2s, off, grid, "iou_fgbg", world_size=world, all_reduce=sync)) for _ in range(3)]
    t.cuda.synchronize()
    st = model.engine.store
    chk = t.stack([st.params.double().sum(), st.params.double().abs().sum(), st.buffers.double().sum()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    res[name] = {"p": st.params.cpu(), "buf": st.buffers.cpu(), "losses": losses, "replicas_equal": bool(t.equal(lo, hi)),
                 "describe": sync.describe()}
    if sync.native is not None:
      sync.native.close()
  t.save(res, os.path.join(out, f"r{rank}.pt"))
  dist.barrier(); dist.destroy_process_group()


def test_two_real_rccl_ranks_torch_and_native_transport(tmp_path):
  """First contact with a multi-GPU node (skipped on the one-GPU driver box): two ranks on two GPUs over RCCL / xGMI, three
  overlapped training steps through GradientSync(native=False) (torch.distributed, backend "nccl" = RCCL) and through
  GradientSync(native=True) (the library's communicator, csrc/comm_rccl.hip): the replicas are bit-identical after the steps
  (parameters and BatchRenorm buffers, pipeline.py:199-200 semantics incl. the start-up broadcast: rank 1 starts from perturbed
  running statistics), the MIN == MAX checksum test of bench.py's `rccl.replicas` agrees, and both transports land on the
  same parameters (a sum of two addends is the same in either order)."""
  if t.cuda.device_count() < 2:
    pytest.skip("needs two GPUs (RCCL cannot host two ranks on one device); the gloo tests above cover the step's wiring")
  world, port = 2, 29100 + os.getpid() % 200
  mp.spawn(_rccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
  a, b = t.load(tmp_path / "r0.pt"), t.load(tmp_path / "r1.pt")
  for name in ("torch", "native"):
    assert t.equal(a[name]["p"], b[name]["p"]) and t.equal(a[name]["buf"], b[name]["buf"]), name
    assert a[name]["replicas_equal"] and b[name]["replicas_equal"], name
    assert a[name]["describe"]["ranks"] == 2 and a[name]["describe"]["buffers_on_first_bucket"]
  assert "native RCCL" in a["native"]["describe"]["transport"] and "nccl" in a["torch"]["describe"]["transport"]
  assert t.equal(a["torch"]["p"], a["native"]["p"])


def test_scale_probe_dry_run_output_format(tmp_path):
  """tools/scale_probe.sh is the ONE command to run on first contact with a multi-GPU node; its dry-run mode (two ranks share this
  GPU over gloo, two steps) runs every command of the real thing except the RCCL ones and must leave the files and the summary
  lines the 1 -> 8 scaling question is answered from."""
  import json, subprocess
  out = str(tmp_path / "probe")
  r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_probe.sh"), out], env=dict(os.environ, SCALE_PROBE_DRY="1"),
                     capture_output=True, text=True, timeout=1500, cwd=ROOT)
  assert r.returncode == 0, r.stderr[-2000:]
  for f in ("bench_n1.json", "bench_n2.json", "bench_n2_oneslab.json", "allreduce_default.json", "summary.txt"):
    assert os.path.exists(os.path.join(out, f)), (f, os.listdir(out))
  d2 = json.loads([l for l in open(os.path.join(out, "bench_n2.json")) if l.startswith("{")][-1])
  assert d2["n_gpus"] == 2 and d2["rccl"]["replicas"]["equal"] is True
  summ = open(os.path.join(out, "summary.txt")).read()
  lines = [l for l in summ.splitlines() if l.startswith("bench_n")]
  assert len(lines) == 3 and all("ms/step" in l and "efficiency" in l and "exposed exchange" in l for l in lines), summ
  assert "efficiency 1.000" in [l for l in lines if l.startswith("bench_n1.json")][0]
  assert any(l.startswith("allreduce_default.json") and "torch us" in l for l in summ.splitlines()), summ
