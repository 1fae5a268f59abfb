#!/usr/bin/env python
"""Headline benchmark (BASELINE.json: voxels/sec fwd+bwd @128^3).

One "step" = the reference's training hot loop body (pipeline.py:224-233) on one
synthetic batch already resident in HBM: CoReNet forward -> iou_fgbg loss ->
backward -> gradient all-reduce (RCCL, N>1) -> Adam, B=4 samples of 256x256 RGB
-> 128^3 voxels per GPU (configs/models/h7.json5:42,62-67).  value = global_batch * 128^3 * steps / time.

Arithmetic: the headline line is the product's default math mode, decoder_math="bf16x3" (fp32 tensors everywhere;
the convolutions of decoder stages 3-6 and the encoder's 3x3 layers multiply operands split into two bf16 terms,
three bf16 MFMAs per product, fp32 accumulation -- wider than the "bf16" BASELINE.json names for h7, narrower than
the reference's fp32; inside north_star's 1e-3 on every reference fixture).  The same step with every convolution
on the fp32 MFMA engine is timed in the same run and printed under "fp32_math"; "parity" holds the eval-mode logits
error of BOTH modes against the oracle on the bench's own weights and inputs (rank 0, after the timed region).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch as t  # noqa: E402
import torch.distributed as dist  # noqa: E402

# algorithmic work: per sample
CONV6_FLOP = 2 * 64 ** 3 * 28 * 125 * 16          # stage_6 c1: Conv3d 28->16 k5 @64^3
RAY64_BYTES = 64 ** 3 * 12 * 4 + 64 * 64 * 12 * 4  # ray-sample 64^3 x 12ch: output + map
RAY64_IDX_BYTES = 64 ** 3 * 2                      # ... + the saved index tensor (uint16 per voxel) the scatter reads
PEAK_F32_MFMA = 157.3e12                           # MI355X_MICROARCH.md: fp32 matrix peak
PEAK_BF16_MFMA = 2500e12                           # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM = 8.0e12                                  # HBM3E spec peak
# HBM bytes per launch come from separate rocprofv3 --pmc passes of this command (tools/pmc_traffic.sh); the record is
# keyed by kernel name + grid, so both are named here, next to the kernel they describe
TRAFFIC_FILES = {"bf16x3": "r06_pmc_traffic.json", "fp32": "r06_pmc_traffic_fp32.json", "bf16x3_c14": "r06_pmc_traffic_c14.json"}
# decoder stage_6.t1 forward (ConvTranspose3d 16 -> C, k 7, stride 2 @64^3 -> 128^3; reconstruction_decoder.py:89-95): real FLOP per sample and class
CONVT6_FLOP_PER_CLASS = 2.0 * 64 ** 3 * 16 * 343
CONV_BF3_TRAFFIC_KERNEL, CONV_BF3_THREADS = "conv_bf3_half_kernel<1, 1, 7, 1", 512


def canonical_camera():
  """The dataset's canonical camera (doc/data_format_and_coordinate_systems.md:103-111)."""
  import math
  from corenet_amd.geometry import transformations as T
  return T.perspective_rh(math.radians(60.0), 1.0, 1e-4, 10.0) @ T.look_at_rh(
      [0.5, 0.5, -0.8666666], [0.5, 0.5, 0.5], [0, -1, 0])


def synthetic_batch(batch, seed, num_classes):
  """SURVEY 8(d) synthetic inputs: seeded uint8 256x256 images, the canonical camera, v2s = camera @ scale(1/128),
  sampling offset 0.5, ground truth = analytic ball(s)."""
  g = t.Generator().manual_seed(1000 + seed)
  image = t.randint(0, 256, (batch, 3, 256, 256), generator=g, dtype=t.uint8)
  v2s = (canonical_camera() @ t.diag(t.tensor([1 / 128.0] * 3 + [1.0])))[None].expand(batch, 4, 4).contiguous()
  offset = t.full((batch, 3), 0.5)
  ax = t.arange(128, dtype=t.float32) + 0.5
  zz, yy, xx = t.meshgrid(ax, ax, ax, indexing="ij")
  grid = t.zeros((batch, 128, 128, 128), dtype=t.int64)
  nballs = 1 if num_classes == 2 else 3
  for b in range(batch):
    for k in range(nballs):
      cx = (0.5 + 0.22 * (k - (nballs - 1) / 2)) * 128
      r = (0.3 if nballs == 1 else 0.1) * 128
      grid[b][((xx - cx) ** 2 + (yy - 64) ** 2 + (zz - 64) ** 2) <= r * r] = \
          1 if num_classes == 2 else (1 + (3 * b + k) % (num_classes - 1))
  return image, v2s, offset, grid


def synthetic_meshes(batch, dev):
  """UV-sphere meshes of 20 k triangles scaled into the unit cube, three per sample; returns the triangles
  (view space), triangles per mesh and the per-mesh view->voxel matrices (batched_example.py:153-160)."""
  import numpy as np
  from corenet_amd.data import batched_example as BE
  nlat = nlon = 100
  th, ph = np.linspace(0, np.pi, nlat + 1), np.linspace(0, 2 * np.pi, nlon + 1)
  unit = np.stack([np.outer(np.sin(th), np.cos(ph)), np.outer(np.sin(th), np.sin(ph)),
                   np.outer(np.cos(th), np.ones_like(ph))], -1)                       # [nlat+1, nlon+1, 3]
  a, b, c, d = unit[:-1, :-1], unit[1:, :-1], unit[1:, 1:], unit[:-1, 1:]
  sphere = np.concatenate([np.stack([a, b, c], 2), np.stack([a, c, d], 2)], 1).reshape(-1, 3, 3)   # 20000 triangles
  rng = np.random.RandomState(0)
  meshes = [(0.3 + 0.4 * rng.rand(3) + (0.08 + 0.1 * rng.rand()) * sphere).astype(np.float32) for _ in range(3 * batch)]
  tris = t.tensor(np.concatenate(meshes)).to(dev)
  nt = t.tensor([len(m) for m in meshes], dtype=t.int32)
  v2x = BE.view2voxel_matrices(t.full((batch, 3), 0.5), (128,) * 3)
  mv = t.cat([v2x[b:b + 1].expand(3, 4, 4) for b in range(batch)])
  return tris, nt, mv


def load_traffic(math, B, C, want):
  """HBM bytes per launch from the PMC passes of this same command (profiles/r06_pmc_traffic*.json; FETCH_SIZE / WRITE_SIZE
  need their own rocprofv3 runs and cannot be read from inside the process).  Returns ({key: bytes}, file name)."""
  traffic = {}
  traffic_file = os.path.join("profiles", TRAFFIC_FILES[math])
  try:
    tj = json.load(open(os.path.join(ROOT, traffic_file)))
  except (OSError, ValueError) as e:
    print(f"bench.py: no HBM-traffic record ({traffic_file}: {e}); roofline.traffic = null", file=sys.stderr)
    return traffic, traffic_file
  if B == 4 and C == 14:
    for k, v in tj.get("kernels", {}).items():
      if "convt_par_fwd_kernel" in k and k.endswith(f"grid {2048 * 512}"):
        traffic["convt"] = v["hbm_bytes"]
    if "convt" in want and "convt" not in traffic:
      print(f"bench.py: {traffic_file} has no record of convt_par_fwd_kernel -- re-run tools/pmc_traffic.sh; traffic = null", file=sys.stderr)
  if B == 4 and C == 2:
    for k, v in tj.get("kernels", {}).items():
      if math == "bf16x3":
        # stage_6.c1 fwd is the only launch of conv_bf3_half_kernel<NSUB 1, unit-stride x, 5x5 plane> on 2048 tiles
        if CONV_BF3_TRAFFIC_KERNEL in k and k.endswith(f"grid {2048 * CONV_BF3_THREADS}"):
          traffic["conv"] = v["hbm_bytes"]
      # stage_6.c1 fwd (fp32 engine): conv_fwd_kernel<8,1,xvec>, 2048 tiles x 1 N-block.  stage_5.t1 and stage_6.t1 fwd
      # share that (kernel, grid); per step the dispatch order is s5.t1, s6.c1, s6.t1 -> every 3rd from 1
      elif "conv_fwd_kernel<8, 1, 1>" in k and k.endswith(f"grid {2048 * 256}"):
        pl = v.get("per_launch_hbm_bytes", [])
        if len(pl) >= 3 and len(pl) % 3 == 0:
          mine = pl[1::3]
          traffic["conv"] = sum(mine) / len(mine)
      if "ray_sample_fwd_kernel" in k and k.endswith("grid 262144"):        # 64^3 x 12 ch
        traffic["ray"] = v["hbm_bytes"]
      if "ray_scatter_kernel<4, 8" in k and k.endswith(f"grid {128 * 3 * 4 * 256}"):   # 64^3: (2 x 8 tiles x 8 z segments) x 3 channel groups x B of 256 threads
        traffic["ray_bwd"] = v["hbm_bytes"]
      if "fill_fused_kernel" in k:
        traffic["fill"] = v["hbm_bytes"]
    missing = [k for k in want if k not in traffic]
    if missing:
      # a kernel was renamed / re-gridded since the PMC passes were taken: say so instead of printing a stale or
      # silently empty number (tools/pmc_traffic.sh regenerates the file)
      print(f"bench.py: {traffic_file} has no record for {missing} (kernel names / grids changed?) -- "
            f"re-run tools/pmc_traffic.sh; roofline.traffic = null for those", file=sys.stderr)
  return traffic, traffic_file


def host_threads():
  try:
    avail = len(os.sched_getaffinity(0))
  except AttributeError:
    avail = os.cpu_count() or 1
  return avail


def cpu_baseline(state, batch, loss_name, seconds_budget=25.0):
  """The oracle (torch-CPU restatement of the reference, validated against the imported reference in
  oracle/gen_golden.py) timed on this host's cores on the bench's own weights and inputs: fwd+loss+bwd of the
  bench batch (B=4) when one step fits the budget, else of its first sample."""
  from oracle import corenet_oracle as O
  # oneDNN conv3d oversubscribes badly on many-core hosts (256 threads: 340 s/step measured);
  # 16 threads is the fastest setting found and is what `cores` reports.
  avail = host_threads()
  nthreads = max(1, min(avail, 16))
  t.set_num_threads(nthreads)
  sd = {k: v.detach().cpu().clone() for k, v in state.items()}
  for k in sd:
    if sd[k].dtype == t.float32 and "running" not in k:
      sd[k].requires_grad_(True)
  image, v2s, off, grid = [x.cpu() for x in batch]
  grid = grid.long()
  def step(n):
    for v in sd.values():
      v.grad = None
    loss = getattr(O, loss_name)(grid[:n], O.corenet_forward(sd, image[:n], v2s[:n], off[:n], training=True))
    loss.backward()
  t0 = time.time(); step(1); warm1 = time.time() - t0       # warm-up (also sizes the sample)
  B = image.shape[0] if warm1 * image.shape[0] * 2.5 < seconds_budget else 1
  n, t0 = 0, time.time()
  while n < 2 or (time.time() - t0 < seconds_budget - 2 * warm1 * B and n < 20):
    step(B); n += 1
  dt = (time.time() - t0) / n
  return {"value": B * 128 ** 3 / dt, "unit": "voxels/s", "cores": nthreads, "threads_used": nthreads,
          "host_cores": os.cpu_count(), "host_cores_available": avail, "kind": "port", "batch": B,
          "sample": f"{n} steps of B={B} fwd+loss+bwd (oracle/corenet_oracle.py, torch-CPU fp32, the bench's own "
                    f"weights and inputs)"}


def parity_check(state, batch, classes, dev, headline_math):
  """Eval-mode logits of both math modes of the library against the oracle, on the bench's own weights and inputs,
  every voxel of the bench batch (the oracle is the checker here, never the thing measured)."""
  from oracle import corenet_oracle as O
  from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
  image, v2s, off, _ = batch
  with t.no_grad():
    want = O.corenet_forward({k: v.detach().cpu().clone() for k, v in state.items()}, image, v2s, off, training=False)
    out = {}
    for math in dict.fromkeys((headline_math, "fp32")):
      m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), classes, 2, 64, 0.75)), device=dev, decoder_math=math)
      m.load_state_dict(state); m.eval()
      got = m(image.to(dev), v2s.to(dev), off.to(dev)).cpu()
      out[math] = float((got.double() - want.double()).abs().max() / want.double().abs().max())
      del m
  return {"eval_logits_max_rel_err_vs_oracle": out, "voxels_compared": int(want.numel()), "tolerance": 1e-3,
          "note": "north_star: logits within 1e-3 relative; eval mode (running statistics), B = the bench batch"}


def super_resolution_leg(state, batch_cpu, dev, math, reps=5):
  """h7 at 256^3 (BASELINE configs[4]) = x2 super-resolution, super_resolution.py:92-112: 8 native passes at shifted sampling
  offsets, interleaved.  Timed: the drop-in with encoder reuse (the encoder does not depend on the offset in eval mode: 1 encoder
  + 8 decoder passes, softmax + interleave in one kernel) and the reference's schedule on the same kernels (8 full forwards).
  Parity: pmf of sample 0 against the oracle's 8-pass result on the same weights and image, every one of its 2 x 256^3 values."""
  from corenet_amd import super_resolution as SR
  from corenet_amd.geometry import transformations as T
  from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig
  from oracle import corenet_oracle as O
  image, _, off, _ = batch_cpu
  B = image.shape[0]
  m = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 2, 2, 64, 0.75)), device=dev, decoder_math=math)
  m.load_state_dict(state); m.eval()

  class _State: pass
  st = _State(); st.model = m
  sr = SR.super_resolution_from_state(st)
  cam = canonical_camera()[None].expand(B, 4, 4).contiguous().to(dev)
  v2v = T.scale([256.0] * 3)[None].expand(B, 4, 4).contiguous().to(dev)      # view -> voxel of the 256^3 output grid
  go, img = off.to(dev), image.to(dev)
  e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
  for _ in range(2):
    pmf = sr(img, cam, v2v, go, (256, 256, 256))
  e0.record()
  for _ in range(reps):
    pmf = sr(img, cam, v2v, go, (256, 256, 256))
  e1.record(); t.cuda.synchronize()
  reuse_s = e0.elapsed_time(e1) / reps * 1e-3
  native = sr.get_native_offsets((256, 256, 256), go)                             # [8, B, 3]
  vs = cam @ (v2v @ T.scale([0.5] * 3).to(dev)).inverse()
  with t.no_grad():
    for _ in range(2):
      for n in range(8):
        m(img, vs, native[n])
    e0.record()
    for _ in range(reps):
      for n in range(8):
        m(img, vs, native[n])
    e1.record(); t.cuda.synchronize()
  eight_s = e0.elapsed_time(e1) / reps * 1e-3
  with t.no_grad():                                                              # the oracle's 8 passes on sample 0
    t.set_num_threads(max(1, min(host_threads(), 16)))           # (oneDNN oversubscribes on many-core hosts: see cpu_baseline)
    want = t.empty(2, 256, 256, 256)
    for n in range(8):
      iz, iy, ix = n // 4, (n // 2) % 2, n % 2
      lg = O.corenet_forward({k: v.clone() for k, v in state.items()}, image[:1], vs[:1].cpu(), native[n, :1].cpu(), training=False)
      want[:, iz::2, iy::2, ix::2] = lg.softmax(1)[0]
  err = float((pmf[0].cpu() - want).abs().max())
  del m
  return {"config": "h7 at 256^3: x2 super-resolution (8 sampling offsets interleaved), C=2, B=%d, eval mode, decoder_math=%s" % (B, math),
          "ms_per_batch": reuse_s * 1e3, "value": B * 256 ** 3 / reuse_s, "unit": "output voxels/s",
          "schedule": "encoder once + 8 decoder passes + fused softmax/interleave (CoreNet.multi_offset_pmf)",
          "eight_full_forwards_ms": eight_s * 1e3, "speedup_vs_eight_full_forwards": eight_s / reuse_s,
          "parity": {"max_abs_pmf_err_vs_oracle_8_pass": err, "values_compared": 2 * 256 ** 3, "sample": 0, "tolerance": 1e-3,
                     "note": "pmf in [0, 1]: absolute error; oracle = oracle/corenet_oracle.py forward per offset + softmax + interleave (super_resolution.py:105-112)"}}


def cpu_baseline_fill(shells_cpu, seconds_budget=5.0):
  """fill_voxels on the host: the library's own C++ twin (crn_fill_voxels_cpu, csrc/fill_voxels_cpu.cpp; one
  host thread per grid), same 12 x 128^3 shells as the GPU leg (SURVEY 8(d) last row)."""
  from corenet_amd.cc import fill_voxels
  fill_voxels.fill_inside_voxels_cpu(shells_cpu)
  n, t0 = 0, time.time()
  while n < 3 or (time.time() - t0 < seconds_budget and n < 50):
    fill_voxels.fill_inside_voxels_cpu(shells_cpu); n += 1
  dt = (time.time() - t0) / n
  threads = min(shells_cpu.shape[0], host_threads())
  return {"value": shells_cpu.numel() / dt, "unit": "voxels/s", "cores": threads, "threads_used": threads,
          "host_cores": os.cpu_count(), "kind": "port (library's own C++ operator fill_inside_voxels_cpu)",
          "sample": f"{n} calls on {shells_cpu.shape[0]} x 128^3 fp32 shells"}


GRAPH = os.environ.get("CRN_GRAPH", "0") == "1"      # replay the captured HIP graph of the step (off: measured slower)


_T0 = time.time()


def progress(msg):
  """Timestamped progress on stderr (stdout carries the one JSON line only)."""
  if os.environ.get("RANK", "0") == "0":
    print("bench.py [%6.1f s] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--batch", type=int, default=4, help="samples per GPU (h7.json5:42)")
  ap.add_argument("--classes", type=int, default=2, help="2 = h7 (FG/BG); 14 = m7/m9")
  ap.add_argument("--math", default=os.environ.get("CRN_DECODER_MATH", "bf16x3"), choices=["fp32", "bf16x3"],
                  help="decoder stage 4-6 convolutions: fp32 MFMA, or split-bf16 (3 bf16 MFMAs per product, fp32 accumulate)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-fp32-side", action="store_true", help="skip the fp32-math run printed beside the headline (profiling)")
  ap.add_argument("--no-m9-side", action="store_true", help="skip the 14-class (m7 / m9) step printed beside the headline")
  ap.add_argument("--no-sr-side", action="store_true", help="skip the x2 super-resolution (h7 at 256^3) leg")
  args = ap.parse_args()

  from corenet_amd import distributed as D
  from corenet_amd.model.core_net import CoreNet, CoreNetConfig, DecoderConfig

  rank, local, world = D.init_from_env()
  assert world == args.gpus or world == 1, (world, args.gpus)
  if os.environ.get("CRN_DIST_BACKEND") == "gloo":      # dry run of the N > 1 path on fewer GPUs than ranks
    local = local % t.cuda.device_count()
  t.cuda.set_device(local)
  dev = f"cuda:{local}"
  C, B = args.classes, args.batch
  loss_name = "iou_fgbg" if C == 2 else "xent_times_iou_agnostic"
  model = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), C, 2, 64, 0.75)), device=dev, decoder_math=args.math)
  model.reset_parameters(seed=0)          # the product's own initialiser (resnet50.py:40-47 + torch defaults)
  model.train()
  state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if rank == 0 else None
  batch_cpu = synthetic_batch(B, seed=rank, num_classes=C)
  image, v2s, off, grid = [x.to(dev) for x in batch_cpu]
  grid = grid.to(t.int32)
  # gradient exchange: overlapped buckets; rank 0's BatchRenorm buffers ride on the first bucket (what DDP's per-forward
  # buffer broadcast delivers, without the blocking collective in front of every step); one broadcast up front
  sync = D.GradientSync(world).attach(model.engine)      # (owes one broadcast of rank 0's buffers: paid by the first step)
  sync.probe = world > 1
  plan = model.engine.plan(B)

  def timed(mdl, pl):
    """W warm-up steps, then K steps timed between barrier + synchronize on both sides, max over ranks."""
    def step():
      return mdl.train_step(image, v2s, off, grid, loss_name, lr=4e-4, adam_eps=1e-4, world_size=world,
                            all_reduce=sync if world > 1 else None)
    for _ in range(args.warmup):
      step()
    if world > 1 or not GRAPH:     # launch-by-launch steps: HIP-event probes ride along in the timed region
      pl.probes = {"conv3d_stage6_c1_fwd": [], "ray_sample_fwd_64": [], "grad_exchange_wait": []}
    if world > 1:
      dist.barrier()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
      loss = step()
    t.cuda.synchronize()
    if world > 1:
      dist.barrier()
    dt = time.perf_counter() - t0
    tt = t.tensor([dt], dtype=t.float64, device=dev)
    if world > 1:
      dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if world == 1 and GRAPH:
      # the timed steps were replays of the captured HIP graph (one launch per step), which cannot carry timing
      # events: the per-kernel probes come from 3 launch-by-launch steps on the same buffers right after it
      pl.probes = {"conv3d_stage6_c1_fwd": [], "ray_sample_fwd_64": [], "grad_exchange_wait": []}
      for _ in range(3):
        step()
      t.cuda.synchronize()
    pr = {k: sum(a.elapsed_time(b) for a, b in v) / max(1, len(v)) * 1e-3 for k, v in pl.probes.items()}
    pl.probes = None
    return float(tt), pr, loss

  progress("model built; timing the headline step")
  dt, probes, loss = timed(model, plan)
  progress("headline: %.3f ms per step" % (dt / args.steps * 1e3))
  replicas = None
  if world > 1:
    # data parallelism keeps the replicas identical (pipeline.py:199: DDP averages the gradients, every rank takes the same
    # Adam step): after the timed steps the parameters (and the BatchRenorm buffers, which ride on the first bucket) must
    # have the same checksum on every rank -- MIN == MAX over the communicator, whose own size is reported beside it
    st = model.engine.store
    chk = t.stack([st.params.double().sum(), st.params.double().abs().sum(), st.buffers.double().sum()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    replicas = {"equal": bool(t.equal(lo, hi)), "comm_ranks": dist.get_world_size(),
                "checksum_spread": float((hi - lo).abs().max())}
    if not replicas["equal"] and rank == 0:      # loud, but the measured line is still printed (with equal: false in it)
      print(f"bench.py: REPLICAS DIVERGED after {args.warmup + args.steps} steps: {replicas}", file=sys.stderr)
  fp32_side = None
  if world == 1 and args.math != "fp32" and not args.no_fp32_side:
    # the same step with every convolution on the fp32 MFMA engine (the parity default), printed beside the headline
    m32 = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), C, 2, 64, 0.75)), device=dev, decoder_math="fp32")
    m32.load_state_dict(state0); m32.train()
    dt32, pr32, loss32 = timed(m32, m32.engine.plan(B))
    c32 = pr32["conv3d_stage6_c1_fwd"]
    fp32_side = {"ms_per_step": dt32 / args.steps * 1e3, "value": B * 128 ** 3 * args.steps / dt32, "unit": "voxels/s",
                 "dtype": "f32", "loss": float(loss32),
                 "roofline": {"kernel": "conv_fwd_kernel<8,1,xvec> (stage_6.c1 Conv3d 28->16 k5 @64^3, fwd, fp32 MFMA engine)",
                              "bound": "mfma", "achieved": CONV6_FLOP * B / c32 / 1e12, "peak": PEAK_F32_MFMA / 1e12,
                              "unit": "TFLOP/s", "frac": CONV6_FLOP * B / c32 / PEAK_F32_MFMA, "avg_launch_ms": c32 * 1e3}}
    del m32
    progress("fp32 leg done")
  m9_side = None
  if world == 1 and C == 2 and not args.no_m9_side:
    # the m7 / m9 head (BASELINE configs: 14 classes incl. void, xent_times_iou_agnostic): the same step, same batch
    # and steps, timed in the same run so that the driver's clock covers it too
    m14 = CoreNet(CoreNetConfig(DecoderConfig((128, 128, 128), 14, 2, 64, 0.75)), device=dev, decoder_math=args.math)
    m14.reset_parameters(seed=0); m14.train()
    b14 = synthetic_batch(B, seed=rank, num_classes=14)
    i14, v14, o14, g14 = [x.to(dev) for x in b14]
    g14 = g14.to(t.int32)
    def step14():
      return m14.train_step(i14, v14, o14, g14, "xent_times_iou_agnostic", lr=4e-4, adam_eps=1e-4)
    for _ in range(args.warmup):
      step14()
    pl14 = m14.engine.plan(B)
    pl14.probes = {"conv3d_stage6_t1_fwd": []}         # HIP events around the logits layer's forward launch, inside the timed steps
    t.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
      l14 = step14()
    t.cuda.synchronize(); dt14 = time.perf_counter() - t0
    t1s = sum(a.elapsed_time(b) for a, b in pl14.probes["conv3d_stage6_t1_fwd"]) / max(1, len(pl14.probes["conv3d_stage6_t1_fwd"])) * 1e-3
    pl14.probes = None
    peak14 = PEAK_BF16_MFMA / 3 if args.math == "bf16x3" else PEAK_F32_MFMA
    tr14, tr14_file = load_traffic(args.math + "_c14", B, 14, ("convt",)) if args.math == "bf16x3" else ({}, None)
    walk = m14.engine.convs["decoder.stage_6.t1."].ct_kind == "par"
    m9_side = {"config": "m7/m9: C=14, xent_times_iou_agnostic, B=%d/GPU, decoder_math=%s" % (B, args.math),
               "ms_per_step": dt14 / args.steps * 1e3, "value": B * 128 ** 3 * args.steps / dt14, "unit": "voxels/s",
               "loss": float(l14),
               "roofline": {"kernel": ("convt_par_fwd_kernel (stage_6.t1 ConvTranspose3d 16->14 k7 s2 @64^3 -> 128^3, fwd: the patch of a tile "
                                       "resident in LDS, the 8 output parities walked over exactly their window rows, split-bf16 MFMA)"
                                       if walk else "stage_6.t1 forward on the generic engine"),
                            "bound": "mfma", "achieved": CONVT6_FLOP_PER_CLASS * 14 * B / t1s / 1e12, "peak": peak14 / 1e12,
                            "unit": "TFLOP/s", "frac": CONVT6_FLOP_PER_CLASS * 14 * B / t1s / peak14, "traffic": tr14.get("convt"),
                            "traffic_source": f"{tr14_file} (separate rocprofv3 --pmc passes of `bench.py --classes 14`; not measured in this run)",
                            "avg_launch_ms": t1s * 1e3,
                            "note": "achieved counts the layer's real 2*M*K*N (16 x 343 taps x 14 classes per coarse voxel)"}}
    del m14, pl14
    progress("m7/m9 leg done")
  if rank != 0:
    return
  conv_s, ray_s = probes["conv3d_stage6_c1_fwd"], probes["ray_sample_fwd_64"]
  traffic, traffic_file = load_traffic(args.math, B, C, ("conv", "ray", "ray_bwd", "fill"))
  if fp32_side is not None:
    t32, f32file = load_traffic("fp32", B, C, ("conv",))
    fp32_side["roofline"]["traffic"] = t32.get("conv")
    fp32_side["roofline"]["traffic_source"] = f"{f32file} (separate rocprofv3 --pmc passes of `bench.py --math fp32`; not measured in this run)"
  # ground-truth side: fill_voxels on 3 hollow shells per sample, whole call (one launch) timed with HIP
  # events; the kernel-only time is in profiles/
  ax = t.arange(128, device=dev, dtype=t.float32) - 63.5
  dist3 = (ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2).sqrt()
  shells = t.stack([((dist3 <= r) & (dist3 > r - 1.5)).float() for r in (10, 30, 50)] * B)
  filled = t.empty_like(shells)
  be = model.engine.be
  for _ in range(2):
    be.fill_voxels(shells, filled)
  e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
  # the call is asynchronous; a host-side synchronize on this shared
  # host occasionally takes tens of milliseconds whatever was launched, so: median of 9 timed calls
  fill_times = []
  for _ in range(9):
    e0.record()
    be.fill_voxels(shells, filled)
    e1.record(); t.cuda.synchronize()
    fill_times.append(e0.elapsed_time(e1) * 1e-3)
  fill_s = sorted(fill_times)[len(fill_times) // 2]
  assert bool((filled == t.stack([(dist3 <= r).float() for r in (10, 30, 50)] * B)).all())
  fill_bytes = 8.0 * shells.numel()
  # surface voxelizer (voxelization.py:32-164) on the ground-truth side of the same batch: 3 UV spheres of 20 k triangles per
  # sample, 128^3, image_resolution_multiplier 8 (h7.json5:54).  Algorithmic bytes: the zero-initialised output grids
  # (4 B/voxel, the reference allocates them per call too) + one read of the triangles; the rasterisation itself writes
  # only surface voxels
  from corenet_amd.geometry import voxelization
  vox_tris, vox_nt, vox_mv = synthetic_meshes(B, dev)
  vox_fn = lambda: voxelization.voxelize_mesh(vox_tris, vox_nt, (128,) * 3, vox_mv, image_resolution_multiplier=8)
  for _ in range(2):
    vox_fn()
  e0.record()
  for _ in range(10):
    vox_fn()
  e1.record(); t.cuda.synchronize()
  vox_s = e0.elapsed_time(e1) / 10 * 1e-3
  vox_bytes = 4.0 * vox_nt.numel() * 128 ** 3 + 4.0 * vox_tris.numel()
  # ray-sample gather at 64^3 again as a burst of 20 launches on the step's own buffers: one HIP-event pair
  # around a single ~13 us launch (the in-step probe) also times ~3-4 us of marker / kernel-boundary latency
  k5 = model.engine.skip_ch[5]
  u6 = plan.dec[6]["u"]
  ray_args = (plan.smap[5], plan.smap[5].stride(0), B, k5, 64, 64, plan.layer_mats[3], plan.offset,
              u6[:, u6.shape[1] - k5:], u6.stride(0), 64, 64, 64)
  for _ in range(3):
    be.ray_sample_fwd(*ray_args, map_sC=1, map_sP=k5)
  e0.record()
  for _ in range(20):
    be.ray_sample_fwd(*ray_args, map_sC=1, map_sP=k5)
  e1.record(); t.cuda.synchronize()
  ray_burst_s = e0.elapsed_time(e1) / 20 * 1e-3
  # ... and the scatter of the backward pass (from the saved index tensor the training-mode gather leaves), same form
  gu6 = plan.dec[6]["gu"]
  bwd_args = (gu6[:, gu6.shape[1] - k5:], gu6.stride(0), B, k5, 64, 64, 64, plan.ray_idx[5], plan.gsmap[5], plan.gsmap[5].stride(0),
              64, 64, False) if plan.ray_idx is not None else None
  ray_bwd_s = None
  if bwd_args is not None:
    be.ray_sample_fwd_idx(*ray_args, plan.ray_idx[5], map_sC=1, map_sP=k5)
    gu6.normal_()
    for _ in range(3):
      be.ray_sample_bwd_idx(*bwd_args)
    e0.record()
    for _ in range(20):
      be.ray_sample_bwd_idx(*bwd_args)
    e1.record(); t.cuda.synchronize()
    ray_bwd_s = e0.elapsed_time(e1) / 20 * 1e-3
  # forward-only (eval mode, running BatchRenorm statistics): SURVEY 8(d) asks for it next to the train step
  model.eval()
  with t.no_grad():
    for _ in range(2):
      model(image, v2s, off)
    t.cuda.synchronize()
    e0.record()
    for _ in range(10):
      model(image, v2s, off)
    e1.record(); t.cuda.synchronize()
  eval_s = e0.elapsed_time(e1) / 10 * 1e-3
  model.train()
  if args.math == "bf16x3":
    dtype_note = ("bf16x3 (the convolutions of decoder stages 3-6 and the encoder's 3x3 convolutions -- forward of stages 4-5, data "
                  "gradient of all -- multiply operands split into two bf16 terms: three bf16 MFMAs per product, fp32 "
                  "accumulation, 16 mantissa bits, ~3e-6 relative per layer; tensors, everything else and the fp32_math leg: f32)")
    conv_kernel_name = ("conv_bf3_half_kernel<1,1,7,1,slabs> (stage_6.c1 Conv3d 28->16 k5 @64^3, fwd, split-bf16 MFMA engine; the last "
                        "chunk of 4 channels multiplies 8 taps x 4 channels per MFMA)")
    conv_peak = PEAK_BF16_MFMA / 3
    conv_peak_note = "dense bf16 MFMA peak 2500 TFLOP/s / 3 MFMAs per fp32-equivalent product; achieved counts the layer's real 2*M*K*N"
  else:
    dtype_note = "f32"
    conv_kernel_name = "conv_fwd_kernel<8,1,xvec> (stage_6.c1 Conv3d 28->16 k5 @64^3, fwd)"
    conv_peak, conv_peak_note = PEAK_F32_MFMA, "fp32 matrix peak (v_mfma_f32_16x16x4_f32)"
  out = {
      "metric": "voxels/sec fwd+bwd @128^3", "value": world * B * 128 ** 3 * args.steps / dt,
      "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
      "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
      "vs_baseline": None, "dtype": dtype_note, "data": "synthetic",
      "config": {"workload": f"{'h7' if C == 2 else 'm7/m9'}: CoReNet train step (fwd+{loss_name}+bwd+allreduce+Adam), "
                             f"256x256 RGB -> 128^3, C={C}, B={B}/GPU, decoder_math={args.math}, random-init weights",
                 "global_batch": world * B, "parallelism": f"dp{world}"},
      "loss": float(loss),
      "launch": ("one captured HIP graph per step (CoreNet.train_step); kernel probes from 3 launch-by-launch steps after "
                 "the timed region" if world == 1 and GRAPH else "launch by launch (kernel probes inside the timed region)"),
      "eval_forward": {"ms_per_batch": eval_s * 1e3, "value": B * 128 ** 3 / eval_s, "unit": "voxels/s",
                       "note": "rank 0, forward only, eval mode, same inputs"},
      "roofline": {"kernel": conv_kernel_name,
                   "bound": "mfma", "achieved": CONV6_FLOP * B / conv_s / 1e12, "peak": conv_peak / 1e12,
                   "peak_note": conv_peak_note,
                   "unit": "TFLOP/s", "frac": CONV6_FLOP * B / conv_s / conv_peak,
                   "traffic": traffic.get("conv"),
                   "traffic_source": f"{traffic_file} (separate rocprofv3 --pmc passes of this command; not measured in this run)",
                   "avg_launch_ms": conv_s * 1e3},
      # duration = burst of 20 launches on the step's own buffers (12.2 us; rocprofv3 kernel-trace of the in-step
      # launches: 12.6-13.4 us).  The in-step probe brackets ONE launch with a HIP-event pair, which also times
      # ~5 us of marker latency (rocprof shows a 7 us gap in front of the probed launches) -- reported beside it.
      "roofline_ray_sample": {"kernel": "ray_sample_fwd_kernel<4,channel-last map> (64^3 x 12 ch)", "bound": "hbm",
                              "achieved": RAY64_BYTES * B / ray_burst_s / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                              "frac": RAY64_BYTES * B / ray_burst_s / PEAK_HBM, "traffic": traffic.get("ray"),
                              "avg_launch_ms": ray_burst_s * 1e3, "in_step_event_pair_ms": ray_s * 1e3},
      "roofline_fill_voxels": {"kernel": f"fill_fused_kernel<float,2> ({3 * B} x 128^3 shells, whole call)", "bound": "hbm",
                               "achieved": fill_bytes / fill_s / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                               "frac": fill_bytes / fill_s / PEAK_HBM, "traffic": traffic.get("fill"),
                               "avg_launch_ms": fill_s * 1e3},
  }
  if ray_bwd_s is not None:
    out["roofline_ray_sample_bwd"] = {
        "kernel": "ray_scatter_kernel<4,8,u16> (64^3 x 12 ch gradient -> 64 x 64 map, from the saved index tensor; 64-bit fixed-point LDS window)",
        "bound": "hbm", "achieved": (RAY64_BYTES + RAY64_IDX_BYTES) * B / ray_bwd_s / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
        "frac": (RAY64_BYTES + RAY64_IDX_BYTES) * B / ray_bwd_s / PEAK_HBM, "traffic": traffic.get("ray_bwd"),
        "avg_launch_ms": ray_bwd_s * 1e3, "note": "burst of 20 launches on the step's own buffers (the map gradient is zeroed by the caller)"}
  out["roofline_voxelize"] = {"kernel": f"voxelize_kernel ({3 * B} meshes x {vox_tris.shape[0] // (3 * B)} triangles -> 128^3, multiplier 8, whole call "
                                        "incl. the zero-initialisation of the grids)", "bound": "hbm",
                              "achieved": vox_bytes / vox_s / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                              "frac": vox_bytes / vox_s / PEAK_HBM, "traffic": None, "avg_launch_ms": vox_s * 1e3,
                              "note": "latency-bound rasterisation (one wavefront per triangle); off the model's critical path"}
  for k in ("roofline_ray_sample", "roofline_ray_sample_bwd", "roofline_fill_voxels"):
    if k in out:
      out[k]["traffic_source"] = out["roofline"]["traffic_source"]
  if fp32_side is not None:
    out["fp32_math"] = fp32_side
    # which of the numbers on this line is at the REFERENCE's arithmetic (fp32 end to end, model/losses.py:33): the
    # headline `value` is the product default (split-bf16 products, 16 mantissa bits, error measured under `parity`), this block
    # is the same step with every convolution on the fp32 MFMA engine, timed in the same run
    out["reference_precision"] = {"dtype": "f32", "value": fp32_side["value"], "unit": "voxels/s",
                                  "ms_per_step": fp32_side["ms_per_step"], "roofline": fp32_side["roofline"],
                                  "note": "fp32_math leg: every convolution on v_mfma_f32_16x16x4_f32; the headline value is bf16x3"}
  if m9_side is not None:
    out["m7_m9"] = m9_side
  if world > 1:
    out["rccl"] = dict(sync.describe(), backend=dist.get_backend(),
                       buckets_mb=[round((hi - lo) * 4 / 1e6, 1) for _, lo, hi in model.engine.grad_buckets],
                       exposed_ms_per_bucket=[round(v, 4) for v in sync.exposed_ms_per_bucket()],
                       exposed_exchange_ms=probes.get("grad_exchange_wait", 0.0) * 1e3, replicas=replicas)
  progress("kernel legs done")
  if not args.no_cpu_baseline and world == 1:
    out["cpu_baseline"] = cpu_baseline(state0, batch_cpu, loss_name)
    progress("cpu baseline done")
    del model, plan
    out["parity"] = parity_check(state0, batch_cpu, C, dev, args.math)
    progress("parity check done")
    if C == 2 and not args.no_sr_side:
      out["super_resolution_x2"] = super_resolution_leg(state0, batch_cpu, dev, args.math)
      progress("super-resolution leg done")
    out["cpu_baseline_fill_voxels"] = cpu_baseline_fill(shells.cpu())
  print(json.dumps(out))


if __name__ == "__main__":
  main()
