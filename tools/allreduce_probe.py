#!/usr/bin/env python
"""Bare all-reduce time of the engine's gradient buckets (and of the whole 144.6 MB slab) over RCCL, one process per
GPU.  Run under torchrun on a multi-GPU node (tools/scale_probe.sh does, once per NCCL_ALGO / NCCL_PROTO setting):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/allreduce_probe.py
Prints one JSON line from rank 0: per-bucket microseconds (median of 20) for the torch.distributed transport and for
the library's own communicator (crn_allreduce_f32), next to the xGMI estimate of SURVEY section 5."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as t
import torch.distributed as dist
from corenet_amd import distributed as D
from corenet_amd.model.engine import param_specs, ParamStore, GRAD_BUCKET_LABELS

rank, local, world = D.init_from_env()
DRY = os.environ.get("CRN_DIST_BACKEND") == "gloo"      # tools/scale_probe.sh dry run: the ranks share one GPU, no RCCL
if DRY:
  local = local % t.cuda.device_count()
t.cuda.set_device(local)
specs = param_specs(2)
off, n = {}, 0
for key, shape, kind in specs:
  if kind == "param":
    m = 1
    for d in shape: m *= d
    off[key] = n; n += (m + 3) // 4 * 4
los = [min(o for k, o in off.items() if k.startswith(lb)) for lb in GRAD_BUCKET_LABELS[:-1]] + [0]
sizes, hi = [], n
for lo in los:
  sizes.append(hi - lo); hi = lo
sizes.append(n)                                     # the whole slab in one piece
buf = t.randn(n, device="cuda")
native = D.NativeComm() if (world > 1 and not DRY) or os.environ.get("CRN_PROBE_NATIVE") else None

def timeit(fn, x):
  for _ in range(3): fn(x)
  t.cuda.synchronize()
  ts = []
  for _ in range(20):
    a, b = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    if world > 1: dist.barrier()
    a.record(); fn(x); b.record(); t.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
  return sorted(ts)[len(ts) // 2]

res = {"ranks": world, "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO"),
       "bucket_mb": [round(s * 4 / 1e6, 1) for s in sizes], "torch_us": [], "native_us": []}
for s in sizes:
  x = buf[:s]
  if world > 1:
    res["torch_us"].append(round(timeit(lambda v: dist.all_reduce(v), x), 1))
  if native is not None:
    res["native_us"].append(round(timeit(native.all_reduce, x), 1))
# ring over one xGMI link (153 GB/s) vs direct reduce-scatter + all-gather over the 7 links of a GPU
res["estimate_us"] = {"ring_one_link": [round(2 * (world - 1) / max(world, 1) * s * 4 / 153e9 * 1e6, 1) for s in sizes],
                      "direct_seven_links": [round(2 * (world - 1) / max(world, 1) * s * 4 / (7 * 153e9) * 1e6, 1) for s in sizes]}
if rank == 0:
  print(json.dumps(res))
if world > 1:
  dist.barrier(); dist.destroy_process_group()
