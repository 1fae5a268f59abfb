#!/bin/bash
# First contact with a multi-GPU node: everything the 1 -> 8 scaling question needs, in one run.
#   bash tools/scale_probe.sh [out_dir]          (from the repo root; needs N >= 2 GPUs, uses 1 / 2 / 4 / 8 of them)
# Writes   <out>/allreduce_<algo>.json   bare all-reduce time of every gradient bucket (+ the whole slab) per NCCL_ALGO /
#                                        NCCL_PROTO setting, torch.distributed and the library's own communicator
#          <out>/bench_n<N>.json         the bench line at N = 1, 2, 4, 8 (rccl.exposed_ms_per_bucket, exposed_exchange_ms)
#          <out>/bench_n<max>_native.json, ..._oneslab.json   the same step with crn_allreduce_f32 / without overlap
#          <out>/summary.txt             voxels/s and scaling efficiency per N, exposed exchange per step
# SCALE_PROBE_DRY=1: dry run of this script on a ONE-GPU box (tests/test_dist_gpu.py): two ranks share the GPU over gloo
# (CRN_DIST_BACKEND=gloo), 2 steps, no side legs, no RCCL -- the commands, files and the summary format are what is checked.
set -u
OUT=${1:-gpurun_out/scale_probe}; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
DRY=${SCALE_PROBE_DRY:-0}
NG=$(python -c "import torch; print(torch.cuda.device_count())")
STEPS="--steps 20 --warmup 5"
if [ "$DRY" = "1" ]; then export CRN_DIST_BACKEND=gloo; NG=2; STEPS="--steps 2 --warmup 1 --no-fp32-side --no-m9-side"; fi
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29700 + RANDOM % 200)) "${@:2}"; }
MAXN=1; for n in 2 4 8; do [ $NG -ge $n ] && MAXN=$n; done
echo "GPUs visible: $NG, probing up to $MAXN ranks" | tee $OUT/summary.txt
if [ $MAXN -ge 2 ]; then
  CFGS="default:: ring:Ring: tree:Tree: ring_simple:Ring:Simple ring_ll128:Ring:LL128"; [ "$DRY" = "1" ] && CFGS="default::"
  for cfg in $CFGS; do
    IFS=: read name algo proto <<< "$cfg"
    ( [ -n "$algo" ] && export NCCL_ALGO=$algo; [ -n "$proto" ] && export NCCL_PROTO=$proto
      run $MAXN tools/allreduce_probe.py > $OUT/allreduce_$name.json 2> $OUT/allreduce_$name.err )
  done
fi
python bench.py --gpus 1 $STEPS --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err
for n in 2 4 8; do
  [ $NG -ge $n ] || continue
  run $n bench.py --gpus $n $STEPS > $OUT/bench_n$n.json 2> $OUT/bench_n$n.err
done
if [ $MAXN -ge 2 ]; then
  [ "$DRY" = "1" ] || CRN_NATIVE_RCCL=1 run $MAXN bench.py --gpus $MAXN $STEPS > $OUT/bench_n${MAXN}_native.json 2> $OUT/bench_native.err
  CRN_OVERLAP_ALLREDUCE=0 run $MAXN bench.py --gpus $MAXN $STEPS > $OUT/bench_n${MAXN}_oneslab.json 2> $OUT/bench_oneslab.err
fi
python - $OUT <<'PY' | tee -a $OUT/summary.txt
import glob, json, os, sys
out = sys.argv[1]
base = None
for f in sorted(glob.glob(os.path.join(out, "bench_n*.json")), key=lambda p: (len(p), p)):
  try: d = json.loads([l for l in open(f) if l.startswith("{")][-1])
  except Exception as e: print(os.path.basename(f), "no line:", e); continue
  n = d["n_gpus"]
  if n == 1: base = d["value"]
  eff = f"{d['value'] / (base * n):.3f}" if base else "n/a"
  r = d.get("rccl", {})
  print(f"{os.path.basename(f):28s} n={n} {d['ms_per_step']:.3f} ms/step  {d['value'] / 1e9:.3f} G voxels/s  efficiency {eff}  "
        f"transport {r.get('transport')}  exposed exchange {r.get('exposed_exchange_ms')} ms  per bucket {r.get('exposed_ms_per_bucket')}")
for f in sorted(glob.glob(os.path.join(out, "allreduce_*.json"))):
  try: d = json.loads(open(f).read().strip().splitlines()[-1])
  except Exception as e: print(os.path.basename(f), "no line:", e); continue
  print(f"{os.path.basename(f):28s} buckets MB {d['bucket_mb']}  torch us {d['torch_us']}  native us {d['native_us']}")
PY
