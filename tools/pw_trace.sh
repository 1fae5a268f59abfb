# GPU-side durations (rocprofv3 kernel trace) of the encoder's 1x1 layers, both pointwise kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  rm -rf /tmp/pwt$v
  for l in e2a0 e2c e2a e3c e3a e4c e4a e5c e5a; do for m in fwd dgrad; do
    CRN_PW2=$v rocprofv3 --kernel-trace --output-format csv -d /tmp/pwt$v -o ${l}_${m} -- python $R/tools/bench_conv.py $m $l 20 4 fp32 > /dev/null 2>&1
  done; done
done
python - <<'PY'
import csv, glob, os, collections
for v in (0, 1):
  print("CRN_PW2 =", v)
  for f in sorted(glob.glob(f"/tmp/pwt{v}/**/*_kernel_trace.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
      n = r["Kernel_Name"]
      if "pointwise" in n or "pw2" in n or "splitk" in n:
        agg[n.split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tag = os.path.basename(f).replace("_kernel_trace.csv", "")
    print(f"  {tag:12s} " + "  ".join(f"{k}: n={len(x)} med {sorted(x)[len(x)//2]:.1f} us" for k, x in agg.items()))
PY
