# GPU-side durations (rocprofv3 kernel trace) of the encoder's 1x1 layers through crn_conv_fwd.  -> gpurun_out/pw_ab.txt
# usage: pw_ab.sh ["ENV=val ENV=val" ...]   one column per environment setting (default: one column, no setting)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
[ $# -eq 0 ] && set -- "X=0"
i=0
for v in "$@"; do
  rm -rf /tmp/pwt$i
  for l in e2a0 e2c e2a e3c e3a e4c e4a e5c e5a; do for m in fwd dgrad; do
    env $v timeout 120 rocprofv3 --kernel-trace --output-format csv -d /tmp/pwt$i -o ${l}_${m} -- python $R/tools/bench_conv.py $m $l 20 4 fp32 > /dev/null 2>&1
  done; done
  i=$((i+1))
done
python - "$@" <<'PY' | tee $R/gpurun_out/pw_ab.txt
import csv, glob, os, sys, collections
res = collections.defaultdict(dict)
for i, v in enumerate(sys.argv[1:]):
  for f in sorted(glob.glob(f"/tmp/pwt{i}/**/*_kernel_trace.csv", recursive=True)):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
      n = r["Kernel_Name"]
      if "pointwise" in n or "splitk" in n:
        agg["pw" if "pointwise" in n else "red"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    med = lambda x: sorted(x)[len(x) // 2] if x else 0.0
    res[os.path.basename(f).replace("_kernel_trace.csv", "")][i] = (med(agg["pw"]), med(agg["red"]))
print("layer         " + "".join(f"{v:>22s}" for v in sys.argv[1:]) + "   (kernel us + split-K reduction us, medians of 23 launches, B=4)")
tot = collections.defaultdict(float)
for tag, d in res.items():
  print(f"{tag:14s}" + "".join(f"{d[i][0]:15.1f} +{d[i][1]:5.1f}" if i in d else " " * 22 for i in range(len(sys.argv) - 1)))
  for i in d: tot[i] += d[i][0] + d[i][1]
print(f"{'sum':14s}" + "".join(f"{tot[i]:22.1f}" for i in range(len(sys.argv) - 1)))
PY
