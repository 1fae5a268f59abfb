R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for cfg in "1 0" "0 0" "1 2" "1 1"; do set -- $cfg
  rm -rf /tmp/pr; CRN_RAY_TX4=8 CRN_RAY_BWD2=$1 CRN_RAY_DBG=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o r -- python $R/tools/bench_small.py > /dev/null 2>&1
  echo "== BWD2=$1 DBG=$2"; python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/pr/r_kernel_stats.csv')):
  n = r['Name']
  if 'ray_sample_bwd' in n or 'fillBuffer' in n or 'memset' in n.lower():
    print(f"  {int(r['Calls']):5d} calls avg {float(r['AverageNs'])/1e3:7.2f} us min {float(r['MinNs'])/1e3:7.2f} max {float(r['MaxNs'])/1e3:7.2f}  {n[:80]}")
PY
done > $O/r04_ray_prof.log 2>&1
python - <<'PY' >> $O/r04_ray_prof.log
import csv, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/pr/r_kernel_trace.csv')):
  if 'ray_sample_bwd' in r['Kernel_Name']:
    d[(r['Kernel_Name'][:60], r['Grid_Size'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items(): print(k, len(v), 'avg %.2f us' % (sum(v) / len(v)))
PY
