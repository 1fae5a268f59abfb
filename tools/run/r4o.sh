R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/tools/prof_step.py bf16x3 4 > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$f" > $R/gpurun_out/r04_order.log <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
print(sorted(set((r['Kernel_Name'][:60], r.get('Grid_Size'), r.get('Grid_Size_X')) for r in rows if 'ray' in r['Kernel_Name'])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
# every 64^3 ray_sample_bwd launch: what ran on the other queue around its start
for i, r in enumerate(rows):
  if 'ray_sample_bwd' in r['Kernel_Name'] and int(r.get('Grid_Size_X') or 0) == 32768:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"ray_sample_bwd grid {r.get('Grid_Size')} {r.get('Grid_Size_X')} queue {r['Queue_Id']} start {(s - t0) / 1e3:.1f} us end {(e - t0) / 1e3:.1f}")
    # the last bn_bwd_apply / bn kernels before it, any queue
    prev = [q for q in rows[max(0, i - 40):i + 10] if 'bn_bwd' in q['Kernel_Name'] or 'conv_bf3' in q['Kernel_Name']]
    for q in prev[-8:]:
      qs, qe = int(q['Start_Timestamp']), int(q['End_Timestamp'])
      print(f"    {q['Kernel_Name'][28:70]:44s} grid {str(q.get('Grid_Size_X')):>9s} queue {q['Queue_Id']} start {(qs - t0) / 1e3:9.1f} end {(qe - t0) / 1e3:9.1f}  {'OVERLAPS' if qe > s and qs < e else ''}")
PY
