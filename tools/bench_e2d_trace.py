#!/usr/bin/env python
"""GPU-side durations of tools/bench_e2d.py from a rocprofv3 kernel trace (the event timing of bench_e2d.py is bound by
the host's launch rate for kernels this short):
  rocprofv3 --kernel-trace --output-format csv -d /tmp/p -o e -- python tools/bench_e2d.py 20
  python tools/bench_e2d_trace.py /tmp/p/e_kernel_trace.csv 20"""
import csv, sys
sys.path.insert(0, __file__.rsplit("/", 2)[0])
rows = sorted((r for r in csv.DictReader(open(sys.argv[1])) if "conv_e2d_kernel" in r["Kernel_Name"] or "splitk_reduce" in r["Kernel_Name"]),
              key=lambda r: int(r["Start_Timestamp"]))
iters = int(sys.argv[2]) + 5
names = ["s2_3x3", "s3_3x3", "s4_3x3", "s5_3x3", "s2_1x1_64_256", "s2_1x1_256_64", "s3_1x1_512_128", "s3_1x1_128_512",
         "s4_1x1_1024_256", "s4_1x1_256_1024", "s5_1x1_2048_512", "s5_1x1_512_2048"]
i, tot = 0, 0.0
for nm in names:
  k, r, n = 0.0, 0.0, 0
  grid = None
  while n < iters and i < len(rows):
    row = rows[i]; i += 1
    d = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    if "conv_e2d" in row["Kernel_Name"]:
      k += d; n += 1
      grid = (int(row["Grid_Size_X"]) // 256, row["Grid_Size_Y"], row["Grid_Size_Z"], row["LDS_Block_Size"], row["VGPR_Count"])
      if i < len(rows) and "splitk" in rows[i]["Kernel_Name"]:
        r += (int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3; i += 1
  tot += (k + r) / iters
  print(f"{nm:18s} kernel {k / iters:6.1f} us  reduce {r / iters:5.1f} us   grid {grid}")
print(f"sum {tot:.1f} us")
