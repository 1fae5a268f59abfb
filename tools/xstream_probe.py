#!/usr/bin/env python
"""Cross-stream hand-over probe (round 4): stream A writes a tensor, records an event and goes on with a long kernel; stream B
waits for the event and reads the tensor at once.  Counts stale elements.  usage: xstream_probe.py [MB] [iters]"""
import sys
import torch as t
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 58
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = mb * (1 << 20) // 4
side = t.cuda.Stream()
X = t.zeros(n, device="cuda")
big = t.randn(4096, 4096, device="cuda")
bad = t.zeros(iters, dtype=t.int64, device="cuda")
for name, main in (("default stream", t.cuda.current_stream()), ("a non-default stream", t.cuda.Stream())):
  bad.zero_()
  with t.cuda.stream(main):
    for i in range(iters):
      X.fill_(float(i + 1))
      ev = t.cuda.Event(); ev.record()
      for _ in range(2): big @ big                      # the producer stream stays busy
      with t.cuda.stream(side):
        side.wait_event(ev)
        bad[i] = (X != float(i + 1)).sum()
      main.wait_stream(side)                            # (X is rewritten next iteration)
  t.cuda.synchronize()
  nb = bad.cpu()
  print(f"producer on {name}: {int((nb > 0).sum())} of {iters} hand-overs saw stale elements (max {int(nb.max())} of {n})")
