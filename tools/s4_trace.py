#!/usr/bin/env python3
"""Reads the time stamps a debug launch of the v4 score kernel leaves (RH_S4_TRACE=<n-th launch>, RH_S4_TRACE_FILE):
16 shader-clock stamps per wave.  Prints the distribution of every phase.  python tools/s4_trace.py file [waves_per_block]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
a = a[a[:, 0] != 0]
t0 = a[:, 0].min()
print("waves", len(a), "first start .. last end (cycles):", int((a.max() - t0)))
names = ["start"]
k = (a != 0).sum(axis=1).max()
for i in range(1, k):
    d = a[:, i] - a[:, i - 1]
    ok = (a[:, i] != 0) & (a[:, i - 1] != 0)
    d = d[ok]
    print("stamp %2d - %2d: n %6d  mean %8.0f  p50 %8.0f  p90 %8.0f  max %8.0f   | stamp at: mean %8.0f max %8.0f" %
          (i, i - 1, len(d), d.mean(), np.median(d), np.percentile(d, 90), d.max(), (a[ok, i] - t0).mean(), (a[ok, i] - t0).max()))
print("start spread: p50 %d p90 %d max %d" % tuple(np.percentile(a[:, 0] - t0, [50, 90, 100])))
