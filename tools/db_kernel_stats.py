#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 results database (rocpd sqlite): python tools/db_kernel_stats.py x_results.db [first-kernel-substring]
Only dispatches from the first launch of the named kernel on (skips set-up) when a substring is given."""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
def key(n):
    n = re.sub(r'^void ', '', n).replace('(anonymous namespace)::', '')
    return n.split('(')[0][:70]
st = 0
if len(sys.argv) > 2:
    hits = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    st = hits[skip] if len(hits) > skip else 0
leg = rows[st:]
tot = collections.defaultdict(lambda: [0, 0])
for n, s, e in leg:
    k = key(n); tot[k][0] += 1; tot[k][1] += e - s
span = leg[-1][2] - leg[0][1]
busy = sum(v[1] for v in tot.values())
print("span %.2f ms, busy %.2f ms, %d dispatches" % (span / 1e6, busy / 1e6, len(leg)))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 30]:
    print("%-70s %6d %9.3f ms  %8.2f us" % (k, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))
