"""Print the kernel/copy timeline of a rocprofv3 --kernel-trace (--memory-copy-trace) results .db between two
extractions: python tools/timeline.py results.db [which_refit_from_end]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cur = db.cursor()
rows = [(n, s, e, g, "k") for n, s, e, g in cur.execute("select name,start,end,grid_x from kernels")]
try:
    rows += [(n, s, e, sz, "c") for n, s, e, sz in cur.execute("select name,start,end,size from memory_copies")]
except sqlite3.Error:
    pass
rows.sort(key=lambda r: r[1])


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
    return n[:44]


ref = [i for i, r in enumerate(rows) if "refit_mask" in r[0]]
i0, i1 = ref[-back] - 2, ref[-back + 1] + 1
t0, prev = rows[i0][1], None
for r in rows[i0:i1]:
    gap = (r[1] - prev) / 1e3 if prev else 0.0
    print("%9.1f  gap %6.1f  dur %7.1f  %s %s=%d" % ((r[1] - t0) / 1e3, gap, (r[2] - r[1]) / 1e3, short(r[0]), "grid" if r[4] == "k" else "bytes", r[3]))
    prev = max(prev or 0, r[2])
