"""Per-kernel busy time of the LAST forward in a rocprofv3 results .db (python tools/prof_breakdown.py <db> [n_chunks])."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
idx = [i for i, r in enumerate(rows) if "k_x_to_bf16" in r[0]]
seg = rows[idx[-n_chunks]:]
d = defaultdict(lambda: [0, 0])
for n, s, e in seg:
    d[n[:70]][0] += 1
    d[n[:70]][1] += e - s
print("launches", len(seg), "busy ms", sum(e - s for _, s, e in seg) / 1e6, "span ms", (seg[-1][2] - seg[0][1]) / 1e6)
for k, v in d.items():
    print(f"{v[1] / 1e6:8.3f} ms  x{v[0]:4d}  {k}")
