"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database (view counters_collection).
usage: python tools/rocpd_pmc.py results.db [kernel-substring]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
need = {"kernel": None, "counter": None, "value": None, "dispatch": None}
for c in cols:
    lc = c.lower()
    if lc in ("kernel_name", "name") and need["kernel"] is None: need["kernel"] = c
    if lc in ("counter_name",): need["counter"] = c
    if lc in ("value", "counter_value"): need["value"] = c
    if lc in ("dispatch_id",): need["dispatch"] = c
if None in (need["kernel"], need["counter"], need["value"]):
    print("columns:", cols)
    sys.exit(1)
q = "select %s, %s, %s, %s from counters_collection" % (need["kernel"], need["counter"], need["value"], need["dispatch"] or "0")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for k, c, v, d in cur.execute(q):
    if pat and pat not in k:
        continue
    acc[k][c] += v
    disp[k].add(d)
for k in acc:
    n = max(len(disp[k]), 1)
    print("%s  (dispatches %d)" % (k[:100], n))
    for c in sorted(acc[k]):
        print("    %-32s %16.1f per dispatch" % (c, acc[k][c] / n))
