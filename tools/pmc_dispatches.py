#!/usr/bin/env python3
"""Per-dispatch counter values from a rocprofv3 results .db: for every (kernel, counter) the dispatches in launch order,
summed over the FIRST and the SECOND half of them - bench.py runs its headline step (sparse set) before the dense leg in
the same process, so the second half of a kernel's dispatches are the dense leg's (tools/pmc_dense_r06.sh).
usage: pmc_dispatches.py results.db kernel-substring [kernel-substring ...]"""
import sqlite3
import sys

db, pats = sys.argv[1], sys.argv[2:]
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
order = "dispatch_id" if "dispatch_id" in cols else "rowid"
dur = None
for a, b in (("start", "end"), ("start_timestamp", "end_timestamp")):
    if a in cols and b in cols:
        dur = (a, b)
print("columns:", ",".join(cols))
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
# per-dispatch durations from the kernel trace, when the database has a view of it
for tname in ("kernels", "rocpd_kernel_dispatch"):
    if tname in tabs:
        kc = [r[1] for r in cur.execute(f"pragma table_info({tname})")]
        nm = next((c for c in ("name", "kernel_name") if c in kc), None)
        st_, en_ = next((c for c in ("start", "start_timestamp") if c in kc), None), next((c for c in ("end", "end_timestamp") if c in kc), None)
        if nm and st_ and en_:
            wk = " or ".join(f"{nm} like '%%{p}%%'" for p in pats) or "1"
            kd = {}
            for n_, a_, b_ in cur.execute(f"select {nm}, {st_}, {en_} from {tname} where {wk} order by {st_}"):
                kd.setdefault(n_.split("(")[0].replace("void ", "").replace("amc::", ""), []).append((b_ - a_) * 1e-6)
            for n_, v_ in sorted(kd.items()):
                h_ = len(v_) // 2
                print(f"{n_:34s} {'duration_ms':24s} dispatches={len(v_):3d} first_half_sum={sum(v_[:h_]):.4f} second_half_sum={sum(v_[h_:]):.4f}")
        break
where = " or ".join("kernel_name like '%%%s%%'" % p for p in pats) or "1"
rows = list(cur.execute(f"select kernel_name, counter_name, {order}, value" + (f", {dur[0]}, {dur[1]}" if dur else "") +
                        f" from counters_collection where {where} order by {order}"))
by = {}
for r in rows:
    name = r[0].split("(")[0].replace("void ", "").replace("amc::", "")
    by.setdefault((name, r[1]), []).append(r[2:])
for (name, counter), v in sorted(by.items()):
    # several rows of one dispatch (one per instance) are summed
    per = {}
    t = {}
    for x in v:
        per[x[0]] = per.get(x[0], 0.0) + float(x[1])
        if dur:
            t[x[0]] = (x[3] - x[2]) * 1e-6
    ids = sorted(per)
    h = len(ids) // 2
    a, b = sum(per[i] for i in ids[:h]), sum(per[i] for i in ids[h:])
    line = f"{name:34s} {counter:24s} dispatches={len(ids):3d} first_half_sum={a:.6g} second_half_sum={b:.6g}"
    if dur:
        line += f" first_half_ms={sum(t[i] for i in ids[:h]):.3f} second_half_ms={sum(t[i] for i in ids[h:]):.3f}"
    print(line)
