#!/usr/bin/env python
"""Kernel sequence of ONE steady-state training step out of a rocprofv3 kernel trace (csv): per queue the busy time and launch count
per step, and the launches of the last full step in start order (queue, start offset, duration, name) — what runs next to what.
Steps are delimited by the optimizer's last kernel (mt_adam_clip_kernel).   usage: step_sequence.py <kernel_trace.csv> [max_rows]"""
import collections, csv, sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")
    return n.split("(")[0][:84]


rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ends = [i for i, r in enumerate(rows[:-1]) if "mt_adam_clip" in r["Kernel_Name"] and "mt_" not in rows[i + 1]["Kernel_Name"]]
if len(ends) < 3:
    raise SystemExit("fewer than 3 optimizer steps in the trace")
a, b = ends[-2], ends[-1]
sel = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
print("last full step: %.3f ms, %d launches" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e6, len(sel)))
steps = len(ends) - 1
allsel = rows[ends[0] + 1:ends[-1] + 1]
byq = collections.defaultdict(list)
for r in allsel:
    byq[r["Queue_Id"]].append(r)
qname = {}
for i, (q, rs) in enumerate(sorted(byq.items(), key=lambda kv: -sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kv[1]))):
    qname[q] = "Q%d" % i
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print("%s (queue id %s): %.1f launches / step, busy %.3f ms / step" % (qname[q], q, len(rs) / steps, busy / steps / 1e6))
    agg, cnt = collections.Counter(), collections.Counter()
    for r in rs:
        agg[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[short(r["Kernel_Name"])] += 1
    for n, v in agg.most_common(40):
        print("    %8.3f ms/step %6.1f x %8.1f us  %s" % (v / steps / 1e6, cnt[n] / steps, v / cnt[n] / 1e3, n))
print("sequence of the last step (queue, start us, duration us, name):")
prev_end = {}
for r in sel[:limit]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = qname.get(r["Queue_Id"], r["Queue_Id"])
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    print("  %s %9.1f %8.1f %s%s" % (q, (s - t0) / 1e3, (e - s) / 1e3, short(r["Kernel_Name"]), "   [gap %.0f]" % gap if gap > 15 else ""))
