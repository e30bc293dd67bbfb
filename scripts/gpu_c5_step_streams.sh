#!/bin/bash
# Where a distinct-batch C5 step's GPU time goes BY QUEUE: model kernels on the main stream, batch assembly / pair tables / deferred
# weight gradients on the side streams, and the idle gaps of the main stream (what the fixed-batch figure does not see).
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5_streams; rm -rf $O; mkdir -p $O; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o c5 -- \
    python $R/bench.py --config C5 --steps 10 --warmup 4 --no-roofline --no-extras --no-cpu-baseline > $O/bench_c5.json 2> $O/err.txt
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/c5_streams"
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:70]
idx = [i for i, r in enumerate(rows[:-1]) if "mt_adam_clip" in r["Kernel_Name"] and "mt_" not in rows[i + 1]["Kernel_Name"]]
a, b = idx[5], idx[-1]                       # steady state: after the 6th optimizer step up to the last
steps = len(idx) - 1 - 5
t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
sel = rows[a + 1:b + 1]
print("steps", steps, "span per step ms", (t1 - t0) / steps / 1e6)
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
main = max(byq, key=lambda q: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in byq[q]))
out = []
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    out.append("queue %s%s: %d kernels, busy %.3f ms per step" % (q, " (main)" if q == main else "", len(rs), busy / steps / 1e6))
    agg = collections.Counter()
    cnt = collections.Counter()
    for r in rs:
        agg[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[short(r["Kernel_Name"])] += 1
    for n, v in agg.most_common(14 if q != main else 8):
        out.append("    %8.3f ms/step  %6.1f calls/step  %s" % (v / steps / 1e6, cnt[n] / steps, n))
# idle gaps of the main queue
rs = byq[main]
gaps = []
for p, c in zip(rs[:-1], rs[1:]):
    g = int(c["Start_Timestamp"]) - int(p["End_Timestamp"])
    if g > 20000:
        gaps.append((g, short(p["Kernel_Name"]), short(c["Kernel_Name"])))
out.append("main queue: idle in gaps > 20 us: %.3f ms per step (%d gaps); all gaps %.3f ms per step" % (
    sum(g for g, _, _ in gaps) / steps / 1e6, len(gaps),
    sum(max(0, int(c["Start_Timestamp"]) - int(p["End_Timestamp"])) for p, c in zip(rs[:-1], rs[1:])) / steps / 1e6))
agg = collections.Counter()
for g, p, c in gaps:
    agg[(p, c)] += g
for (p, c), v in agg.most_common(12):
    out.append("    %8.3f ms/step between  %s  ->  %s" % (v / steps / 1e6, p, c))
big = max(range(len(rs) - 1), key=lambda i: int(rs[i + 1]["Start_Timestamp"]) - int(rs[i]["End_Timestamp"]))
base = int(rs[big]["End_Timestamp"])
out.append("around the largest gap of the main queue (us relative to its start, duration, queue, kernel):")
allq = sorted(sel, key=lambda r: int(r["Start_Timestamp"]))
for r in allq:
    s = int(r["Start_Timestamp"]) - base
    if -300e3 < s < 9000e3 and (s < 100e3 or s > 7000e3):
        out.append("   %10.1f %8.1f q%s %s" % (s / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"], short(r["Kernel_Name"])))
open(O + "/streams.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find $O -name "*kernel_trace.csv" -delete; rm -rf $O/trace
cut -c1-400 $O/bench_c5.json
