#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02v; rm -rf $O; mkdir -p $O; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- \
    python $R/bench.py --steps 40 --warmup 10 --no-roofline --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/err.txt
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r02v"
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 30% of the trace = steady state; compute busy time (union of intervals) and per-stream sums
n = len(rows); sub = rows[int(n * 0.5):]
t0, t1 = int(sub[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sub)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sub)
busy = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
tot = sum(e - s for s, e in iv)
print("window %.3f ms, GPU busy (union) %.3f ms = %.1f %%, sum of kernel durations %.3f ms" % ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), tot / 1e6))
by = collections.Counter()
for r in sub: by[r.get("Stream_Id", r.get("Queue_Id", "?"))] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("per queue/stream ms:", {k: round(v / 1e6, 3) for k, v in by.items()})
# biggest idle gaps
gaps = []
cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: gaps.append((s - ce, ce)); 
    ce = max(ce, e)
ends = {}
for r in sub: ends[int(r["End_Timestamp"])] = r
starts = sorted((int(r["Start_Timestamp"]), r) for r in sub)
import bisect
gaps.sort(reverse=True)
print("total gap ms %.3f over %d gaps > 0; gaps > 20us: %d" % (sum(g for g, _ in gaps) / 1e6, len(gaps), sum(1 for g, _ in gaps if g > 20000)))
hist = collections.Counter()
for g, ce in gaps:
    if g < 20000: continue
    before = ends.get(ce, {}).get("Kernel_Name", "?")[:60]
    i = bisect.bisect_left(starts, (ce + g, {})) if False else next(k for k, (s_, _) in enumerate(starts) if s_ >= ce + g)
    after = starts[i][1]["Kernel_Name"][:60]
    hist[(before, after)] += g
for (b, a), g in hist.most_common(12):
    print("%8.1f us total | after %-60s | before %-60s" % (g / 1e3, b, a))
PY
find $O -name "*kernel_trace.csv" -delete
cat $O/bench_under_rocprof.json | cut -c1-300
