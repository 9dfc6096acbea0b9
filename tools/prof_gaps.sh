#!/bin/bash
# usage: tools/prof_gaps.sh <tag> <python script and args...>
# rocprofv3 kernel trace of the command; per kernel name: calls, average duration, average idle time of the device in
# front of it (start - end of the previous kernel, launches in start order); the raw trace is deleted
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps_$tag -- python3 "$@" > gpurun_out/gaps_$tag.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/gaps_$tag/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
rows = rows[len(rows) // 3:]          # (skip the warm-up third)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
prev_end = None
for s, e, n in rows:
    name = n.replace("ptd::(anonymous namespace)::", "").replace("ptd::", "")[:70]
    a = agg[name]
    a[0] += 1
    a[1] += (e - s) / 1e3
    if prev_end is not None:
        a[2] += max(0, s - prev_end) / 1e3
    prev_end = max(prev_end or 0, e)
span = (rows[-1][1] - rows[0][0]) / 1e3
print("span %.1f us, kernels %d" % (span, len(rows)))
for name, (n, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s calls %5d  avg %8.2f us  idle before %6.2f us" % (name, n, d / n, g / n))
PY
rm -f gpurun_out/gaps_$tag/*/*kernel_trace.csv
