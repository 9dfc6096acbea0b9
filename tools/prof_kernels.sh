#!/bin/bash
# usage: tools/prof_kernels.sh <tag> <python script and args...>   -- rocprofv3 kernel stats, top kernels printed
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$tag/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms" % (tot / 1e6))
for r in rows[:28]:
    print("%-70s %6s calls %9.1f us total %8.2f us avg %5.1f %%" % (r["Name"][:70].replace("ptd::(anonymous namespace)::","").replace("ptd::",""), r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
# (the raw trace is hundreds of MB: gpurun copies back at most 64 MiB)
rm -f gpurun_out/prof_$tag/*/*kernel_trace.csv
