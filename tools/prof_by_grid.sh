#!/bin/bash
# usage: tools/prof_by_grid.sh <tag> <kernel-name substring> <python script and args...>
# rocprofv3 kernel trace of the command, then the launches whose name contains the substring grouped by grid size:
# calls, total and average duration (the raw trace is deleted: hundreds of MB)
tag=$1; pat=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/grid_$tag -- python3 "$@" > gpurun_out/grid_$tag.log 2>&1
python3 - "$pat" <<PY
import csv, glob, sys, collections
pat = sys.argv[1]
f = glob.glob("gpurun_out/grid_$tag/*/*kernel_trace.csv")[0]
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
for r in csv.DictReader(open(f)):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    if pat in r["Kernel_Name"]:
        name = r["Kernel_Name"].replace("ptd::(anonymous namespace)::", "").replace("ptd::", "")[:60]
        key = (name, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r.get("LDS_Block_Size", 0) or 0))
        agg[key][0] += 1
        agg[key][1] += d
print("total kernel time %.1f ms" % (tot / 1e3))
for key, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-62s wgs %6d y %3d lds %6d  calls %6d  total %9.1f us  avg %8.2f us" % (key[0], key[1], key[2], key[3], n, d, d / n))
PY
rm -f gpurun_out/grid_$tag/*/*kernel_trace.csv
