#!/bin/bash
# usage: tools/prof_ts.sh <tag> [n]   -- rocprofv3 kernel stats of tools/twostage_check.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python3 tools/twostage_check.py ${2:-4096} > gpurun_out/prof_$1.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$1/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:64].replace("ptd::(anonymous namespace)::",""), r["Calls"], int(float(r["TotalDurationNs"]))//1000, "us total", float(r["AverageNs"])/1000, "us avg")
PY
