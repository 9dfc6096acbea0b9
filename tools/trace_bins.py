"""Bin the per-dispatch durations of a rocprofv3 --kernel-trace csv by kernel and launch order.
Usage: python tools/trace_bins.py <kernel_trace.csv> [bins]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
by = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"].replace("ptd::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    by[nm].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
print("span %.2f ms" % ((t1 - t0) / 1e6))
for nm, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
    v.sort()
    tot = sum(e - s for s, e in v)
    line = "%-28s n=%5d total %7.2f ms avg %6.2f us" % (nm[:28], len(v), tot / 1e6, tot / len(v) / 1e3)
    if len(v) >= 4 * nb:
        step = len(v) // nb
        line += "  bins: " + " ".join("%.1f" % (sum(e - s for s, e in v[k * step:(k + 1) * step]) / step / 1e3) for k in range(nb))
    print(line)
