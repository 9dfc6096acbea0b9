"""Timeline of the last `window_ms` of a rocprofv3 --kernel-trace csv: consecutive launches of the same
kernel are merged into one line (count, busy time, span), gaps between groups are printed.
Usage: python tools/trace_timeline.py <kernel_trace.csv> [window_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].replace("ptd::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44],
             r.get("Grid_Size") or (r["Grid_Size_X"] + "x" + r["Grid_Size_Y"]), r.get("Workgroup_Size") or r["Workgroup_Size_X"]) for r in rows)
tend = ev[-1][1]
ev = [e for e in ev if e[0] >= tend - win * 1e6]
t0 = ev[0][0]
groups = []
for s, e, nm, g, w in ev:
    if groups and groups[-1]["nm"] == nm and (nm.startswith("sytrd") or nm.startswith("gemm_f64") or groups[-1]["grid"] == g):
        gr = groups[-1]; gr["n"] += 1; gr["busy"] += e - s; gr["end"] = e
    else:
        groups.append({"nm": nm, "n": 1, "busy": e - s, "start": s, "end": e, "grid": g, "wg": w})
prev = None
busy_total = 0
for gr in groups:
    gap = (gr["start"] - prev) / 1e3 if prev else 0.0
    busy_total += gr["busy"]
    print("%9.3f ms  gap %8.1f us  %-44s x%-5d busy %9.1f us  span %9.1f us  grid %s/%s" % (
        (gr["start"] - t0) / 1e6, gap, gr["nm"], gr["n"], gr["busy"] / 1e3, (gr["end"] - gr["start"]) / 1e3, gr["grid"], gr["wg"]))
    prev = gr["end"]
print("window %.1f ms, GPU busy %.1f ms" % ((tend - t0) / 1e6, busy_total / 1e6))
