"""Condense rocprofv3 --pmc passes over sytrd_symv_kernel (tools/pmc_driver) into
profiles/pmc_symv_rNN.{json,csv}.  Usage: python tools/pmc_summary.py 01 [n]

Passes (one counter set per run, as the PMC slot table requires):
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex sytrd_symv --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- tools/pmc_driver 4096
  rocprofv3 --pmc WRITE_SIZE ... -d gpurun_out/pmc_write ...
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ... -d gpurun_out/pmc_tcc ...
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE (KiB) tallies the 128-byte requests of a
wide coalesced stream at 64 bytes, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.
"""
import csv, glob, json, os, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "01"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
count = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # matrices per launch (tools/pmc_driver batched <count>)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(tag):
    f = sorted(glob.glob(os.path.join(root, "gpurun_out", tag, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    out = {}
    for r in csv.DictReader(open(f)):
        out.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["Grid_Size"])))
    return {k: sorted(v) for k, v in out.items()}


fetch = load("pmc_fetch")["FETCH_SIZE"]
write = load("pmc_write")["WRITE_SIZE"]
tcc = load("pmc_tcc")
hit, miss = tcc["TCC_HIT_sum"], tcc["TCC_MISS_sum"]
L = len(fetch)
# one SYMV launch per column j = 0 .. L - 1; the last columns (trailing order <= 3072) run in the resident kernels
assert L == len(write) == len(hit) == len(miss) and L <= n - 1, (L, len(write), len(hit))
rows = []
for j in range(L):
    m = n - j - 1
    alg = count * 8.0 * m * (m - 1)              # rows j+1.., columns j+2.. of the trailing matrix, f64, every matrix
    rd = 2.0 * fetch[j][1] * 1024.0              # gfx950 correction
    wr = write[j][1] * 1024.0
    rows.append((j, m, alg, rd, wr, hit[j][1], miss[j][1]))
with open(os.path.join(root, "profiles", f"pmc_symv_r{rnd}.csv"), "w") as f:
    f.write("column_j,trailing_m,algorithmic_bytes,read_bytes_corrected,write_bytes,tcc_hit,tcc_miss\n")
    for r in rows:
        f.write("%d,%d,%.0f,%.0f,%.0f,%.0f,%.0f\n" % r)
tot_alg = sum(r[2] for r in rows)
tot_rd = sum(r[3] for r in rows)
tot_wr = sum(r[4] for r in rows)
big = [r for r in rows if r[1] >= 2048]
summary = {
    "kernel": "sytrd_symv2_kernel (trailing order >= 1024: lower triangle only) + sytrd_symv_kernel (smaller trailing orders)",
    "n": n, "matrices_per_launch": count, "launches": L, "columns_without_a_launch": n - 1 - L,
    "command": "rocprofv3 --pmc <COUNTER> --kernel-include-regex sytrd_symv --kernel-trace --output-format csv -- tools/pmc_driver %s (separate passes: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum)" % (n if count == 1 else "batched %d" % count),
    "correction": "read bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE exact",
    "algorithmic_bytes_per_launch": tot_alg / L,
    "read_bytes_per_launch": tot_rd / L,
    "write_bytes_per_launch": tot_wr / L,
    "traffic_bytes_per_launch": (tot_rd + tot_wr) / L,
    "traffic_over_algorithmic": (tot_rd + tot_wr) / tot_alg,
    "traffic_over_algorithmic_m_ge_2048": sum(r[3] + r[4] for r in big) / sum(r[2] for r in big),
    "first_launch": {"algorithmic": rows[0][2], "read": rows[0][3], "write": rows[0][4]},
    "l2_hit_rate": sum(r[5] for r in rows) / (sum(r[5] for r in rows) + sum(r[6] for r in rows)),
}
sys.path.insert(0, root)
from ptdeco_amd import _hip  # noqa: E402  (provenance: bench.py marks the figures stale when the kernel source changes)
summary["source_sha16"] = _hip.source_sha16("eigh_tridiag.hip")
summary["source_files"] = ["ptdeco_amd/csrc/eigh_tridiag.hip"]
json.dump(summary, open(os.path.join(root, "profiles", f"pmc_symv_r{rnd}.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
