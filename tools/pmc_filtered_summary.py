"""Condense rocprofv3 --pmc passes over gemm_f64_glds_kernel (tools/pmc_driver eigh: ptd_eigh_topk n = 4096, k = 1024
through the filtered route) into profiles/pmc_gemm_f64_rNN.json.  Usage: python tools/pmc_filtered_summary.py 03

Passes (one counter set per run):
  rocprofv3 --pmc FETCH_SIZE --kernel-include-regex gemm_f64_glds --kernel-trace --output-format csv -d gpurun_out/pmcf_fetch -- tools/pmc_driver eigh
  ... --pmc WRITE_SIZE ... -d gpurun_out/pmcf_write;  ... --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE ... -d gpurun_out/pmcf_mfma
gfx950 correction (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE KiB x 1024; WRITE_SIZE exact.
Only the launches of the product C X of the filter (grid 512 x 1: 4096 x 1280 output in 128 x 80 tiles, K = 4096) are kept.
"""
import csv, glob, json, os, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, m = 4096, 1280


def load(tag):
    f = sorted(glob.glob(os.path.join(root, "gpurun_out", tag, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    out = {}
    for r in csv.DictReader(open(f)):
        if "gemm_f64_glds_kernel<5, false>" not in r["Kernel_Name"]:
            continue
        if int(r["Grid_Size"]) != 512 * 256:      # the n x n x m products (X W of a pass has the same grid: told apart below)
            continue
        out.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return {k: [v for _, v in sorted(vals)] for k, vals in out.items()}


fetch = load("pmcf_fetch")["FETCH_SIZE"]
write = load("pmcf_write")["WRITE_SIZE"]
mf = load("pmcf_mfma")
alg = 8.0 * (n * n + 2.0 * n * m)          # C once, X in, Y out (+ the addend of the Chebyshev step, counted as the out pass)
rd = [2.0 * v * 1024.0 for v in fetch]
wr = [v * 1024.0 for v in write]
# the same template also runs the X W products of the Cholesky-QR passes (K = 1280: ~150 MB fetched); the products with C
# fetch C once (134 MB) and X once per XCD (8 x 42 MB, served by the Infinity Cache): split at 300 MB
big = [i for i, v in enumerate(rd) if v > 300e6]
summary = {
    "kernel": "gemm_f64_glds_kernel<5, false> (C X of the Chebyshev filter: 4096 x 4096 x 1280 f64, 128 x 80 tiles)",
    "command": "rocprofv3 --pmc <COUNTERS> --kernel-include-regex gemm_f64_glds --kernel-trace --output-format csv -- tools/pmc_driver eigh "
               "(separate passes: FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE)",
    "correction": "read bytes = 2 x FETCH_SIZE KiB x 1024 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE exact",
    "launches_seen": len(rd), "products_with_C": len(big),
    "algorithmic_bytes_per_launch": alg,
    "read_bytes_per_launch": sum(rd[i] for i in big) / max(len(big), 1),
    "write_bytes_per_launch": sum(wr[i] for i in big) / max(len(big), 1) if len(wr) == len(rd) else None,
}
if summary["write_bytes_per_launch"] is not None:
    summary["traffic_bytes_per_launch"] = summary["read_bytes_per_launch"] + summary["write_bytes_per_launch"]
    summary["traffic_over_algorithmic"] = summary["traffic_bytes_per_launch"] / alg
    summary["traffic_note"] = ("FETCH_SIZE counts requests leaving the XCD L2s (Infinity-Cache hits included): C is fetched once "
                               "(134 MB), X (42 MB) once per XCD -- the eight L2s do not share; the kernel is matrix-pipe bound "
                               "(busy share below), the re-reads cost no time at ~0.6 TB/s")
if "SQ_VALU_MFMA_BUSY_CYCLES" in mf and "SQ_BUSY_CU_CYCLES" in mf:
    busy, cu = mf["SQ_VALU_MFMA_BUSY_CYCLES"], mf["SQ_BUSY_CU_CYCLES"]
    k = min(len(busy), len(cu))
    # MfmaUtil as rocprofv3 derives it on gfx94x: MFMA busy cycles / (CU busy cycles x 4 SIMDs)
    summary["mfma_busy_over_cu_busy_x4_percent"] = 100.0 * sum(busy[:k]) / (4.0 * sum(cu[:k]))
sys.path.insert(0, root)
from ptdeco_amd import _hip  # noqa: E402
summary["source_sha16"] = _hip.source_sha16("gemm_f64.hip", "eigh_filtered.hip")
summary["source_files"] = ["ptdeco_amd/csrc/gemm_f64.hip", "ptdeco_amd/csrc/eigh_filtered.hip"]
json.dump(summary, open(os.path.join(root, "profiles", f"pmc_gemm_f64_r{rnd}.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
