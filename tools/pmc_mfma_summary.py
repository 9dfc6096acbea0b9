"""Condense the MFMA counter passes over `tools/pmc_driver mfma` into profiles/pmc_mfma_rNN.json.
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-include-regex "syrk|gemm" --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- tools/pmc_driver mfma
  rocprofv3 --pmc MfmaUtil ... -d gpurun_out/pmc_mfma2 -- tools/pmc_driver mfma
Usage: python tools/pmc_mfma_summary.py 01"""
import collections, csv, glob, json, os, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"command": "rocprofv3 --pmc <counters> --kernel-include-regex 'syrk|gemm' --kernel-trace --output-format csv -- tools/pmc_driver mfma "
                  "(f32 covariance SYRK n = T = 4096, f32 and bf16 layer-output GEMM 4096^3; 3 launches each, averages)",
       "note": "MfmaUtil is rocprofv3's derived metric (gfx94x formula on this ROCm); busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / "
               "(GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs)", "kernels": {}}
for d in ("pmc_mfma", "pmc_mfma2"):
    f = sorted(glob.glob(os.path.join(root, "gpurun_out", d, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"].replace("ptd::(anonymous namespace)::", "").replace("ptd::", "").replace("void ", "").split("(")[0]
        agg[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for nm, c in agg.items():
        out["kernels"].setdefault(nm, {}).update({k: sum(v) / len(v) for k, v in c.items()})
for nm, c in out["kernels"].items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        c["busy_fraction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
sys.path.insert(0, root)
from ptdeco_amd import _hip  # noqa: E402  (provenance: bench.py marks the figures stale when the kernel sources change)
out["source_sha16"] = _hip.source_sha16("gemm_f32.hip", "gemm_bf16.hip")
out["source_files"] = ["ptdeco_amd/csrc/gemm_f32.hip", "ptdeco_amd/csrc/gemm_bf16.hip"]
json.dump(out, open(os.path.join(root, "profiles", f"pmc_mfma_r{rnd}.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
