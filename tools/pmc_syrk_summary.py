"""Condense the counter passes over `tools/pmc_driver syrk` (bf16 covariance product at n = T = 4096 and at the Llama-3-8B
calibration shapes, three launches per shape) into profiles/pmc_syrk_rNN.json: per shape the launch time from the kernel
trace, the matrix-pipe busy share, the memory-side bytes (2 x FETCH_SIZE + WRITE_SIZE, x 1024: MI355X_MICROARCH.md) and
both roofs -- the MFMA peak on the triangle's flops and the HBM peak on y once + the live triangle of E read and written.
  for c in MfmaUtil FETCH_SIZE WRITE_SIZE: rocprofv3 --pmc $c --kernel-include-regex 'syrk|gemm_bf16' --kernel-trace --output-format csv -d gpurun_out/pmcs_$c -- tools/pmc_driver syrk
Usage: python tools/pmc_syrk_summary.py 04"""
import csv, glob, json, os, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = [(4096, 4096, 1), (4096, 2048, 1), (1024, 2048, 1), (14336, 2048, 1),      # (n, T, calibration steps per launch)
          (4096, 2048, 8), (1024, 2048, 8), (14336, 2048, 8)]                      # ptd_syrk_accumulate_multi (round 5)


def rows(d, suffix):
    f = sorted(glob.glob(os.path.join(root, "gpurun_out", d, "*", "*" + suffix)), key=os.path.getmtime)[-1]
    return list(csv.DictReader(open(f)))


def per_dispatch(d, counter):
    """counter value per syrk dispatch, in dispatch order (a kernel's value = the sum over its rows: one per XCD / SE)"""
    acc = {}
    for r in rows(d, "counter_collection.csv"):
        if not ("syrk" in r["Kernel_Name"] or "gemm_bf16" in r["Kernel_Name"]) or r["Counter_Name"] != counter:
            continue
        acc[int(r["Dispatch_Id"])] = acc.get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    return [acc[k] for k in sorted(acc)]


def durations(d):
    t = [(int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, r["Kernel_Name"])
         for r in rows(d, "kernel_trace.csv") if "syrk" in r["Kernel_Name"] or "gemm_bf16" in r["Kernel_Name"]]
    return sorted(t)


mf, fe, wr = per_dispatch("pmcs_MfmaUtil", "MfmaUtil"), per_dispatch("pmcs_FETCH_SIZE", "FETCH_SIZE"), per_dispatch("pmcs_WRITE_SIZE", "WRITE_SIZE")
du = durations("pmcs_MfmaUtil")
assert len(du) == 3 * len(shapes) == len(mf) == len(fe) == len(wr), (len(du), len(mf), len(fe), len(wr))   # one kernel per call
out = {"command": "rocprofv3 --pmc <MfmaUtil | FETCH_SIZE | WRITE_SIZE> --kernel-include-regex 'syrk|gemm_bf16' --kernel-trace --output-format csv -- "
                  "tools/pmc_driver syrk (bf16 y, f64 accumulator, three launches per shape; separate passes per counter)",
       "note": "us = launch duration in the MfmaUtil pass (profiled clocks run a few per cent low); traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 "
               "bytes per launch (gfx950: FETCH_SIZE counts half of a wide streaming read); algorithmic bytes = 2 T n (y once) + 8 n (n + 1) "
               "(the live triangle of E read and written ONCE per launch); flops = T n (n + 1) per step; steps_per_launch = 8: "
               "ptd_syrk_accumulate_multi", "shapes": []}
for i, (n, t, steps) in enumerate(shapes):
    sl = slice(3 * i, 3 * i + 3)
    us = sum(x[1] for x in du[sl]) / 3
    flops, byts = steps * t * n * (n + 1), steps * 2 * t * n + 8 * n * (n + 1)
    traffic = (2 * sum(fe[sl]) / 3 + sum(wr[sl]) / 3) * 1024
    out["shapes"].append({"n": n, "T": t, "steps_per_launch": steps, "us_per_step": us / steps,
                          "kernel": du[3 * i][2].replace("ptd::(anonymous namespace)::", "").split("(")[0], "us": us,
                          "MfmaUtil": sum(mf[sl]) / 3, "tflops": flops / us / 1e6, "frac_of_bf16_mfma_peak": flops / us / 1e6 / 2500,
                          "traffic_bytes": traffic, "algorithmic_bytes": byts, "traffic_over_algorithmic": traffic / byts,
                          "algorithmic_gbps": byts / us / 1e3, "frac_of_hbm_peak": byts / us / 1e3 / 8000,
                          "mfma_bound_us": flops / 2.5e15 * 1e6, "hbm_bound_us": byts / 8e12 * 1e6})
sys.path.insert(0, root)
from ptdeco_amd import _hip  # noqa: E402
out["source_sha16"] = _hip.source_sha16("gemm_bf16.hip")
json.dump(out, open(os.path.join(root, "profiles", f"pmc_syrk_r{rnd}.json"), "w"), indent=1)
print(json.dumps(out["shapes"], indent=1))
