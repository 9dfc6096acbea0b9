"""CPU only: runs the oracle alone on the full-width C3 / C4 parity cases (tests/fullwidth_cases.py) for a few
seeds and prints, per seed, the smallest distance of any step of the oracle's run from a threshold it is
compared with -- the tests assert that margin, so the seeds they use are picked here.
Usage: python tools/scan_fullwidth_seeds.py c3|c4 seed [seed ...]"""
import copy, json, os, sys, time, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (root, os.path.join(root, "tests"), os.path.join(root, "oracle")):
    sys.path.insert(0, p)
import fullwidth_cases as fc
import ptdeco_oracle as orc

which, seeds = sys.argv[1], [int(s) for s in sys.argv[2:]]
for seed in seeds:
    t0 = time.perf_counter()
    trace = []
    if which == "c3":
        model, pool = fc.c3_case(seed, seed + 1, depth=int(os.environ.get("C3_DEPTH", "3")))
        cfg = orc.falor_decompose(module=model, data_iterator=fc.cycle(pool), trace=trace, **fc.C3_KW)
        rel = min(min(abs(r["nsr"] / fc.C3_KW["nsr_final_threshold"] - 1), abs(r["kl"] / fc.C3_KW["kl_final_threshold"] - 1))
                  for r in trace)
        out = {"seed": seed, "steps": len(trace), "replaced": list(cfg), "min_rel_margin": rel}
    else:
        model, batches = fc.c4_case(seed)
        cfg = orc.dwain_decompose(module=model, data_iterator=fc.cycle(batches), loss_fn=fc.seq_ce,
                                  metric_iterator=fc.cycle(batches[5:]), trace=trace, **fc.C4_KW)
        rel = min(min(abs(t["ppl_diff"] - t["threshold"]), abs(t["ppl_diff"] - fc.C4_KW["max_accepted_ppl_diff"]),
                      abs(t["nsr"] - fc.C4_KW["nsr_final_threshold"])) / max(abs(t["ppl_diff"]), 1e-12) for t in trace)
        out = {"seed": seed, "steps": len(trace), "replaced": {k: v["__meta__"]["proportion"] for k, v in cfg.items()},
               "min_rel_margin": rel, "trace": [(t["layer"], t["rank"], t["accepted"], round(t["ppl_diff"], 6), round(t["threshold"], 6), round(t["nsr"], 5)) for t in trace]}
    out["seconds"] = time.perf_counter() - t0
    print(json.dumps(out), flush=True)
