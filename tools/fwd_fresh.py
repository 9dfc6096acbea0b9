"""bench.decomposed_forward_lines in a fresh process (no eigensolver run before it), twice: how much of the distance
between bench.py's forward numbers and tools/cold_inputs.py is the state of the process / chip."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
for rnd in range(2):
    f = bench.decomposed_forward_lines(dev)
    print(json.dumps({"round": rnd, "dense_ms": f["dense_ms"], "dense_lib_ms": f["dense_torch_hipblaslt_ms"],
                      **{r: {k: round(f[r][k], 4) for k in ("ms", "ms_rotating_inputs", "torch_hipblaslt_pair_ms", "torch_hipblaslt_pair_ms_rotating_inputs", "module_ms")} for r in ("r256", "r512", "r1024")}}), flush=True)
