"""Which solver each ptd_eigh call of the C3 run (tools/c3_vit.py) ends up on, and how long it takes."""
import json, os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from ptdeco_amd import ops
ops.EIGH_PROFILE = []
sys.argv = ["tools/c3_vit.py"] + sys.argv[1:]
runpy.run_path(os.path.join(root, "tools", "c3_vit.py"), run_name="__main__")
rows = {}
for p in ops.EIGH_PROFILE:
    key = (p["n"], p["k"], p["method"])
    r = rows.setdefault(key, {"calls": 0, "ms": 0.0, "sweeps": 0})
    r["calls"] += 1; r["ms"] += p["total_ms"]; r["sweeps"] += p["sweeps"]
print(json.dumps([{"n": k[0], "k": k[1], "method": k[2], **{a: round(b, 2) for a, b in v.items()}} for k, v in sorted(rows.items())]))
