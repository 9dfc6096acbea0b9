"""BASELINE configs[2] alone (bench.c3_line): seconds, phases.  python tools/probes/c3_run.py"""
import json, os, sys, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import bench
r = bench.c3_line(torch.device("cuda", 0))
print(json.dumps({k: v for k, v in r.items() if k != "workload"}), flush=True)
