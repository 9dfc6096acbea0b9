import json,sys
d=json.load(open(sys.argv[1]))
for dt in ("f32","bf16"):
    b=d["c4_block"][dt]; print(sys.argv[1], "block",dt, round(b["ms_per_step"],1), b["step_ms"], "B_eigh", b["phases_ms"]["B_eigh"])
print("  bf16_stack", d["bf16_stack"]["step_ms"], "stack B_eigh", d["stack_phases_ms"]["B_eigh"], "value", round(d["value"],2))
