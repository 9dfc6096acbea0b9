import cProfile, pstats, sys, runpy, io
sys.argv = ["tools/c4_stack.py", "2", "bf16"]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path("/root/repo/tools/c4_stack.py", run_name="__main__")
finally:
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
