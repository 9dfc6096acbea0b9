"""Host-side profile of the C3 run (cProfile, sorted by own time and by cumulative time)."""
import cProfile, pstats, sys, runpy, io, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = ["tools/c3_vit.py"] + sys.argv[1:]
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path(os.path.join(root, "tools", "c3_vit.py"), run_name="__main__")
finally:
    pr.disable()
for key, cnt in (("tottime", 25), ("cumulative", 45)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(cnt)
    print(s.getvalue()[:9000])
