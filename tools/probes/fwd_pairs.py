"""The decomposed forward's pair against the library's at T = 4096 / 16384 / 65536 (bench.decomposed_forward_lines,
short form), once per value of the environment switches given as arguments (NAME=V[,NAME=V])."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
for env in sys.argv[1:] or [""]:
    for kv in filter(None, env.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    for T in (4096, 16384, 65536):
        d = bench.decomposed_forward_lines(dev, T, full=False)
        print(env, f"T={T} dense {d['dense_ms'] * 1e3:.0f} / lib {d['dense_torch_hipblaslt_ms'] * 1e3:.0f} us;",
              "; ".join(f"r={r}: {d['r%d' % r]['ms'] * 1e3:.1f} / lib {d['r%d' % r]['torch_hipblaslt_pair_ms'] * 1e3:.1f}" for r in (256, 512, 1024)), flush=True)
