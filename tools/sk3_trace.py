import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PTD_SK3_DEBUG"] = "1"
from ptdeco_amd import ops, _hip
dev = torch.device("cuda")
T, r = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 256
h = torch.randn(T, r, device=dev, dtype=torch.bfloat16); b = torch.randn(4096, r, device=dev, dtype=torch.bfloat16) / 16
dbg = torch.zeros(4096, dtype=torch.int64, device=dev)
fake_bias = dbg.view(torch.bfloat16)   # the debug build writes timestamps through the bias pointer
c = torch.empty(T, 4096, dtype=torch.bfloat16, device=dev)
lib = _hip.load()
for it in range(3):
    rc = lib.ptd_gemm(h.data_ptr(), r, 1, b.data_ptr(), 1, r, c.data_ptr(), 4096, T, 4096, r, 2, 2, 1.0, fake_bias.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
d = dbg.cpu().tolist()
t0 = d[0]
steps = T // 16 // 64
print("cycles (clock64 = 100 MHz? see scale) relative to loop start; columns: top, after wait, after barrier, after stage, after mfma, after store")
prev = t0
for k in range(steps):
    row = d[8 + k * 8: 8 + k * 8 + 6]
    print(k, [x - t0 for x in row], "step", row[5] - prev); prev = row[5]
