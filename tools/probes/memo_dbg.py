import itertools, os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import toy_models as tm
from ptdeco_amd import _engine as eng
DEV = torch.device("cuda", 0)
torch.manual_seed(271828)
model = tm.ResNet18().eval().to(DEV)
g = torch.Generator().manual_seed(1314159)
x = torch.rand(5, 3, 224, 224, generator=g).to(DEV)
# record inputs and outputs of every conv in two consecutive plain forwards
def run():
    rec = []
    hs = [m.register_forward_hook(lambda mod, a, o, n=n: rec.append((n, a[0].clone(), o.clone())))
          for n, m in model.named_modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear))]
    with torch.no_grad():
        y = model(x)
    for h in hs: h.remove()
    return rec, y
r1, y1 = run()
r2, y2 = run()
r3, y3 = run()
for (n, i1, o1), (_, i2, o2), (_, i3, o3) in zip(r1, r2, r3):
    print(n, "in12", torch.equal(i1, i2), "out12", torch.equal(o1, o2), float((o1 - o2).abs().max()), "in23", torch.equal(i2, i3), "out23", torch.equal(o2, o3))
