import torch
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for (cin, cout, hw, k) in ((64, 64, 56, 3), (64, 128, 28, 3), (128, 128, 28, 3), (256, 256, 14, 3), (64, 128, 56, 1)):
    conv = torch.nn.Conv2d(cin, cout, k, padding=k // 2, bias=False).to(dev).eval()
    x = torch.randn(8, cin, hw, hw, generator=g, device=dev)
    with torch.no_grad():
        ys = [conv(x) for _ in range(4)]
    print((cin, cout, hw, k), [torch.equal(ys[0], y) for y in ys[1:]], [float((ys[0] - y).abs().max()) for y in ys[1:]])
