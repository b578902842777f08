import torch, time
dev = torch.device("cuda:0")
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (B, C, H, W) in ((4, 3, 64, 2656), (4, 8, 64, 2656), (4, 26, 64, 2048), (4, 8, 64, 2048)):
    buf = torch.randn(B, H, W, 32, device=dev)
    g = buf[..., :C].permute(0, 3, 1, 2)
    ref = g.double().sum(dim=(0, 2, 3))
    fns = {
        "sum(0,2,3)": lambda: g.float().sum(dim=(0, 2, 3)),
        "nhwc sum(0,1,2)": lambda: g.permute(0, 2, 3, 1).sum(dim=(0, 1, 2)),
        "two-stage W then rows": lambda: g.permute(0, 2, 3, 1).sum(dim=2).sum(dim=(0, 1)),
        "two-stage rows view": lambda: g.permute(0, 2, 3, 1).reshape(B * H, W, C).sum(dim=1).sum(dim=0),
        "ones matmul": lambda: torch.ones(B * H * W, device=dev) @ g.permute(0, 2, 3, 1).reshape(-1, C),
    }
    for k, f in fns.items():
        try:
            t = bench(f); err = float((f().double() - ref).abs().max() / ref.abs().max())
            print(f"{(B,C,H,W)} {k:26s} {t:8.1f} us  err {err:.1e}")
        except Exception as e:
            print(k, "failed", e)
