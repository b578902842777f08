import torch, time
dev='cuda:0'
def bench(fn,n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
for gb in (0.27, 2.4):
    n=int(gb*1e9/2)
    x=torch.empty(n,dtype=torch.bfloat16,device=dev); y=torch.empty(n,dtype=torch.bfloat16,device=dev)
    t=bench(lambda: x.fill_(1.0)); print(f"fill  {gb} GB: {t*1e3:.0f} us {2*n/t/1e9:.2f} TB/s written")
    t=bench(lambda: x.zero_()); print(f"zero  {gb} GB: {t*1e3:.0f} us {2*n/t/1e9:.2f} TB/s written")
    t=bench(lambda: y.copy_(x)); print(f"copy  {gb} GB: {t*1e3:.0f} us {4*n/t/1e9:.2f} TB/s r+w")
    t=bench(lambda: torch.add(x, 1.0, out=y)); print(f"add   {gb} GB: {t*1e3:.0f} us {4*n/t/1e9:.2f} TB/s r+w")
    t=bench(lambda: x.sum()); print(f"sum   {gb} GB: {t*1e3:.0f} us {2*n/t/1e9:.2f} TB/s read")
