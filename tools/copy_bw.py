import torch,time
for mb in (24,48,96,256,1024):
    n=mb*1024*1024//4
    a=torch.empty(n,device='cuda'); b=torch.randn(n,device='cuda')
    for _ in range(5): a.copy_(b)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): a.copy_(b)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/50
    print(f"copy {mb} MB: {us:.1f} us  -> {2*mb*1.048576/us*1e3/1e3:.2f} TB/s (read+write)")
