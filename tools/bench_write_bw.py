import torch
x=torch.empty(8*160**3*48, dtype=torch.bfloat16, device='cuda')
for fn,name in ((lambda: x.zero_(),"zero_"),(lambda: x.fill_(1.5),"fill_")):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize(); ms=a.elapsed_time(b)/10
    print(name, ms, "ms", x.numel()*2/ms/1e6, "GB/s")
