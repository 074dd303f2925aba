import torch, time
dev=torch.device("cuda:0")
for mib in (256,512,1024,2048,4096):
    n=mib*2**20//4
    x=torch.rand(n,device=dev); y=torch.empty_like(x)
    for _ in range(20): y.copy_(x)
    torch.cuda.synchronize()
    best=1e9
    for rep in range(5):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y.copy_(x)
        e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)/20)
    print(f"copy {mib} MiB per buffer: {best:.4f} ms  {2*mib*2**20/best/1e6:.0f} GB/s", flush=True)
    del x,y
