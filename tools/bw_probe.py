import torch
dev="cuda:0"
x=torch.rand(1,1024,1024,11,11,device=dev)
def t(fn,n=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
b=x.numel()*4
ts=t(lambda: x.sum()); print(f"sum   {ts*1e3:7.1f} us  {b/ts/1e9:6.2f} TB/s read")
y=torch.empty_like(x)
tc=t(lambda: y.copy_(x)); print(f"copy  {tc*1e3:7.1f} us  {2*b/tc/1e9:6.2f} TB/s r+w")
tm=t(lambda: x.amax()); print(f"amax  {tm*1e3:7.1f} us  {b/tm/1e9:6.2f} TB/s read")
