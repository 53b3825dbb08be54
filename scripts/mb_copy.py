import torch
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e3
for mb in (64, 268, 537, 1074):
    n = mb*1000*1000//2
    x=torch.randn(n,device="cuda",dtype=torch.bfloat16); y=torch.empty_like(x)
    tc=t(lambda: y.copy_(x)); ta=t(lambda: torch.add(x,x,out=y)); tm=t(lambda: y.zero_())
    print(f"{mb} MB: copy {tc:7.1f}us {2*n*2/tc/1e6:5.2f} TB/s | add(x,x) {ta:7.1f}us {2*n*2/ta/1e6:5.2f} TB/s | fill {tm:7.1f}us {n*2/tm/1e6:5.2f} TB/s")
