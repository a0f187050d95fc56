import torch
dev="cuda"
big=torch.empty(256<<20,device=dev)
def t(fn,cold,n=10):
    fn(); torch.cuda.synchronize()
    tot=0
    for _ in range(n):
        if cold: big.fill_(1.0)
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); tot+=a.elapsed_time(b)
    return tot/n*1e3
for mb in (35,70,141,282,1128):
    n=mb*1000*1000//4
    x=torch.rand(n,device=dev); y=torch.empty_like(x)
    print("copy of %4d MB (traffic %4d MB): warm %.1f us cold %.1f us | sum (read only): warm %.1f cold %.1f"%(mb,2*mb,t(lambda:y.copy_(x),False),t(lambda:y.copy_(x),True),t(lambda:x.sum(),False),t(lambda:x.sum(),True)),flush=True)
