import torch, time
dev="cuda"
N=1<<29            # 536M elements
idx=torch.randint(0,N,(N//2,),device=dev,dtype=torch.int64)
src4=torch.arange(N,device=dev,dtype=torch.int32)
src8=torch.arange(N,device=dev,dtype=torch.int64)
vals=torch.arange(N//2,device=dev,dtype=torch.int32)
def t(fn,name,n):
    fn(); torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/3
    print("%-40s %7.2f ms  %.2e /s" % (name, dt*1e3, n/dt))
out=torch.empty(N//2,device=dev,dtype=torch.int32)
t(lambda: torch.index_select(src4,0,idx,out=out),"gather 4B random (2 GB table)",N//2)
out8=torch.empty(N//2,device=dev,dtype=torch.int64)
t(lambda: torch.index_select(src8,0,idx,out=out8),"gather 8B random (4 GB table)",N//2)
dst4=torch.zeros(N,device=dev,dtype=torch.int32)
t(lambda: dst4.index_copy_(0,idx,vals),"scatter 4B random (2 GB table)",N//2)
# sorted indices (ascending sparse)
sidx,_=torch.sort(idx)
t(lambda: torch.index_select(src4,0,sidx,out=out),"gather 4B ascending sparse",N//2)
t(lambda: dst4.index_copy_(0,sidx,vals),"scatter 4B ascending sparse",N//2)
# chained dependent gathers: 3 hops
def chase():
    a=torch.index_select(src8,0,idx)
    a=torch.index_select(src8,0,a)
    a=torch.index_select(src8,0,a)
    return a
