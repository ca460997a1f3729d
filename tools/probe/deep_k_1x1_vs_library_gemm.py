import torch, time, sys
sys.path.insert(0,'/root/repo')
from efficientteacher_amd import ops
dev=torch.device('cuda:0')
def timeit(f, iters=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/iters*1e3
for (B,h,cin,cout) in [(64,40,512,512),(64,40,512,256),(64,40,1024,256),(64,20,1024,512),(64,20,1024,1024),(64,20,2048,1024),(64,20,512,512),(64,80,512,128),(32,40,512,512),(32,20,1024,1024)]:
    M=B*h*h
    xs=[torch.randn(B,h,h,cin,device=dev).bfloat16() for _ in range(3)]
    w=(torch.randn(cout,1,1,cin,device=dev)*cin**-0.5).bfloat16()
    w2=w.reshape(cout,cin)
    ys=[torch.empty(B,h,h,cout,device=dev,dtype=torch.bfloat16) for _ in range(3)]
    i=[0]
    def mine():
        k=i[0]%3; i[0]+=1
        ops.conv2d_fwd(xs[k], w, 1, 0, out=ys[k])
    def lib():
        k=i[0]%3; i[0]+=1
        torch.mm(xs[k].view(M,cin), w2.t(), out=ys[k].view(M,cout))
    tm=timeit(mine); tl=timeit(lib)
    fl=2.0*M*cin*cout
    print(f"B{B} {cin:5d}->{cout:5d} @{h:3d}  mine {tm:7.1f} us {fl/tm/1e6:6.0f} TF | torch.mm {tl:7.1f} us {fl/tl/1e6:6.0f} TF   {ops.kernel_name('fwd', torch.bfloat16, B,h,h,cin,cout,1,1,0)[:40]}", flush=True)
