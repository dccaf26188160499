import torch
dev=torch.device('cuda:0')
T=1000108
y=torch.empty((T,1152),device=dev,dtype=torch.float16)
z=torch.randn((T,1152),device=dev).half()
x=torch.randn((T,384),device=dev).half()
def timed(fn,it=10):
    fn();fn();torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record();torch.cuda.synchronize()
    return a.elapsed_time(b)/it
ms=timed(lambda: y.fill_(1.0)); print('fill 2.3 GB: %.3f ms  %.0f GB/s'%(ms, y.numel()*2/ms/1e6))
ms=timed(lambda: y.copy_(z)); print('copy 2.3 GB: %.3f ms  %.0f GB/s (r+w)'%(ms, 2*y.numel()*2/ms/1e6))
y3=y.view(T,3,384)
ms=timed(lambda: y3.copy_(x[:,None,:].expand(T,3,384))); print('expand-copy read .77 write 2.3: %.3f ms  %.0f GB/s alg'%(ms, (x.numel()*2+y.numel()*2)/ms/1e6))
