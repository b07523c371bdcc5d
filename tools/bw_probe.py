"""What torch's own streaming ops reach on this box (copy, sum, comparison + sum over 2.5 GB): the yardstick quoted in DESIGN.md 3.4.
    python tools/bw_probe.py"""
import torch
dev=torch.device('cuda')
n=160*160*192*128
x=torch.rand(n,device=dev)
y=torch.empty_like(x)
def t(f,reps=5):
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
ms=t(lambda: x.sum()); print('sum   %.3f ms  %.0f GB/s read'%(ms, n*4/ms/1e6))
ms=t(lambda: torch.max(x)); print('max   %.3f ms  %.0f GB/s read'%(ms, n*4/ms/1e6))
ms=t(lambda: y.copy_(x)); print('copy  %.3f ms  %.0f GB/s r+w'%(ms, 2*n*4/ms/1e6))
ms=t(lambda: y.fill_(1.0)); print('fill  %.3f ms  %.0f GB/s write'%(ms, n*4/ms/1e6))
b=(x>0.5).to(torch.uint8)
ms=t(lambda: b.sum()); print('u8sum %.3f ms  %.0f GB/s read'%(ms, n/ms/1e6))
