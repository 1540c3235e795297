import torch, time
x = torch.randn(16,160,160,128, device='cuda').bfloat16()
ys = [torch.empty_like(x) for _ in range(4)]
xs = [x.clone() for _ in range(4)]
for n in range(3):
    for i in range(4): ys[i].copy_(xs[i])
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for r in range(10):
    for i in range(4): ys[i].copy_(xs[i])
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/40
print(f"copy 105MB->105MB: {ms*1e3:.1f} us, {2*x.numel()*2/ms/1e9:.2f} TB/s")
