"""Error of a split-precision GEMM (hi.hi + hi.lo + lo.hi on the 16-bit matrix cores, fp32 accumulation)
for bf16 and fp16 halves, with and without a power-of-two scale on the weights; against float64."""
import torch
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(0)
M, K, N = 4096, 384, 1536
x = torch.randn((M, K), generator=g, device=dev)
w = torch.randn((N, K), generator=g, device=dev) * 0.03
ref = x.double() @ w.double().t()
def split3(a, dt, left):
    hi = a.to(dt); lo = (a - hi.float()).to(dt)
    return torch.cat([hi, hi, lo], 1) if left else torch.cat([hi, lo, hi], 1)
def run(dt, scale_w=1.0, scale_x=1.0):
    y = torch.mm(split3(x * scale_x, dt, True), split3(w * scale_w, dt, False).t(), out_dtype=torch.float32)
    y = y.double() / (scale_w * scale_x)
    return float((y - ref).abs().max()), float((y - ref).abs().mean())
# subnormal inputs kept?
a = torch.full((64, 64), 3e-6, device=dev, dtype=torch.float16); b = torch.full((64, 64), 1.0, device=dev, dtype=torch.float16)
print('fp16 subnormal operand product (expect 64*3e-6=1.9e-4):', float(torch.mm(a, b, out_dtype=torch.float32)[0, 0]))
print('fp32 GEMM            ', float(((x @ w.t()).double() - ref).abs().max()), float(((x @ w.t()).double() - ref).abs().mean()))
print('bf16 x3              ', run(torch.bfloat16))
print('fp16 x3              ', run(torch.float16))
print('fp16 x3, w * 2^10    ', run(torch.float16, 1024.0))
print('fp16 x3, w*2^10 x*2^4', run(torch.float16, 1024.0, 16.0))
print('ref magnitude', float(ref.abs().mean()))
