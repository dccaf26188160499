"""Two ranks on one GPU over gloo: all_gather_into_tensor of CUDA tensors shaped like the embedding exchange."""
import os, sys, time
import torch
import torch.distributed as dist
dist.init_process_group('gloo')
r, w = dist.get_rank(), dist.get_world_size()
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
for shape in ((32, 100), (32, 768), (512, 768)):
    piece = torch.full(shape, float(r), device=dev)
    out = torch.empty((w * shape[0], shape[1]), device=dev)
    t0 = time.time()
    dist.all_gather_into_tensor(out, piece)
    torch.cuda.synchronize()
    print(r, shape, 'ok', round(time.time() - t0, 3), float(out[0, 0]), float(out[-1, 0]), flush=True)
dist.barrier()
dist.destroy_process_group()
