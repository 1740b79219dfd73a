"""compose_pair_kernel alone at the bench shape (32 x 3 x 224 x 224 per view, bf16 channels-last output, key rows in a
shuffled order) for the `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes -> profiles/compose_traffic.json."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from cp2_amd import ops, synthetic
b = synthetic.make_batch(32, 224, 224, 'cuda', seed=0)
perm = torch.randperm(32).cuda()
flush = torch.empty(192 << 20, device='cuda')
for i in range(6):
    flush.fill_(float(i))          # push the inputs out of the 256 MiB infinity cache, as the rest of a step does
    ops.compose_pair(b["img_a"], b["bg0"], b["img_b"], b["bg1"], 16, perm, True, torch.bfloat16)
torch.cuda.synchronize()
print("done")
