"""On-device input pipeline at the bench shape (32 images per GPU, 224 x 224 views from a uint8 dataset of 256 x 320 images):
make_step_batch with and without the photometric transforms, for rocprofv3 --kernel-trace (profiles/r03_augment_kernels.txt)."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import time
import numpy as np
import torch
from cp2_amd import augment as A
g = torch.Generator().manual_seed(0)
ds = A.DeviceDataset(torch.randint(0, 256, (512, 3, 256, 320), dtype=torch.uint8, generator=g))
rng = np.random.default_rng(0)
s = [A.EpochSampler(len(ds), 1, 0, seed).indices(0) for seed in (0, 1024, 2048)]
for photometric in (True, False):
    for i in range(3):
        A.make_step_batch(ds, s[0][:32], s[1][:32], s[2][:32], 224, 224, rng, photometric=photometric)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for i in range(n):
        A.make_step_batch(ds, s[0][:32], s[1][:32], s[2][:32], 224, 224, rng, photometric=photometric)
    torch.cuda.synchronize()
    print(f"photometric={photometric}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step batch (4 x 32 views, host + device)")
