import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import torch
import voxel_sweep as V
torch.cuda.set_device(0)
for (H, W) in ((480, 640), (512, 512), (384, 768), (512, 768)):
    V.timing(10_000_000, H, W, 5, paths=("v2",))
