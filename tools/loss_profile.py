"""Times the loss kernels alone at 1080p (run under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import sys
sys.path[:0] = ['/root/repo', '/root/repo/3d-gaussian-splatting_amd']
import torch
from gs_train import ImageLoss
dev = torch.device('cuda:0')
H, W = 1080, 1920
x, y = torch.rand(H, W, 3, device=dev), torch.rand(H, W, 3, device=dev)
loss = ImageLoss(H, W, 0.1, dev)
for _ in range(30):
    loss(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    loss(x, y)
e1.record(); e1.synchronize()
print("loss ms", e0.elapsed_time(e1) / 50)
