"""Kernel-level timing of the contrastive front end (include/mi_contrastive.h) at 32 x 1080p x 10 gates x 1000 rays: HIP events around
20 forward / backward calls each (through ctypes, no autograd), for A/B runs of library variants (MI_RAST_LIB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seganygaussians_amd.contrastive_frontend import contrastive_front_end  # noqa: E402

C, H, W, N, S = 32, 1080, 1920, 10, 1000
g = torch.Generator().manual_seed(0)
rendered = torch.randn(C, H, W, generator=g).cuda().requires_grad_(True)
gates = torch.rand(N, C, generator=g).cuda().requires_grad_(True)
idx = torch.randperm(H * W, generator=g)[:S].sort().values
ray_yx = torch.stack([idx // W, idx % W], 1).to(torch.int32).cuda()
up = torch.randn(N, S, C, generator=g).cuda()
gn = torch.tensor(0.3).cuda()
ms_f, ms_b = [], []
for it in range(25):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    out, nrm = contrastive_front_end(rendered, (H, W), ray_yx, gates)
    e[1].record()
    torch.autograd.grad([out, nrm], [rendered, gates], [up, gn])
    e[2].record()
    torch.cuda.synchronize()
    if it >= 5:
        ms_f.append(e[0].elapsed_time(e[1]))
        ms_b.append(e[1].elapsed_time(e[2]))
tag = os.path.basename(os.environ.get("MI_RAST_LIB", "default"))
print(f"{tag}: forward {min(ms_f):.4f} ms (median {sorted(ms_f)[len(ms_f) // 2]:.4f}), backward {min(ms_b):.4f} ms (median {sorted(ms_b)[len(ms_b) // 2]:.4f})")
