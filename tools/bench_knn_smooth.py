#!/usr/bin/env python
"""Measurement of the fused KNN feature smoothing on one MI355X.

Workload: P = 1 M Gaussians, C = 32, K = 16 cached neighbours, 8 columns drawn per iteration (the reference's
get_smoothed_point_features(K=16, dropout=0.5) followed by the renderer's re-normalisation), forward + backward.
The neighbour map is synthetic but spatially local (points ordered along a Morton curve, neighbours = the 16
nearest positions on the curve): the one-off KNN build is outside this row.  Prints one JSON line:
per-pass kernel times (HIP events on the launch stream), algorithmic bytes and HBM-roofline fraction, and the
reference expression (plain PyTorch, CPU threads) timed on a bounded sample as the CPU baseline."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seganygaussians_amd import knn_smooth as ks  # noqa: E402

HBM_PEAK_GBPS = 8000.0


def morton_local_map(xyz: torch.Tensor, K: int) -> torch.Tensor:
    q = ((xyz - xyz.min(0).values) / (xyz.max(0).values - xyz.min(0).values + 1e-9) * 1023).long().clamp(0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    order = torch.argsort(code)
    P = xyz.size(0)
    pos = torch.empty(P, dtype=torch.long, device=xyz.device)
    pos[order] = torch.arange(P, device=xyz.device)
    offs = torch.tensor([0] + [s * d for d in range(1, K // 2 + 1) for s in (1, -1)][: K - 1], device=xyz.device)
    nb = (pos[:, None] + offs[None, :]).clamp(0, P - 1)
    return order[nb]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    P, C, K, k = args.points, 32, 16, 8
    g = torch.Generator(device="cpu").manual_seed(0)
    xyz = torch.randn(P, 3, generator=g).to(dev)
    F = torch.randn(P, C, generator=g).to(dev).requires_grad_(True)
    dL = torch.randn(P, C, generator=g).to(dev)
    nmap = ks.NeighbourMap(morton_local_map(xyz, K))
    cols = torch.randperm(K, generator=g)[:k]

    def step():
        F.grad = None
        out = ks.smooth_point_features(F, nmap, cols, True)
        out.backward(dL)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(args.steps):
        F.grad = None
        ev[0].record()
        out = ks.smooth_point_features(F, nmap, cols, True)
        ev[1].record()
        out.backward(dL)
        ev[2].record()
        torch.cuda.synchronize()
        tf += ev[0].elapsed_time(ev[1])
        tb += ev[1].elapsed_time(ev[2])
    tf, tb = tf / args.steps, tb / args.steps
    bytes_f = P * (k * (4 * C + 4) + 4 * C)
    bytes_b = P * (k * (4 * C + 4) + 2 * 4 * C) + P * (k * (4 * C + 4) + 2 * 4 * C)
    # CPU baseline: the reference's own expression (gaussian_model_ff.py:354-362 + renderer :362-363) on host threads
    n = min(P, args.cpu_sample)
    Fc = F.detach()[:n].cpu().requires_grad_(True)
    idc = nmap.idx[:n].long().cpu().clamp(0, n - 1)
    dLc = dL[:n].cpu()
    t0 = time.perf_counter()
    normed = torch.nn.functional.normalize(Fc, dim=-1, p=2)
    ret = normed[idc[:, cols], :].mean(dim=1)
    ret = ret / (ret.norm(dim=1, keepdim=True) + 1e-9)
    ret.backward(dLc)
    cpu_s = time.perf_counter() - t0
    line = {
        "metric": "KNN feature smoothing fwd+bwd, 1M Gaussians, 32-D, K=16, 8 columns", "value": round(1e3 / (tf + tb), 2),
        "unit": "iterations/s", "n_gpus": 1, "steps": args.steps, "ms_per_step": round(tf + tb, 4), "dtype": "f32",
        "data": "synthetic (Morton-local neighbour map)", "config": {"workload": f"P={P}, C={C}, K={K}, k={k}"},
        "forward_ms": round(tf, 4), "backward_ms": round(tb, 4),
        "roofline": {"bound": "hbm", "kernel": "knn_smooth fwd+bwd (3 gather kernels + torch allocs)",
                     "achieved": round((bytes_f + bytes_b) / ((tf + tb) * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": round((bytes_f + bytes_b) / ((tf + tb) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "algorithmic_bytes": bytes_f + bytes_b, "traffic": None},
        "cpu_baseline": {"value": round(1.0 / (cpu_s * P / n), 3), "unit": "iterations/s", "cores": torch.get_num_threads(),
                         "kind": "reference", "sample": f"the reference's PyTorch expression fwd+bwd on {n} of {P} rows in {cpu_s:.2f} s, scaled"},
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
