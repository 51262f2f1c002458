"""Import shim for `from simple_knn._C import distCUDA2` (scene/gaussian_model.py:20,
scene/gaussian_model_ff.py:21).  simple-knn is OUT OF SCOPE of the rasterizer hot path (SURVEY.md 2.1
row 6: used only by create_from_pcd); the reference modules import it at load time, so the name must
exist.  This is a plain PyTorch evaluation of the same quantity (mean squared distance to the 3 nearest
neighbours, KNN/simple_knn.cu:147-183), chunked so memory stays bounded."""
import torch


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    P = points.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=points.device)
    pts = points.float()
    chunk = max(1, min(P, (1 << 26) // max(P, 1)))
    for s in range(0, P, chunk):
        d = torch.cdist(pts[s:s + chunk], pts).pow(2)
        k = min(4, P)
        best = torch.topk(d, k, dim=1, largest=False).values[:, 1:]   # drop self (distance 0)
        out[s:s + chunk] = best.sum(1) / 3.0 if k > 1 else 0.0
    return out
