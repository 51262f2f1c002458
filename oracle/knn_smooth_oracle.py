"""CPU oracle of the KNN feature smoothing -- TEST INFRASTRUCTURE ONLY.

Only tests/ (and tools/bench_knn_smooth.py's cpu_baseline leg) may import this module; the product path is the HIP
kernels behind include/mi_knn_smooth.h and has no CPU fallback.

Restates, in numpy float64, FeatureGaussianModel.get_smoothed_point_features
(/root/reference/scene/gaussian_model_ff.py:338-364) and the renderer's re-normalisation
(/root/reference/gaussian_renderer/__init__.py:362-363), with the analytic backward.  Pinned in
tests/test_knn_smooth.py against the reference's own expression evaluated by PyTorch autograd on the CPU (the
reference implementation of this row IS plain PyTorch, so that pin is the reference itself)."""
import numpy as np


def forward(F, idx, cols, normalize_out=True):
    """F [P,C], idx [P,K] int, cols: iterable of neighbour columns.  Returns out [P,C] (float64)."""
    F = np.asarray(F, np.float64)
    cols = np.asarray(list(cols), np.int64)
    nrm = np.maximum(np.linalg.norm(F, axis=1, keepdims=True), 1e-12)   # F.normalize(p=2, eps=1e-12)
    n = F / nrm
    m = n[np.asarray(idx)[:, cols], :].mean(axis=1)                      # gaussian_model_ff.py:356-362
    if normalize_out:
        m = m / (np.linalg.norm(m, axis=1, keepdims=True) + 1e-9)        # gaussian_renderer/__init__.py:362-363
    return m


def backward(F, idx, cols, g, normalize_out=True):
    """dL/dF for upstream gradient g [P,C] (float64)."""
    F = np.asarray(F, np.float64)
    g = np.asarray(g, np.float64)
    idx = np.asarray(idx)
    cols = np.asarray(list(cols), np.int64)
    P, C = F.shape
    k = len(cols)
    raw = np.linalg.norm(F, axis=1, keepdims=True)
    nrm = np.maximum(raw, 1e-12)
    n = F / nrm
    m = n[idx[:, cols], :].mean(axis=1)
    if normalize_out:
        r = np.linalg.norm(m, axis=1, keepdims=True)
        a = 1.0 / (r + 1e-9)
        mg = (m * g).sum(axis=1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            b = np.where(r > 0, mg * a * a / r, 0.0)
        dm = g * a - m * b
    else:
        dm = g
    dm = dm / k
    dn = np.zeros_like(F)
    for c in cols:                                                       # index_put with accumulation
        np.add.at(dn, idx[:, c], dm)
    inside = (raw >= 1e-12)
    dF = np.where(inside, (dn - n * (n * dn).sum(axis=1, keepdims=True)) / nrm, dn / 1e-12)
    return dF
