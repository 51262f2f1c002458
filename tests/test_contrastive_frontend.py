"""the ray-first contrastive front end equals the reference expression
(train_contrastive_feature.py:237-253), values and gradients."""
import pytest
import torch

from seganygaussians_amd.contrastive_frontend import contrastive_front_end, sample_scale_conditioned_features


def _reference(rendered, out_hw, sampled_ray, gates):
    r = torch.nn.functional.interpolate(rendered.unsqueeze(0), out_hw, mode='bilinear').squeeze(0)
    fws = r.unsqueeze(0).repeat([gates.shape[0], 1, 1, 1])
    fws = fws * gates.unsqueeze(-1).unsqueeze(-1)
    s = fws[:, :, sampled_ray].permute([0, 2, 1])
    return torch.nn.functional.normalize(s, dim=-1, p=2)


@pytest.mark.parametrize("shape", [((32, 45, 80), (90, 160)), ((32, 37, 53), (111, 200)), ((8, 64, 64), (64, 64)),
                                   ((16, 120, 90), (60, 45))])
def test_matches_reference_expression(shape):
    (C, h, w), (H, W) = shape
    g = torch.Generator().manual_seed(C + h)
    rendered = torch.randn(C, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    gates = torch.rand(10, C, generator=g, dtype=torch.float64, requires_grad=True)
    sampled_ray = torch.rand(H, W, generator=g) < 0.05
    up = torch.randn(10, int(sampled_ray.sum()), C, generator=g, dtype=torch.float64)
    want = _reference(rendered, (H, W), sampled_ray, gates)
    gw = torch.autograd.grad(want, [rendered, gates], up)
    got = sample_scale_conditioned_features(rendered, (H, W), sampled_ray, gates)
    gg = torch.autograd.grad(got, [rendered, gates], up)
    torch.testing.assert_close(got, want, rtol=1e-12, atol=1e-13)
    torch.testing.assert_close(gg[0], gw[0], rtol=1e-10, atol=1e-12)
    torch.testing.assert_close(gg[1], gw[1], rtol=1e-10, atol=1e-12)


def test_shape_check():
    with pytest.raises(ValueError):
        sample_scale_conditioned_features(torch.zeros(4, 8, 8), (16, 16), torch.zeros(8, 8, dtype=torch.bool), torch.ones(2, 4))


@pytest.mark.gpu
def test_gpu_full_size_memory_and_time():
    """SURVEY 8(f) row 3 on the MI355X at the sizes of train_contrastive_feature.py: a (32, 1080, 1920) feature render,
    10 sampled scales, 1000 sampled rays.  Values and gradients equal the reference expression (fp32, same device);
    peak memory and time of both are measured: the reference materialises the (10, 32, 1080, 1920) tensor (2.65 GB, twice:
    repeat + product, and again in the backward), the ray-first form touches 1000 x 4 taps."""
    dev = "cuda:0"
    C, H, W, N, S = 32, 1080, 1920, 10, 1000
    g = torch.Generator().manual_seed(0)
    rendered = torch.randn(C, H, W, generator=g).to(dev).requires_grad_(True)
    gates = torch.rand(N, C, generator=g).to(dev).requires_grad_(True)
    ray_idx = torch.randperm(H * W, generator=g)[:S]
    sampled_ray = torch.zeros(H * W, dtype=torch.bool)
    sampled_ray[ray_idx] = True
    sampled_ray = sampled_ray.view(H, W).to(dev)
    up = torch.randn(N, S, C, generator=g).to(dev)

    def run(fn):
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(rendered, (H, W), sampled_ray, gates)
        grads = torch.autograd.grad(out, [rendered, gates], up)
        e1.record()
        torch.cuda.synchronize()
        return out.detach(), grads, (torch.cuda.max_memory_allocated() - base) / 2**20, e0.elapsed_time(e1)

    run(sample_scale_conditioned_features)   # warm-up
    got, gg, mem_new, ms_new = run(sample_scale_conditioned_features)
    want, gw, mem_ref, ms_ref = run(_reference)
    print(f"contrastive front end at 1080p x {N} scales x {S} rays: ray-first {ms_new:.2f} ms, peak +{mem_new:.0f} MiB; "
          f"reference expression {ms_ref:.2f} ms, peak +{mem_ref:.0f} MiB")
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(gg[1], gw[1], rtol=1e-4, atol=1e-5)
    # the image gradient is 1000 x 4 taps of a 66 M element tensor: compare where it is non-zero, and the zero pattern
    nz = gw[0] != 0
    assert torch.equal(nz, gg[0] != 0)
    torch.testing.assert_close(gg[0][nz], gw[0][nz], rtol=1e-4, atol=1e-6)
    assert mem_new < 0.2 * mem_ref and mem_ref > 5000


def _reference_with_norm(rendered, out_hw, sampled_ray, gates):
    """train_contrastive_feature.py:234-254 verbatim: the regulariser's norm on the un-resized render, then the ray features."""
    return _reference(rendered, out_hw, sampled_ray, gates), rendered.norm(dim=0, p=2).mean()


def _hip_case(C, h, w, H, W, N, S, seed, zero_pixel=False):
    dev = "cuda:0"
    g = torch.Generator().manual_seed(seed)
    rendered = torch.randn(C, h, w, generator=g)
    if zero_pixel:
        rendered[:, 0, 0] = 0.0          # ||f|| = 0: torch's norm backward gives 0 there
    rendered = rendered.to(dev).requires_grad_(True)
    gates = torch.rand(N, C, generator=g).to(dev).requires_grad_(True)
    sampled_ray = torch.zeros(H * W, dtype=torch.bool)
    sampled_ray[torch.randperm(H * W, generator=g)[:S]] = True
    sampled_ray = sampled_ray.view(H, W).to(dev)
    up = torch.randn(N, S, C, generator=g).to(dev)
    gn = torch.tensor(0.37, device=dev)
    return rendered, gates, sampled_ray, up, gn


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(7, 33, 51, 66, 102, 3, 40), (32, 64, 96, 64, 96, 10, 300), (80, 31, 45, 17, 29, 2, 25)])
def test_hip_front_end_small(dims):
    """The HIP front end (include/mi_contrastive.h) against the reference expression: odd sizes (no 4-pixel vector path),
    identity resize, down-sampling, C > 64, a pixel with zero norm."""
    C, h, w, H, W, N, S = dims
    rendered, gates, sampled_ray, up, gn = _hip_case(C, h, w, H, W, N, S, seed=C, zero_pixel=True)
    want, want_n = _reference_with_norm(rendered, (H, W), sampled_ray, gates)
    gw = torch.autograd.grad([want, want_n], [rendered, gates], [up, gn], retain_graph=True)
    gw_dense = torch.autograd.grad([want_n], [rendered], [gn])[0]          # the regulariser term alone
    got, got_n = contrastive_front_end(rendered, (H, W), sampled_ray, gates)
    gg = torch.autograd.grad([got, got_n], [rendered, gates], [up, gn], retain_graph=True)
    gg_dense = torch.autograd.grad([got, got_n], [rendered], [torch.zeros_like(up), gn])[0]
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got_n, want_n, rtol=1e-6, atol=0)
    torch.testing.assert_close(gg[0], gw[0], rtol=1e-4, atol=1e-6 * float(gw[0].abs().max()))
    torch.testing.assert_close(gg[1], gw[1], rtol=1e-4, atol=1e-5 * float(gw[1].abs().max()))
    torch.testing.assert_close(gg_dense, gw_dense, rtol=1e-5, atol=1e-7 * float(gw_dense.abs().max()))
    assert torch.isfinite(gg[0]).all() and float(gg_dense[:, 0, 0].abs().max()) == 0.0


@pytest.mark.gpu
def test_hip_front_end_full_size_parity_and_time():
    """32 x 1080p render, 10 scales, 1000 rays: values, the regulariser and both gradients against the reference expression on the
    same device; time of forward + backward (HIP events) against the three 265-MB streams the kernels move."""
    C, H, W, N, S = 32, 1080, 1920, 10, 1000
    rendered, gates, sampled_ray, up, gn = _hip_case(C, H, W, H, W, N, S, seed=0)

    def run(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out, nrm = fn(rendered, (H, W), sampled_ray, gates)
        grads = torch.autograd.grad([out, nrm], [rendered, gates], [up, gn])
        e1.record()
        torch.cuda.synchronize()
        return out.detach(), nrm.detach(), grads, e0.elapsed_time(e1)

    ray_yx = torch.nonzero(sampled_ray).to(torch.int32)   # (the boolean-mask -> coordinates step syncs the host: outside the timing)
    hip = lambda r, hw, m, g: contrastive_front_end(r, hw, ray_yx, g)
    run(hip)
    best = min(run(hip)[3] for _ in range(5))
    got, got_n, gg, _ = run(hip)
    want, want_n, gw, ms_ref = run(_reference_with_norm)
    gbps = 3 * 4 * C * H * W / (best * 1e-3) / 1e9
    print(f"contrastive front end (HIP) at 32 x 1080p x {N} scales x {S} rays: fwd+bwd {best:.3f} ms = {gbps:.0f} GB/s of "
          f"algorithmic bytes ({gbps / 8000:.2f} of 8 TB/s); reference expression {ms_ref:.1f} ms")
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(got_n, want_n, rtol=1e-6, atol=0)
    torch.testing.assert_close(gg[1], gw[1], rtol=1e-4, atol=1e-5 * float(gw[1].abs().max()))
    torch.testing.assert_close(gg[0], gw[0], rtol=1e-4, atol=1e-6 * float(gw[0].abs().max()))
    assert best < 0.5, best   # three streams of 265 MB: 0.1 ms at 8 TB/s


def test_hip_front_end_needs_gpu_tensors():
    with pytest.raises(RuntimeError, match="GPU tensors"):
        contrastive_front_end(torch.zeros(4, 8, 8), (8, 8), torch.zeros(8, 8, dtype=torch.bool), torch.ones(2, 4))
