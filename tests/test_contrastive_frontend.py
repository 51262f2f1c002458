"""the ray-first contrastive front end equals the reference expression
(train_contrastive_feature.py:237-253), values and gradients."""
import pytest
import torch

from seganygaussians_amd.contrastive_frontend import sample_scale_conditioned_features


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
