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
