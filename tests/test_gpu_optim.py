"""GPU: gaustar_amd.optim.Adam against torch.optim.Adam itself -- the optimiser GauSTAR uses
(gaustar_scene/sugar_optimizer.py:87: per-group learning rates, lr = 0 default, eps = 1e-15)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _groups(seed, dev):
    g = torch.Generator().manual_seed(seed)
    shapes = [(4099, 3), (2051, 1, 3), (2051, 15, 3), (2051, 1), (2051, 2), (7,)]       # odd sizes: vector body + scalar tail
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3]
    ps = [torch.randn(*s, generator=g).to(dev).requires_grad_(True) for s in shapes]
    return ps, [{"params": [p], "lr": lr, "name": f"g{i}"} for i, (p, lr) in enumerate(zip(ps, lrs))]


def test_matches_torch_adam_over_many_steps(hip_lib):
    from gaustar_amd import optim
    dev = torch.device("cuda:0")
    pa, ga = _groups(3, dev)
    pb, gb = _groups(3, dev)
    a = optim.Adam(ga, lr=0.0, eps=1e-15)
    b = torch.optim.Adam(gb, lr=0.0, eps=1e-15, foreach=False, fused=False)
    gen = torch.Generator(device=dev).manual_seed(11)
    cpu_gen = torch.Generator().manual_seed(12)
    for it in range(60):
        for grp_a, grp_b in zip(a.param_groups, b.param_groups):       # the trainer rewrites learning rates every iteration
            grp_a["lr"] = grp_b["lr"] = grp_b["lr"] * 0.99
        for x, y in zip(pa, pb):
            if it % 7 == 3 and x.dim() == 1:
                x.grad = y.grad = None                                  # a parameter without gradient is skipped, state untouched
                continue
            g = torch.randn(x.shape, device=dev, generator=gen) * (10.0 ** float(torch.randint(-6, 2, (1,), generator=cpu_gen).item()))
            x.grad, y.grad = g.clone(), g.clone()
        a.step(); b.step()
    for (x, y), grp in zip(zip(pa, pb), b.param_groups):
        # each step moves a parameter by at most ~lr; the two implementations round that move differently by a few ulp
        # (PyTorch's kernels are built with FMA contraction, adam_kernel without): 60 steps x lr x 2^-23 x a few
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-6,
                                   atol=max(1e-7, 60 * 1.0e-6 * (grp["lr"] / 0.99 ** 60)))
        sa, sb = a.state[x], b.state[y]
        assert float(sa["step"]) == float(sb["step"])
        # moments: gradients of very different magnitude were averaged in, so an element's rounding error scales with the
        # largest term it has seen, not with its current (possibly cancelled) value
        for key in ("exp_avg", "exp_avg_sq"):
            ra, rb = sa[key].cpu().numpy(), sb[key].cpu().numpy()
            np.testing.assert_allclose(ra, rb, rtol=2e-6, atol=2e-6 * float(np.abs(rb).max()))


def test_state_dict_round_trip_with_torch_adam(hip_lib):
    from gaustar_amd import optim
    dev = torch.device("cuda:0")
    pa, ga = _groups(5, dev)
    pb, gb = _groups(5, dev)
    a = optim.Adam(ga, lr=0.0, eps=1e-15)
    b = torch.optim.Adam(gb, lr=0.0, eps=1e-15, foreach=False)
    for x in pa:
        x.grad = torch.ones_like(x)
    a.step()
    import copy
    b.load_state_dict(copy.deepcopy(a.state_dict()))   # continue a run under the other implementation (load_state_dict keeps
                                                       # tensors that already match dtype/device: without the copy both would
                                                       # update the SAME moment tensors)
    for x, y in zip(pa, pb):
        with torch.no_grad():
            y.copy_(x)
        x.grad = torch.full_like(x, 0.5); y.grad = torch.full_like(y, 0.5)
    a.step(); b.step()
    for x, y in zip(pa, pb):
        np.testing.assert_allclose(x.detach().cpu().numpy(), y.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)


def test_validation(hip_lib):
    from gaustar_amd import optim
    with pytest.raises(NotImplementedError):
        optim.Adam([torch.zeros(3, device="cuda", requires_grad=True)], weight_decay=0.1)
    p = torch.zeros(3, requires_grad=True)       # CPU tensor: no CPU path
    o = optim.Adam([p], lr=1e-3)
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        o.step()


def test_one_launch_for_all_tensors_equals_one_launch_per_tensor(hip_lib):
    """gsr_adam_step_multi (the default: every tensor of a step in one launch, per-tensor learning rates in the kernel
    arguments) against gsr_adam_step per tensor: bit-identical parameters and moments, including tensors that skip a step
    (their bias correction then differs from the others': a launch of their own) and more than 16 tensors (two launches)."""
    from gaustar_amd import optim
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    shapes = [(4099, 3), (2051, 1, 3), (2051, 15, 3), (2051, 1), (7,), (1,), (260_000, 4)] + [(33 + i, 5) for i in range(14)]
    mk = lambda: [torch.randn(*s, generator=torch.Generator().manual_seed(100 + i)).to(dev).requires_grad_(True) for i, s in enumerate(shapes)]
    pa, pb = mk(), mk()
    grp = lambda ps: [{"params": [p], "lr": 1e-3 * (1 + i)} for i, p in enumerate(ps)]
    a, b = optim.Adam(grp(pa), eps=1e-15), optim.Adam(grp(pb), eps=1e-15)
    b.multi_tensor = False
    gen = torch.Generator(device=dev).manual_seed(4)
    for it in range(7):
        for i, (x, y) in enumerate(zip(pa, pb)):
            if it == 2 and i in (1, 4):
                x.grad = y.grad = None            # skipped once: their step count lags behind from here on
                continue
            gr = torch.randn(x.shape, device=dev, generator=gen)
            x.grad, y.grad = gr.clone(), gr.clone()
        a.step(); b.step()
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
        assert torch.equal(a.state[x]["exp_avg"], b.state[y]["exp_avg"]) and torch.equal(a.state[x]["exp_avg_sq"], b.state[y]["exp_avg_sq"])
        assert float(a.state[x]["step"]) == float(b.state[y]["step"])
