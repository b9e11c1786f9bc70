"""Host-side mirror of the reference's Python API (DGR/diff_gaussian_rasterization/__init__.py):
names, fields, argument checks and error behaviour.  CPU only."""
import numpy as np
import pytest
import torch


def test_drop_in_import_names():
    import diff_gaussian_rasterization as dgr
    from gaustar_amd import GaussianRasterizationSettings, GaussianRasterizer
    assert dgr.GaussianRasterizer is GaussianRasterizer
    assert dgr.GaussianRasterizationSettings is GaussianRasterizationSettings
    assert callable(dgr.rasterize_gaussians)
    # ref :157-169 -- same 12 fields, same order
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")


def _settings():
    from gaustar_amd import GaussianRasterizationSettings
    return GaussianRasterizationSettings(32, 32, np.float64(0.5), np.float32(0.5), torch.zeros(3), 1.0, torch.eye(4),
                                         torch.eye(4), 0, torch.zeros(1, 3), False, False)


def test_optional_argument_checks_raise_like_the_reference():
    from gaustar_amd import GaussianRasterizer
    r = GaussianRasterizer(_settings())
    m, m2, o = torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 1)
    s, q, c, sh = torch.ones(4, 3), torch.ones(4, 4), torch.ones(4, 3), torch.ones(4, 1, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m2, o, scales=s, rotations=q)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m2, o, shs=sh, colors_precomp=c, scales=s, rotations=q)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m2, o, colors_precomp=c, scales=s)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(m, m2, o, colors_precomp=c, scales=s, rotations=q, cov3D_precomp=torch.ones(4, 6))


def test_no_cpu_fallback_and_shape_check():
    from gaustar_amd import GaussianRasterizer
    r = GaussianRasterizer(_settings())
    m, m2, o = torch.zeros(4, 3), torch.zeros(4, 3), torch.zeros(4, 1)
    s, q, c = torch.ones(4, 3), torch.ones(4, 4), torch.ones(4, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(m, m2, o, colors_precomp=c, scales=s, rotations=q)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.markVisible(m)
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        r(torch.zeros(4, 2), m2, o, colors_precomp=c, scales=s, rotations=q)


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from gaustar_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_the_oracle():
    import os
    import re
    from conftest import ROOT
    pkg = os.path.join(ROOT, "gaustar_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "gsr_oracle" not in src and "libgsr_ref" not in src, f


def test_model_copies_leave_the_gradient_sink_behind():
    """harness.SurfaceGaussians on CPU tensors (construction launches nothing): deepcopy / pickle keep parameters and buffers
    and drop the gradient sink and the per-object caches (an optimiser with hooks on the ORIGINAL's parameters must not be
    cloned along); the thickness is read back once per value."""
    import copy
    import pickle
    import torch
    from gaustar_amd import harness, scene
    v, f = scene.icosphere(1, 1.0, (0.0, 0.0, 0.0))
    m = harness.SurfaceGaussians(torch.from_numpy(v).float(), torch.from_numpy(f).long(), 3, 2, surface_mesh_thickness=2e-6,
                                 loose_bind=True)
    assert m._thickness() == 2e-6 or abs(m._thickness() - 2e-6) < 1e-12
    m.grad_sink = object()
    for c in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert c.grad_sink is None and c._geom_cache is None and c._thickness_cache is None
        assert all(torch.equal(a, b) for a, b in zip(c.state_dict().values(), m.state_dict().values()))
        assert c._points is not m._points
    assert m.grad_sink is not None
    assert [tuple(p.shape) for p in m.grad_ready_order()][-1] == tuple(m._points.shape)


def test_camera_key_of_a_host_matrix_is_its_bytes_and_is_cached_by_tensor_identity():
    """The plan key is a function of the view matrix's CONTENTS (the reference's caller builds a fresh tensor per render call,
    gaustar_scene/sugar_model.py:1149-1150): a host matrix is keyed by its bytes without touching the device; the same tensor
    object is not looked at twice until it is written to; an entry disappears with its tensor."""
    import gc
    from gaustar_amd import rasterizer as rz
    dev = torch.device("cuda", 0)
    a = torch.eye(4)
    b = torch.eye(4).t()                      # other object, other strides, same contents
    c = torch.eye(4) * 2.0
    s0 = dict(rz.CAMERA_KEY_STATS)
    ka, kb, kc = (rz._camera_key(None, x, dev) for x in (a, b, c))
    assert ka == kb and ka != kc and ka[0] == "host"
    assert rz._camera_key(None, a, dev) == ka
    assert rz.CAMERA_KEY_STATS["known_tensor"] - s0["known_tensor"] == 1
    a.mul_(3.0)                               # in place: the version counter moves
    assert rz._camera_key(None, a, dev) != ka
    assert rz._camera_key(None, torch.zeros(3, 4), dev) is None and rz._camera_key(None, torch.eye(4).double(), dev) is None
    n = len(rz._CAM_KEYS)
    ia = id(a)
    del a
    gc.collect()
    assert ia not in rz._CAM_KEYS and len(rz._CAM_KEYS) == n - 1
