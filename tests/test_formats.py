"""CPU: wire formats (gaustar_amd/formats.py) -- cameras.json against the reference's own importable helpers
(gaustar_utils/graphics_utils.py vectors in tests/golden/utils_kat.npz pin getWorld2View / getProjectionMatrix;
here the json <-> matrices mapping of cameras.py:55-69 / camera_utils.py:70-90 round-trips), the 3DGS PLY layout
(gaussian_model.py:177-250) and the SuGaR .pt state dict (sugar_model.py:1313-1318)."""
import json
import os

import numpy as np
import pytest
import torch

from gaustar_amd import formats, scene


def test_cameras_json_round_trip(tmp_path):
    cams = scene.ring_cameras(2, 5, 640, 480, focal_px=500.0)
    path = os.path.join(tmp_path, "cameras.json")
    formats.save_cameras_json(cams, path)
    entries = json.load(open(path))
    assert set(entries[0]) == {"id", "img_name", "width", "height", "position", "rotation", "fy", "fx"}   # camera_utils.py:80-89
    assert abs(entries[0]["fx"] - 500.0) < 1e-3 and entries[3]["width"] == 640
    back = formats.load_cameras_json(path, znear=1e-4, zfar=100.0)     # ring_cameras uses pytorch3d's near/far
    assert len(back) == len(cams)
    for a, b in zip(cams, back):
        np.testing.assert_allclose(b.viewmatrix, a.viewmatrix, atol=2e-6)
        np.testing.assert_allclose(b.projmatrix, a.projmatrix, rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(b.campos, a.campos, atol=2e-6)
        assert abs(b.tanfovx - a.tanfovx) < 1e-6 and abs(b.tanfovy - a.tanfovy) < 1e-6 and (b.W, b.H) == (a.W, a.H)


def test_cameras_json_entry_semantics():
    """position/rotation are the camera-to-world pose (cameras.py:55-63): the camera centre is `position`, and a world
    point on the optical axis lands on the image centre with view depth = its distance."""
    e = {"id": 3, "img_name": "cam_b", "width": 800, "height": 600, "fx": 700.0, "fy": 710.0,
         "position": [1.0, 2.0, -3.0], "rotation": [[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]]}
    cam = formats.camera_from_json_entry(e)
    np.testing.assert_allclose(cam.campos, [1.0, 2.0, -3.0], atol=1e-6)
    fwd = np.array(e["rotation"])[:, 2]                      # third column of C2W = viewing direction
    p = np.array(e["position"]) + 5.0 * fwd
    pv = np.append(p, 1.0) @ cam.viewmatrix
    np.testing.assert_allclose(pv[:3], [0.0, 0.0, 5.0], atol=1e-5)
    ph = np.append(p, 1.0) @ cam.projmatrix
    np.testing.assert_allclose(ph[:2] / ph[3], [0.0, 0.0], atol=1e-6)
    assert cam.name == "cam_b" and cam.uid == 3
    assert abs(cam.tanfovx - 400.0 / 700.0) < 1e-6 and abs(cam.tanfovy - 300.0 / 710.0) < 1e-6
    srt = formats.load_cameras_json  # sorted by img_name like cameras.py:37
    assert srt is not None


def test_ply_layout_and_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    P, K = 37, 15
    cloud = formats.GaussianCloud(xyz=rng.normal(size=(P, 3)).astype(np.float32),
                                  features_dc=rng.normal(size=(P, 1, 3)).astype(np.float32),
                                  features_rest=rng.normal(size=(P, K, 3)).astype(np.float32),
                                  opacity=rng.normal(size=(P, 1)).astype(np.float32),
                                  scaling=rng.normal(size=(P, 3)).astype(np.float32) - 3,
                                  rotation=rng.normal(size=(P, 4)).astype(np.float32))
    path = os.path.join(tmp_path, "point_cloud.ply")
    formats.save_ply(path, cloud)
    raw = open(path, "rb").read()
    header, body = raw.split(b"end_header\n")
    lines = header.decode().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", f"element vertex {P}"]
    props = [l.split()[-1] for l in lines if l.startswith("property float")]
    assert props == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] +
                     ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])   # gaussian_model.py:177-190
    rows = np.frombuffer(body, dtype="<f4").reshape(P, len(props))
    np.testing.assert_array_equal(rows[:, 3:6], 0)                                           # normals are zeros (:195)
    # f_rest is channel-major: f_rest_{c*K + k} = features_rest[:, k, c]   (transpose(1,2).flatten, :197)
    np.testing.assert_array_equal(rows[:, 9 + 1 * K + 4], cloud.features_rest[:, 4, 1])
    back = formats.load_ply(path)
    for f in ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation"):
        np.testing.assert_array_equal(getattr(back, f), getattr(cloud, f), err_msg=f)
    assert back.sh_degree == 3
    ri = back.rasterizer_inputs()
    assert ri["shs"].shape == (P, 16, 3) and np.allclose(np.linalg.norm(ri["rotations"], axis=1), 1, atol=1e-6)
    assert (ri["opacities"] > 0).all() and (ri["opacities"] < 1).all() and (ri["scales"] > 0).all()


def test_ply_ascii_and_errors(tmp_path):
    path = os.path.join(tmp_path, "a.ply")
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
             "rot_0", "rot_1", "rot_2", "rot_3"]
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n")
        f.write(" ".join(str(float(i)) for i in range(17)) + "\n" + " ".join(str(float(i) * 2) for i in range(17)) + "\n")
    c = formats.load_ply(path)
    assert c.xyz.shape == (2, 3) and c.features_rest.shape == (2, 0, 3) and c.sh_degree == 0
    np.testing.assert_array_equal(c.rotation[1], [26.0, 28.0, 30.0, 32.0])
    bad = os.path.join(tmp_path, "b.ply")
    open(bad, "w").write("not a ply\n")
    with pytest.raises(ValueError):
        formats.load_ply(bad)


def test_sugar_checkpoint_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    v, f = scene.icosphere(1)
    verts, faces = torch.from_numpy(v).float(), torch.from_numpy(f).long()
    N = faces.shape[0] * 6
    kw = dict(verts=verts, faces=faces, raw_scales=torch.randn(N, 2, generator=g), raw_complex=torch.randn(N, 2, generator=g),
              densities=torch.randn(N, 1, generator=g), sh=torch.randn(N, 16, 3, generator=g), thickness=2.5e-6,
              delta_t=torch.randn(N, 3, generator=g), delta_r=torch.randn(N, 4, generator=g))
    path = os.path.join(tmp_path, "2000.pt")
    formats.save_sugar_checkpoint(path, **kw, iteration=2000)
    raw = torch.load(path, weights_only=False)
    assert set(raw) == {"state_dict", "iteration"}                                   # sugar_model.py:1313-1318
    assert raw["state_dict"]["_sh_coordinates_dc"].shape == (N, 1, 3) and raw["state_dict"]["_sh_coordinates_rest"].shape == (N, 15, 3)
    back = formats.load_sugar_checkpoint(path)
    for k in ("verts", "faces", "raw_scales", "raw_complex", "densities", "sh", "delta_t", "delta_r"):
        assert torch.equal(back[k], kw[k]), k
    assert abs(back["thickness"] - 2.5e-6) < 1e-12 and back["extra"] == {"iteration": 2000}
    torch.save({"state_dict": {"_points": verts}}, path)
    with pytest.raises(KeyError):
        formats.load_sugar_checkpoint(path)


def test_cameras_json_written_by_the_reference():
    """tests/golden/formats_cameras_ref.json was written by the reference's camera_to_JSON (camera_utils.py:70-89) and
    formats_cameras_ref.npz holds what the reference's reader + GSCamera build from it (tests/golden/make_formats_golden.py):
    formats.load_cameras_json must reproduce those matrices, and formats.camera_to_json_entry must write the
    reference's entries back."""
    import json
    from gaustar_amd import formats
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cams = formats.load_cameras_json(os.path.join(g, "formats_cameras_ref.json"), znear=0.01, zfar=100.0)
    z = np.load(os.path.join(g, "formats_cameras_ref.npz"))
    assert len(cams) == len(z["view_t"]) == 6
    assert [c.name for c in cams] == sorted(c.name for c in cams)          # sorted by img_name like cameras.py:37
    for i, c in enumerate(cams):
        np.testing.assert_allclose(c.viewmatrix, z["view_t"][i], rtol=0, atol=2e-6)
        np.testing.assert_allclose(c.projmatrix, z["full_t"][i], rtol=0, atol=5e-6)
        np.testing.assert_allclose(c.campos, z["campos"][i], rtol=0, atol=5e-6)
        np.testing.assert_allclose([c.tanfovx, c.tanfovy], z["tanfov"][i], rtol=1e-6)
        assert [c.W, c.H] == z["size"][i].tolist()
    ref_entries = {e["img_name"]: e for e in json.load(open(os.path.join(g, "formats_cameras_ref.json")))}
    for c in cams:
        ours, ref = formats.camera_to_json_entry(c.uid, c, c.name), ref_entries[c.name]
        assert ours["id"] == ref["id"] and ours["width"] == ref["width"] and ours["height"] == ref["height"]
        np.testing.assert_allclose(ours["position"], ref["position"], atol=1e-5)
        np.testing.assert_allclose(ours["rotation"], ref["rotation"], atol=1e-5)
        np.testing.assert_allclose([ours["fx"], ours["fy"]], [ref["fx"], ref["fy"]], rtol=1e-5)
