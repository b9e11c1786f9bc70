"""Generate tests/golden/formats_cameras_ref.json + formats_cameras_ref.npz by IMPORTING the reference's own camera code
(runs only where /root/reference exists; the fixtures -- data only -- are committed):

  * formats_cameras_ref.json is written by the reference's writer, camera_to_JSON
    (gaussian_splatting/utils/camera_utils.py:70-89), from duck-typed cameras (R, T, FovX, FovY, width, height,
    image_name -- the attributes it reads);
  * formats_cameras_ref.npz holds, per camera, the matrices the reference's reader side builds from such an entry
    (gaustar_scene/cameras.py:55-69 -> GSCamera, gaussian_splatting/scene/cameras.py:56-59): world_view =
    getWorld2View2(R, T)^T, full_proj = world_view @ getProjectionMatrix(0.01, 100, fovx, fovy)^T, camera centre --
    computed with the reference's getWorld2View2 / getProjectionMatrix / focal2fov (utils/graphics_utils.py).

`scene/__init__.py` of the vendored 3DGS pulls in plyfile (absent here), so the `scene` package is entered without
running its __init__: scene.cameras and utils.camera_utils themselves need torch / numpy / PIL only.

    python tests/golden/make_formats_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("GSR_REFERENCE_ROOT", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    gs_root = os.path.join(REF, "gaussian_splatting")
    sys.path.insert(0, gs_root)
    pkg = types.ModuleType("scene")
    pkg.__path__ = [os.path.join(gs_root, "scene")]
    sys.modules["scene"] = pkg
    from utils.camera_utils import camera_to_JSON
    from utils.graphics_utils import focal2fov, getProjectionMatrix, getWorld2View2

    rng = np.random.default_rng(7)
    entries, exp = [], dict(view_t=[], full_t=[], campos=[], tanfov=[], size=[])
    for i in range(6):
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1

        class Duck:   # the attributes camera_to_JSON reads
            R = q
            T = rng.normal(size=3) * 2.0
            FovX = float(rng.uniform(0.4, 1.3))
            FovY = float(rng.uniform(0.3, 1.0))
            width = int(rng.integers(300, 2000))
            height = int(rng.integers(200, 1200))
            image_name = f"img_{(7 * i) % 6:04d}"     # not in id order: the reader sorts by name (cameras.py:37)
        entries.append(camera_to_JSON(i, Duck))
    with open(os.path.join(HERE, "formats_cameras_ref.json"), "w") as f:
        json.dump(entries, f)
    # the reader side (cameras.py:35-78), sorted by img_name
    for e in sorted(entries, key=lambda x: x["img_name"]):
        W2C = np.zeros((4, 4))
        W2C[:3, :3] = np.array(e["rotation"]); W2C[:3, 3] = np.array(e["position"]); W2C[3, 3] = 1
        Rt = np.linalg.inv(W2C)
        T, R = Rt[:3, 3], Rt[:3, :3].transpose()
        fovy, fovx = focal2fov(e["fy"], e["height"]), focal2fov(e["fx"], e["width"])
        view_t = torch.tensor(getWorld2View2(R, T)).transpose(0, 1)
        proj_t = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full_t = (view_t.unsqueeze(0).bmm(proj_t.unsqueeze(0))).squeeze(0)
        exp["view_t"].append(view_t.numpy()); exp["full_t"].append(full_t.numpy())
        exp["campos"].append(view_t.inverse()[3, :3].numpy())
        exp["tanfov"].append([np.tan(fovx * 0.5), np.tan(fovy * 0.5)]); exp["size"].append([e["width"], e["height"]])
    np.savez(os.path.join(HERE, "formats_cameras_ref.npz"), **{k: np.asarray(v) for k, v in exp.items()})
    print("wrote formats_cameras_ref.json / formats_cameras_ref.npz:", len(entries), "cameras")


if __name__ == "__main__":
    main()
