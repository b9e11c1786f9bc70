"""Wire formats around the rasterizer: `cameras.json`, the 3DGS point-cloud PLY and
SuGaR / GauSTAR `{iter}.pt` checkpoints -- enough to render a real checkpoint with this package.

* cameras.json: written by gaussian_splatting/utils/camera_utils.py:70-90 (`camera_to_JSON`), read by
  gaustar_scene/cameras.py:35-78 (`load_gs_cameras`): {id, img_name, width, height, position, rotation, fy, fx};
  position/rotation are the camera-to-world pose.
* PLY: gaussian_splatting/scene/gaussian_model.py:177-250 (`save_ply` / `load_ply`): one `vertex` element of
  float32 properties x y z nx ny nz f_dc_0..2 f_rest_0..(3K-1) opacity scale_0..2 rot_0..3, binary little endian;
  f_rest is stored channel-major ([P,3,K] flattened); opacity is a logit, scales are logs, rot is unnormalised.
* .pt: sugar_model.py:1313-1318 (`save_model`): {'state_dict': ..., extra keys}; state-dict entries `_points`,
  `_surface_mesh_faces`, `_scales`, `_quaternions`, `all_densities`, `_sh_coordinates_dc`, `_sh_coordinates_rest`,
  `surface_mesh_thickness`, `_delta_t`, `_delta_r`.
Pure host code (numpy / torch CPU); no third-party PLY library is needed."""
from __future__ import annotations

import json
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import scene

F32 = np.float32


# ---------------------------------------------------------------------------------------------- cameras.json
def focal2fov(focal: float, pixels: float) -> float:          # gaustar_utils/graphics_utils.py:87-88
    return 2.0 * math.atan(pixels / (2.0 * focal))


def fov2focal(fov: float, pixels: float) -> float:            # graphics_utils.py:84-85
    return pixels / (2.0 * math.tan(fov / 2.0))


def camera_from_json_entry(e: Dict, znear: float = 0.01, zfar: float = 100.0) -> scene.Camera:
    """One cameras.json entry -> the matrices the rasterizer takes, exactly as load_gs_cameras + GSCamera build them
    (cameras.py:55-69, :206-220): R = inv(C2W)[:3,:3]^T, T = inv(C2W)[:3,3]; world_view = getWorld2View2(R, T)^T;
    full_proj = world_view @ getProjectionMatrix(znear, zfar, fovx, fovy)^T; camera centre = inverse(world_view)[3,:3]."""
    c2w = np.zeros((4, 4))
    c2w[:3, :3] = np.array(e["rotation"], dtype=np.float64)
    c2w[:3, 3] = np.array(e["position"], dtype=np.float64)
    c2w[3, 3] = 1.0
    Rt = np.linalg.inv(c2w)
    R, T = Rt[:3, :3].transpose(), Rt[:3, 3]
    W, H = int(e["width"]), int(e["height"])
    fovx, fovy = focal2fov(float(e["fx"]), W), focal2fov(float(e["fy"]), H)
    cam = scene.camera_from_RT(R, T, W, H, fovx, fovy, znear=znear, zfar=zfar)
    cam.name = str(e.get("img_name", ""))
    cam.uid = int(e.get("id", 0))
    return cam


def load_cameras_json(path: str, znear: float = 0.01, zfar: float = 100.0) -> List[scene.Camera]:
    """Sorted by img_name like cameras.py:37."""
    with open(path) as f:
        entries = json.load(f)
    return [camera_from_json_entry(e, znear, zfar) for e in sorted(entries, key=lambda x: x["img_name"])]


def camera_to_json_entry(uid: int, cam: scene.Camera, name: Optional[str] = None) -> Dict:
    """camera_utils.py:70-90, from the transposed world-view matrix this package carries."""
    w2c = np.asarray(cam.viewmatrix, dtype=np.float64).T          # textbook world-to-camera
    c2w = np.linalg.inv(w2c)
    fovx, fovy = 2.0 * math.atan(cam.tanfovx), 2.0 * math.atan(cam.tanfovy)
    return {"id": int(uid), "img_name": name if name is not None else getattr(cam, "name", f"{uid:05d}"),
            "width": int(cam.W), "height": int(cam.H), "position": c2w[:3, 3].tolist(),
            "rotation": [r.tolist() for r in c2w[:3, :3]], "fy": fov2focal(fovy, cam.H), "fx": fov2focal(fovx, cam.W)}


def save_cameras_json(cams: Sequence[scene.Camera], path: str) -> None:
    with open(path, "w") as f:
        json.dump([camera_to_json_entry(i, c) for i, c in enumerate(cams)], f)


# ---------------------------------------------------------------------------------------------- 3DGS PLY
@dataclass
class GaussianCloud:
    """Raw (pre-activation) 3DGS parameters, as stored."""
    xyz: np.ndarray            # [P,3]
    features_dc: np.ndarray    # [P,1,3]
    features_rest: np.ndarray  # [P,K,3]
    opacity: np.ndarray        # [P,1] logits
    scaling: np.ndarray        # [P,3] logs
    rotation: np.ndarray       # [P,4] (w,x,y,z), unnormalised

    @property
    def sh_degree(self) -> int:
        return int(round(math.sqrt(self.features_rest.shape[1] + 1))) - 1

    def rasterizer_inputs(self) -> Dict[str, np.ndarray]:
        """The activations gaussian_renderer/__init__.py:49-70 applies: sigmoid, exp, normalize; shs = cat(dc, rest)."""
        rot = self.rotation / np.maximum(np.linalg.norm(self.rotation, axis=1, keepdims=True), 1e-12)
        return dict(means3D=self.xyz.astype(F32), opacities=(1.0 / (1.0 + np.exp(-self.opacity))).astype(F32),
                    scales=np.exp(self.scaling).astype(F32), rotations=rot.astype(F32),
                    shs=np.concatenate([self.features_dc, self.features_rest], axis=1).astype(F32))


def _ply_properties(K: int) -> List[str]:            # gaussian_model.py:177-190
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * K)] +
            ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])


def save_ply(path: str, cloud: GaussianCloud) -> None:
    P, K = cloud.xyz.shape[0], cloud.features_rest.shape[1]
    f_dc = np.transpose(cloud.features_dc, (0, 2, 1)).reshape(P, -1)       # .transpose(1, 2).flatten(start_dim=1)
    f_rest = np.transpose(cloud.features_rest, (0, 2, 1)).reshape(P, -1)
    attrs = np.concatenate([cloud.xyz, np.zeros_like(cloud.xyz), f_dc, f_rest, cloud.opacity.reshape(P, 1), cloud.scaling,
                            cloud.rotation], axis=1).astype("<f4")
    names = _ply_properties(K)
    assert attrs.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % P
    header += "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(attrs).tobytes())


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2", "int": "<i4",
              "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def load_ply(path: str) -> GaussianCloud:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    raise ValueError(f"{path}: the vertex element must come first")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties are not part of the 3DGS layout")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if count is None:
            raise ValueError(f"{path}: no vertex element")
        if fmt == "binary_little_endian":
            data = np.frombuffer(f.read(count * np.dtype(props).itemsize), dtype=np.dtype(props), count=count)
        elif fmt == "ascii":
            rows = np.loadtxt(f, max_rows=count, ndmin=2)
            data = np.zeros(count, dtype=np.dtype(props))
            for i, (n, _) in enumerate(props):
                data[n] = rows[:, i]
        else:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
    col = lambda n: np.asarray(data[n], dtype=np.float64)
    names = [n for n, _ in props]
    P = count
    xyz = np.stack([col("x"), col("y"), col("z")], axis=1)
    rest_names = sorted((n for n in names if n.startswith("f_rest_")), key=lambda x: int(x.split("_")[-1]))
    if len(rest_names) % 3:
        raise ValueError(f"{path}: {len(rest_names)} f_rest properties is not a multiple of 3")
    K = len(rest_names) // 3
    f_dc = np.stack([col("f_dc_0"), col("f_dc_1"), col("f_dc_2")], axis=1).reshape(P, 3, 1)
    f_rest = (np.stack([col(n) for n in rest_names], axis=1) if K else np.zeros((P, 0))).reshape(P, 3, K)
    scale_names = sorted((n for n in names if n.startswith("scale_")), key=lambda x: int(x.split("_")[-1]))
    rot_names = sorted((n for n in names if n.startswith("rot")), key=lambda x: int(x.split("_")[-1]))
    return GaussianCloud(xyz=xyz.astype(F32), features_dc=np.transpose(f_dc, (0, 2, 1)).astype(F32),
                         features_rest=np.transpose(f_rest, (0, 2, 1)).astype(F32), opacity=col("opacity").reshape(P, 1).astype(F32),
                         scaling=np.stack([col(n) for n in scale_names], axis=1).astype(F32),
                         rotation=np.stack([col(n) for n in rot_names], axis=1).astype(F32))


# ---------------------------------------------------------------------------------------------- SuGaR / GauSTAR .pt
SUGAR_KEYS = ("_points", "_surface_mesh_faces", "_scales", "_quaternions", "all_densities", "_sh_coordinates_dc",
              "_sh_coordinates_rest")


def load_sugar_checkpoint(path: str, map_location="cpu") -> Dict:
    """-> {'verts','faces','raw_scales','raw_complex','densities','sh','thickness','delta_t','delta_r', 'extra'}: the
    arguments of gaustar_amd.producers.mesh_bound_gaussians / points_rgb, from a `{iter}.pt` written by
    SuGaR.save_model (sugar_model.py:1313-1318)."""
    import torch
    # weights_only: a checkpoint is tensors plus plain containers / numbers (state_dict, train_losses, epoch, iteration,
    # optimizer_state_dict); nothing in it needs the unpickler to run code
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    missing = [k for k in SUGAR_KEYS if k not in sd]
    if missing:
        raise KeyError(f"{path}: not a mesh-bound SuGaR state dict, missing {missing}")
    out = dict(verts=sd["_points"].float(), faces=sd["_surface_mesh_faces"].long(), raw_scales=sd["_scales"].float(),
               raw_complex=sd["_quaternions"].float(), densities=sd["all_densities"].float().view(-1, 1),
               sh=torch.cat([sd["_sh_coordinates_dc"], sd["_sh_coordinates_rest"]], dim=1).float(),   # sugar_model.py:449-450
               thickness=float(sd["surface_mesh_thickness"]) if "surface_mesh_thickness" in sd else None,
               delta_t=sd["_delta_t"].float() if "_delta_t" in sd else None,
               delta_r=sd["_delta_r"].float() if "_delta_r" in sd else None,
               extra={k: v for k, v in ckpt.items() if k != "state_dict"} if "state_dict" in ckpt else {})
    if out["raw_complex"].shape[-1] != 2:
        raise ValueError(f"{path}: `_quaternions` has {out['raw_complex'].shape[-1]} columns; mesh-bound models store the 2-D rotation")
    return out


def save_sugar_checkpoint(path: str, verts, faces, raw_scales, raw_complex, densities, sh, thickness, delta_t=None,
                          delta_r=None, **extra) -> None:
    import torch
    sd = {"_points": verts, "_surface_mesh_faces": faces, "_scales": raw_scales, "_quaternions": raw_complex,
          "all_densities": densities.view(-1, 1), "_sh_coordinates_dc": sh[:, :1].contiguous(),
          "_sh_coordinates_rest": sh[:, 1:].contiguous(), "surface_mesh_thickness": torch.tensor(float(thickness))}
    if delta_t is not None:
        sd["_delta_t"] = delta_t
    if delta_r is not None:
        sd["_delta_r"] = delta_r
    ckpt = {"state_dict": {k: (v.detach().cpu() if hasattr(v, "detach") else v) for k, v in sd.items()}}
    ckpt.update(extra)
    torch.save(ckpt, path)
