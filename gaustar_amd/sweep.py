"""Forward-only camera sweeps: the 160-camera loops of
gaustar_trainers/refined_mesh.py:729-775 (`detect_topo_err`) and :1083-1134 (final renders) -- every camera renders
RGB and depth-as-colour of the same Gaussians, results are reduced to small per-view rows.

Here each camera is ONE 4-channel forward (RGB + depth share preprocess / binning / sort / blend),
under `torch.no_grad()`, and the cameras are sharded over the ranks of a `torch.distributed` job (one process per
GPU; `nccl` = RCCL on ROCm, `gloo` for the CPU tests): rank r renders cameras r, r + world, r + 2 world, ...; the
per-view rows are brought together with one all_gather.  There is no other communication."""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as tdist

from . import dist as gdist


def camera_shard(n_cameras: int, rank: Optional[int] = None, world: Optional[int] = None) -> List[int]:
    """Strided shard: neighbouring cameras of a rig see similar amounts of surface, so striding balances the ranks."""
    rank = gdist.rank() if rank is None else rank
    world = gdist.world_size() if world is None else world
    return list(range(rank, n_cameras, world))


def gather_rows(local_rows: torch.Tensor, n_total: int, rank: Optional[int] = None, world: Optional[int] = None) -> torch.Tensor:
    """local_rows [len(camera_shard(n_total)), K] on every rank -> [n_total, K] on every rank, row i = camera i."""
    rank = gdist.rank() if rank is None else rank
    world = gdist.world_size() if world is None else world
    if local_rows.dim() == 1:
        local_rows = local_rows[:, None]
    mine = camera_shard(n_total, rank, world)
    if local_rows.size(0) != len(mine):
        raise ValueError(f"rank {rank} owns {len(mine)} cameras but passed {local_rows.size(0)} rows")
    if world == 1:
        return local_rows
    per = (n_total + world - 1) // world
    pad = torch.zeros(per, local_rows.size(1), dtype=local_rows.dtype, device=local_rows.device)
    pad[:len(mine)] = local_rows
    parts = [torch.empty_like(pad) for _ in range(world)]
    tdist.all_gather(parts, pad)
    out = torch.empty(n_total, local_rows.size(1), dtype=local_rows.dtype, device=local_rows.device)
    for r in range(world):
        idx = camera_shard(n_total, r, world)
        out[idx] = parts[r][:len(idx)]
    return out


class ForwardSweep:
    """Holds one set of Gaussians on the device and renders it from many cameras without autograd.

    means3D [P,3], opacities [P,1], scales [P,3], rotations [P,4] as the rasterizer takes them; colours either
    `rgb` [P,3] precomputed or (`sh` [P,K,3], `sh_levels`) evaluated per camera by producers.points_rgb."""

    def __init__(self, means3D, opacities, scales, rotations, rgb=None, sh=None, sh_levels: int = 1, max_depth: float = 10.0,
                 bg_rgb=(0.0, 1.0, 0.0)):
        if (rgb is None) == (sh is None):
            raise ValueError("provide exactly one of rgb / sh")
        self.means3D, self.opacities, self.scales, self.rotations = means3D, opacities, scales, rotations
        self.rgb, self.sh, self.sh_levels = rgb, sh, int(sh_levels)
        self.max_depth = float(max_depth)
        dev = means3D.device
        self.bg4 = torch.tensor(list(bg_rgb) + [self.max_depth], dtype=torch.float32, device=dev)   # RGB + one depth channel
        self.bg3 = torch.full((3,), self.max_depth, dtype=torch.float32, device=dev)
        self._cam_cache = {}

    def _cam(self, cam):
        # per-camera matrices are uploaded once (the reference re-uploads per call).  Keyed by the matrices' bytes, not by
        # id(cam): ids of transient camera objects are reused after collection, and an edited camera must not hit.
        key = (np.asarray(cam.viewmatrix, np.float32).tobytes(), np.asarray(cam.projmatrix, np.float32).tobytes(),
               np.asarray(cam.campos, np.float32).tobytes())
        if key not in self._cam_cache:
            dev = self.means3D.device
            t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
            self._cam_cache[key] = (t(cam.viewmatrix), t(cam.projmatrix), t(cam.campos))
        return self._cam_cache[key]

    @torch.no_grad()
    def render_rgb_depth(self, cam):
        """-> (rgb [H,W,3], depth [H,W]): refined_mesh.py:733-760 in one pass."""
        from . import GaussianRasterizationSettings, GaussianRasterizer, producers
        view, proj, campos = self._cam(cam)
        if self.rgb is None:   # SH colours and view-space depth from one fused producer
            colors4 = producers.points_rgb_depth(self.means3D, campos, self.sh, self.sh_levels, view, depth_channels=1)
        else:
            colors4 = torch.cat([self.rgb, self.means3D @ view[:3, 2:3] + view[3, 2]], 1)
        s = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, self.bg4, 1.0, view, proj, 0, campos, False, False)
        img, _ = GaussianRasterizer(s)(means3D=self.means3D, means2D=self.means3D, opacities=self.opacities,   # means2D is never read
                                       colors_precomp=colors4, scales=self.scales, rotations=self.rotations)
        return img[:3].permute(1, 2, 0), img[3]

    @torch.no_grad()
    def render_depth(self, cam, scales=None):
        """Depth-as-colour alone, optionally with other scales (the `use_solid_surface` pass, refined_mesh.py:762-772)."""
        from . import GaussianRasterizationSettings, GaussianRasterizer
        view, proj, campos = self._cam(cam)
        depth = (self.means3D @ view[:3, 2:3] + view[3, 2]).expand(-1, 3).contiguous()
        s = GaussianRasterizationSettings(cam.H, cam.W, cam.tanfovx, cam.tanfovy, self.bg3, 1.0, view, proj, 0, campos, False, False)
        img, _ = GaussianRasterizer(s)(means3D=self.means3D, means2D=torch.zeros_like(self.means3D), opacities=self.opacities,
                                       colors_precomp=depth, scales=self.scales if scales is None else scales,
                                       rotations=self.rotations)
        return img[0]

    def sweep(self, cameras: Sequence, per_view: Callable, rank: Optional[int] = None, world: Optional[int] = None,
              views_in_flight: int = 2) -> torch.Tensor:
        """Renders this rank's shard of `cameras`, reduces each view to a row with per_view(index, cam, rgb, depth) ->
        1-D tensor, and returns the [len(cameras), K] table on every rank.  The views of a sweep are independent (nothing
        is written but the rows), so `views_in_flight` of them are rendered at a time on as many streams
        (gaustar_amd.pipelines: the small kernels of one view run under the blend of another); per_view runs on the
        worker's stream."""
        mine = camera_shard(len(cameras), rank, world)
        dev = self.means3D.device
        if views_in_flight > 1 and len(mine) > 1 and dev.type == "cuda":
            from . import pipelines
            for cam in (cameras[i] for i in mine):
                self._cam(cam)                      # upload the matrices before the workers start (the cache is not locked)
            slots = [None] * len(mine)

            def work(_t, j):
                i = mine[j]
                slots[j] = per_view(i, cameras[i], *self.render_rgb_depth(cameras[i]))
            pipelines.ViewPipelines(min(int(views_in_flight), len(mine)), dev).run(work, list(range(len(mine))))
            rows = slots
        else:
            rows = [per_view(i, cameras[i], *self.render_rgb_depth(cameras[i])) for i in mine]
        local = torch.stack(rows) if rows else torch.zeros(0, 1, device=dev)
        return gather_rows(local, len(cameras), rank, world)
