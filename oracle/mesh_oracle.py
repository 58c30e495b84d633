"""CPU side of the ONet-Mesh path (ONet/remesh_defense.py:128-170) - TEST INFRASTRUCTURE.

The two native pieces of the reference, MISE (im2mesh/utils/libmise/mise.pyx) and marching cubes
(im2mesh/utils/libmcubes), are not restated: the reference's own sources are compiled into ``oracle/_ref`` by
``oracle/build_ref.py`` and imported from there (the strongest possible checker; kind "reference" when timed).  What
is restated here is the Python around them - ``Generator3D.generate_from_latent`` / ``extract_mesh``
(im2mesh/onet/generation.py:88-178) with the decoder of ``onet_oracle`` - and ``trimesh.sample.sample_surface``
(trimesh is not installed; its documented algorithm: faces drawn with probability proportional to their area, then a
uniform point of the face from two uniforms reflected into the triangle).  The surface samples of the reference are
unseeded numpy draws, so parity downstream of the mesh is distributional ("parity unpinned" for the samples).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import onet_oracle as OO

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def ref_libs():
    """(mise, mcubes) modules built from the reference sources; ImportError if oracle/_ref is not built."""
    if _REF not in sys.path:
        sys.path.insert(0, _REF)
    import mcubes
    import mise
    return mise, mcubes


def occupancy_grid(w, c_row: torch.Tensor, resolution0: int = 32, upsampling_steps: int = 2, threshold: float = 0.2,
                   padding: float = 0.1, points_batch_size: int = 100000):
    """generate_from_latent (generation.py:97-129): MISE query / eval / update loop -> dense logit grid, logit threshold."""
    mise, _ = ref_libs()
    thr = np.log(threshold) - np.log(1. - threshold)
    box_size = 1 + padding
    m = mise.MISE(resolution0, upsampling_steps, thr)
    points = m.query()
    while points.shape[0] != 0:
        pointsf = torch.FloatTensor(points)
        pointsf = pointsf / m.resolution
        pointsf = box_size * (pointsf - 0.5)
        with torch.no_grad():
            vals = torch.cat([OO.decode_logits(w, p[None], c_row[None])[0] for p in torch.split(pointsf, points_batch_size)])
        m.update(points, vals.numpy().astype(np.float64))
        points = m.query()
    return m.to_dense(), thr


def extract_mesh(value_grid: np.ndarray, thr: float, padding: float = 0.1):
    """extract_mesh (generation.py:155-178): pad with -1e6, marching cubes, undo the shifts, scale to the box."""
    _, mcubes = ref_libs()
    n_x, n_y, n_z = value_grid.shape
    box_size = 1 + padding
    vertices, triangles = mcubes.marching_cubes(np.pad(value_grid, 1, 'constant', constant_values=-1e6), thr)
    vertices -= 0.5
    vertices -= 1
    vertices /= np.array([n_x - 1, n_y - 1, n_z - 1])
    vertices = box_size * (vertices - 0.5)
    return vertices, triangles.astype(np.int64)


def sample_surface(vertices: np.ndarray, faces: np.ndarray, count: int, rng: np.random.Generator = None,
                   uniforms: np.ndarray = None) -> np.ndarray:
    """trimesh.sample.sample_surface (remesh_defense.py:155-156), restated from its documentation: faces drawn with
    probability proportional to their area (cumulative sum + searchsorted on `count` uniforms), then a uniform point of
    the face from two more uniforms, reflected into the triangle when their sum exceeds one.  trimesh itself is not
    installed here (parity against its code is unpinned); `uniforms` [count, 3] (face pick, two barycentric draws) makes
    the restatement a deterministic function, which is what the GPU sampler is pinned against."""
    tri = vertices[faces]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    cum = np.cumsum(area)
    if uniforms is not None:
        pick, r = np.asarray(uniforms[:, 0], np.float64), np.array(uniforms[:, 1:3], np.float64)
    else:
        pick, r = rng.random(count), rng.random((count, 2))
    face = np.searchsorted(cum, pick * cum[-1])
    flip = r.sum(1) > 1
    r[flip] = 1 - r[flip]
    t = tri[face]
    return t[:, 0] + r[:, :1] * (t[:, 1] - t[:, 0]) + r[:, 1:] * (t[:, 2] - t[:, 0])


def normalize_pc(points: np.ndarray) -> np.ndarray:
    """remesh_defense.py:61-66."""
    points = points - np.mean(points, axis=0)[None, :]
    return points / np.max(np.sqrt(np.sum(points ** 2, axis=1)), 0)


def remesh(w, sel: torch.Tensor, count: int = 1024, threshold: float = 0.2, seed: int = 0, **grid_kw) -> np.ndarray:
    """reconstruct_mesh + resample_points + normalize_pc for pre-processed encoder inputs sel [B,T,3]."""
    rng = np.random.default_rng(seed)
    c = OO.encode_latent(w, sel)
    out = []
    for b in range(sel.shape[0]):
        grid, thr = occupancy_grid(w, c[b], threshold=threshold, **grid_kw)
        v, f = extract_mesh(grid, thr)
        out.append(normalize_pc(sample_surface(v, f, count, rng)))
    return np.stack(out).astype(np.float32)
