"""Seeded synthetic topology generator for the hot path's three graphs.

The reference builds its graphs offline with ``anemoi-graphs`` (trimesh / sklearn / PyG,
none of which are shipped here; SURVEY.md §2.2 marks the package out of scope).  The hot
path only needs graphs with the same node counts, edge counts and in-degree distribution,
so this module re-derives them from first principles:

* hidden mesh: refined icosahedron, ``10*4**r + 2`` nodes, sorted by latitude (descending)
  then longitude like ``get_coordinates_ordering`` (reference
  graphs/src/anemoi/graphs/generate/utils.py:15-33); processor edges are the union of the
  directed face edges of refinement levels 1..r mapped to finest ids (multi-scale 1-hop
  edges, reference generate/tri_icosahedron.py:155-199, edges/builders/multi_scale.py:54-76),
  ``sum(60*4**l)`` edges, symmetric.
* data grid: octahedral reduced Gaussian grid O<n> (``4*n*(n+9)`` points; equal-angle
  latitudes are used instead of exact Gaussian latitudes, stated in DESIGN.md) or a
  Fibonacci sphere stand-in for N320.
* encoder edges: data nodes within ``0.6 * reference distance`` of each hidden node, at most
  64 nearest (reference edges/builders/cutoff.py:32-200, utils.py:65-85).
* decoder edges: the 3 nearest hidden nodes of every data node (reference
  edges/builders/knn.py).

All edge lists are returned sorted by destination (stable), int64 ``[2, M]`` with row 0 = src.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
from scipy.spatial import cKDTree


# --------------------------------------------------------------------------- geometry helpers
def latlon_to_xyz(latlon: np.ndarray) -> np.ndarray:
    lat, lon = latlon[:, 0], latlon[:, 1]
    return np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=1)


def xyz_to_latlon(xyz: np.ndarray) -> np.ndarray:
    lat = np.arcsin(np.clip(xyz[:, 2], -1.0, 1.0))
    lon = np.arctan2(xyz[:, 1], xyz[:, 0])
    return np.stack([lat, lon], axis=1)


def _base_icosahedron():
    t = (1.0 + 5.0**0.5) / 2.0
    v = np.array(
        [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
         [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]],
        dtype=np.float64,
    )
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2],
         [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5],
         [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]],
        dtype=np.int64,
    )
    return v, f


def _subdivide(verts: np.ndarray, faces: np.ndarray):
    """Split every triangle in 4; old vertices keep their ids, midpoints are appended."""
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], axis=0)
    e_sorted = np.sort(e, axis=1)
    uniq, inv = np.unique(e_sorted, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    mid = verts[uniq[:, 0]] + verts[uniq[:, 1]]
    mid /= np.linalg.norm(mid, axis=1, keepdims=True)
    n0, nf = verts.shape[0], faces.shape[0]
    m01, m12, m20 = inv[:nf] + n0, inv[nf:2 * nf] + n0, inv[2 * nf:] + n0
    a, b, c = faces[:, 0], faces[:, 1], faces[:, 2]
    new_faces = np.concatenate(
        [np.stack([a, m01, m20], 1), np.stack([m01, b, m12], 1), np.stack([m20, m12, c], 1), np.stack([m01, m12, m20], 1)],
        axis=0,
    )
    return np.concatenate([verts, mid], axis=0), new_faces


def icosphere(resolution: int):
    """Return (vertices[N,3], [faces at level 0..resolution])."""
    v, f = _base_icosahedron()
    levels = [f]
    for _ in range(resolution):
        v, f = _subdivide(v, f)
        levels.append(f)
    return v, levels


def _latlon_order(latlon: np.ndarray) -> np.ndarray:
    """Latitude descending, then longitude ascending (deterministic tie-break)."""
    return np.lexsort((latlon[:, 1], -np.round(latlon[:, 0], 12)))


def icosphere_latlon(resolution: int):
    """Sorted lat/lon [N,2] (radians) and the ordering (new id -> generator id)."""
    v, _ = icosphere(resolution)
    ll = xyz_to_latlon(v)
    order = _latlon_order(ll)
    return ll[order], order


def multiscale_edges(resolution: int, levels=None) -> np.ndarray:
    """Directed multi-scale 1-hop edges [2,M] (src,dst) in sorted numbering, dst-sorted."""
    v, faces = icosphere(resolution)
    ll = xyz_to_latlon(v)
    order = _latlon_order(ll)
    inv = np.empty_like(order)
    inv[order] = np.arange(order.size)
    levels = list(range(1, resolution + 1)) if levels is None else list(levels)
    es = []
    for l in levels:
        f = faces[l]
        es.append(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=0))
    e = inv[np.concatenate(es, axis=0)].T  # [2, M]
    return sort_by_dst(e)


def sort_by_dst(edge_index: np.ndarray) -> np.ndarray:
    # primary dst, secondary src: deterministic
    perm = np.lexsort((edge_index[0], edge_index[1]))
    return np.ascontiguousarray(edge_index[:, perm]).astype(np.int64)


def octahedral_grid(n: int = 96) -> np.ndarray:
    """O<n> reduced grid: 2n latitude rows, row i (from the pole) has 20+4i points. [N,2] radians."""
    lats, lons = [], []
    nlat = 2 * n
    for j in range(nlat):
        lat = np.pi / 2 - (j + 0.5) * np.pi / nlat
        i = j if j < n else nlat - 1 - j
        npts = 20 + 4 * i
        lon = np.arange(npts) * (2 * np.pi / npts)
        lons.append(lon)
        lats.append(np.full(npts, lat))
    ll = np.stack([np.concatenate(lats), np.concatenate(lons)], axis=1)
    ll[:, 1] = np.where(ll[:, 1] > np.pi, ll[:, 1] - 2 * np.pi, ll[:, 1])
    return ll


def fibonacci_grid(n: int) -> np.ndarray:
    """Fibonacci sphere with n points (stand-in for N320 = 542080 points), latitude-descending."""
    k = np.arange(n) + 0.5
    z = 1.0 - 2.0 * k / n
    lon = (np.pi * (1 + 5**0.5) * k) % (2 * np.pi)
    lon = np.where(lon > np.pi, lon - 2 * np.pi, lon)
    return np.stack([np.arcsin(z), lon], axis=1)


def reference_distance(latlon: np.ndarray) -> float:
    """Max nearest-neighbour chord distance (reference graphs/utils.py:65-85)."""
    xyz = latlon_to_xyz(latlon)
    d, _ = cKDTree(xyz).query(xyz, k=2)
    return float(d[:, 1].max())


def cutoff_edges(src_latlon: np.ndarray, dst_latlon: np.ndarray, cutoff_factor: float = 0.6, max_neighbours: int = 64):
    """Encoder edges: for every dst node the src nodes within factor*refdist(dst), <=max nearest."""
    radius = cutoff_factor * reference_distance(dst_latlon)
    sx, dx = latlon_to_xyz(src_latlon), latlon_to_xyz(dst_latlon)
    tree = cKDTree(sx)
    d, idx = tree.query(dx, k=max_neighbours, distance_upper_bound=radius)
    valid = np.isfinite(d)
    dst = np.broadcast_to(np.arange(dx.shape[0])[:, None], idx.shape)[valid]
    src = idx[valid]
    return sort_by_dst(np.stack([src, dst], axis=0))


def knn_edges(src_latlon: np.ndarray, dst_latlon: np.ndarray, k: int = 3):
    """Decoder edges: the k nearest src nodes of every dst node."""
    sx, dx = latlon_to_xyz(src_latlon), latlon_to_xyz(dst_latlon)
    _, idx = cKDTree(sx).query(dx, k=k)
    idx = idx.reshape(dx.shape[0], k)
    dst = np.broadcast_to(np.arange(dx.shape[0])[:, None], idx.shape).reshape(-1)
    return sort_by_dst(np.stack([idx.reshape(-1), dst], axis=0))


def edge_attributes(src_latlon: np.ndarray, dst_latlon: np.ndarray, edge_index: np.ndarray) -> np.ndarray:
    """[M,3] = (edge_length / max, direction_x, direction_y): same shape/normalisation class as the
    reference's EdgeLength + EdgeDirection (graphs/edges/attributes.py:87-100); values are synthetic."""
    s, d = src_latlon[edge_index[0]], dst_latlon[edge_index[1]]
    sx, dx = latlon_to_xyz(s), latlon_to_xyz(d)
    chord = np.linalg.norm(sx - dx, axis=1)
    length = 2 * np.arcsin(np.clip(chord / 2, 0, 1))
    length = length / max(length.max(), 1e-12)
    dlat = s[:, 0] - d[:, 0]
    dlon = (s[:, 1] - d[:, 1] + np.pi) % (2 * np.pi) - np.pi
    vec = np.stack([dlon * np.cos(d[:, 0]), dlat], axis=1)
    nrm = np.linalg.norm(vec, axis=1, keepdims=True)
    vec = np.where(nrm > 0, vec / np.maximum(nrm, 1e-12), 0.0)
    return np.concatenate([length[:, None], vec], axis=1).astype(np.float32)


@dataclass
class SyntheticGraph:
    """Container mirroring what the reference reads from its HeteroData graph file."""

    data_latlon: np.ndarray
    hidden_latlon: np.ndarray
    enc_edge_index: np.ndarray  # data -> hidden
    proc_edge_index: np.ndarray  # hidden -> hidden
    dec_edge_index: np.ndarray  # hidden -> data
    enc_edge_attr: np.ndarray
    proc_edge_attr: np.ndarray
    dec_edge_attr: np.ndarray
    meta: dict = field(default_factory=dict)

    @property
    def num_data(self) -> int:
        return self.data_latlon.shape[0]

    @property
    def num_hidden(self) -> int:
        return self.hidden_latlon.shape[0]


def build_synthetic_graph(data_grid: str = "o96", hidden_resolution: int = 5, *, cutoff_factor: float = 0.6,
                          max_neighbours: int = 64, decoder_k: int = 3) -> SyntheticGraph:
    """``data_grid``: "o<n>" (octahedral), "fib<n>" (Fibonacci with n points) or "n320"."""
    g = data_grid.lower()
    if g == "n320":
        data = fibonacci_grid(542080)
    elif g.startswith("fib"):
        data = fibonacci_grid(int(g[3:]))
    elif g.startswith("o"):
        data = octahedral_grid(int(g[1:]))
    else:
        raise ValueError(f"unknown data grid '{data_grid}'")
    hidden, _ = icosphere_latlon(hidden_resolution)
    enc = cutoff_edges(data, hidden, cutoff_factor, max_neighbours)
    proc = multiscale_edges(hidden_resolution)
    dec = knn_edges(hidden, data, decoder_k)
    return SyntheticGraph(
        data_latlon=data.astype(np.float32),
        hidden_latlon=hidden.astype(np.float32),
        enc_edge_index=enc,
        proc_edge_index=proc,
        dec_edge_index=dec,
        enc_edge_attr=edge_attributes(data, hidden, enc),
        proc_edge_attr=edge_attributes(hidden, hidden, proc),
        dec_edge_attr=edge_attributes(hidden, data, dec),
        meta={"data_grid": data_grid, "hidden_resolution": hidden_resolution},
    )
