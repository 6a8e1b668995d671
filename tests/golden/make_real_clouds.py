"""Makes tests/golden/real_lidar_pair.npz from the only real sensor data the reference ships: the two consecutive LiDAR scans
/root/reference/ndt_omp/data/251370668.pcd and 251371071.pcd (69 088 and 69 792 points, binary PCD v0.7, fields x y z intensity).

The fixture is DATA ONLY: for every scan two disjoint voxel-thinned subsets (0.25 m voxels, first point per voxel of the two halves
of a seeded random split of the raw points; returns closer than 0.5 m to the sensor dropped) as float32 x y z intensity.  Two disjoint samples of one scan see the same surfaces
through different points, which is what a known-transform test needs (tests/test_real_clouds.py).  Run here, once:
    python tests/golden/make_real_clouds.py
(the reference tree does not exist on the GPU box; the tests read only the committed .npz).
"""
import os
import sys

import numpy as np

SRC = ["/root/reference/ndt_omp/data/251370668.pcd", "/root/reference/ndt_omp/data/251371071.pcd"]
VOXEL = 0.25


def read_pcd_binary(path):
    raw = open(path, "rb").read()
    head_end = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
    header = raw[:head_end].decode("ascii", "replace").splitlines()
    fields = [l.split()[1:] for l in header if l.startswith("FIELDS")][0]
    sizes = [l.split()[1:] for l in header if l.startswith("SIZE")][0]
    types = [l.split()[1:] for l in header if l.startswith("TYPE")][0]
    n = int([l.split()[1] for l in header if l.startswith("POINTS")][0])
    assert fields == ["x", "y", "z", "intensity"] and set(sizes) == {"4"} and set(types) == {"F"}, (fields, sizes, types)
    return np.frombuffer(raw, dtype="<f4", count=n * 4, offset=head_end).reshape(n, 4).copy()


def thin(pts, voxel):
    ok = np.isfinite(pts[:, :3]).all(axis=1)
    pts = pts[ok]
    key = np.floor(pts[:, :3].astype(np.float64) / voxel).astype(np.int64)
    _, first = np.unique(key, axis=0, return_index=True)
    return pts[np.sort(first)]  # first point of every voxel, in scan order


def main():
    out = {}
    for name, path in zip("ab", SRC):
        pts = read_pcd_binary(path)
        pts = pts[np.linalg.norm(pts[:, :3], axis=1) > 0.5]
        half = np.random.default_rng(20250704).permutation(len(pts)) % 2 == 0  # the scan interleaves its beams: split at random, not by parity
        out[f"{name}_0"] = thin(pts[half], VOXEL)
        out[f"{name}_1"] = thin(pts[~half], VOXEL)
        print(path, pts.shape, "->", out[f"{name}_0"].shape, out[f"{name}_1"].shape, file=sys.stderr)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "real_lidar_pair.npz")
    np.savez_compressed(dst, voxel=np.float64(VOXEL), **out)
    print(dst, os.path.getsize(dst), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
