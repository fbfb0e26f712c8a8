"""Test helper: write a COLMAP binary reconstruction (format of COLMAP's src/base/reconstruction.cc) from plain data.
Used by tests/golden/make_golden_io.py and tests/test_io.py to fabricate small scenes."""
import os
import struct

import numpy as np


def rotmat2qvec(R):
    """(w, x, y, z) of a rotation matrix (the usual eigenvector construction)."""
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = R.flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0], [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0],
                  [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def write_reconstruction(sparse_dir, width, height, focal, names, c2ws, points, tracks):
    """cameras.bin (one SIMPLE_RADIAL camera, id 1), images.bin (ids 1..N in the given order, poses from the
    camera-to-world matrices), points3D.bin (ids = index, tracks = lists of 1-based image ids)."""
    os.makedirs(sparse_dir, exist_ok=True)
    with open(os.path.join(sparse_dir, "cameras.bin"), "wb") as f:
        f.write(struct.pack("<Q", 1))
        f.write(struct.pack("<iiQQ", 1, 2, width, height))
        f.write(struct.pack("<dddd", focal, width / 2, height / 2, 0.01))
    with open(os.path.join(sparse_dir, "images.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(names)))
        for i, (name, c2w) in enumerate(zip(names, c2ws)):
            w2c = np.linalg.inv(c2w)
            q = rotmat2qvec(w2c[:3, :3])
            f.write(struct.pack("<idddddddi", i + 1, *q, *w2c[:3, 3], 1))
            f.write(name.encode() + b"\x00")
            obs = [(float(5 * k + i), float(3 * k), k) for k, tr in enumerate(tracks) if (i + 1) in tr]
            f.write(struct.pack("<Q", len(obs)))
            for x, y, pid in obs:
                f.write(struct.pack("<ddq", x, y, pid))
    with open(os.path.join(sparse_dir, "points3D.bin"), "wb") as f:
        f.write(struct.pack("<Q", len(points)))
        for k, (p, tr) in enumerate(zip(points, tracks)):
            f.write(struct.pack("<QdddBBBd", k, *p, 10 + k % 200, 20, 30, 0.5))
            f.write(struct.pack("<Q", len(tr)))
            for j in tr:
                f.write(struct.pack("<ii", j, k % 7))
