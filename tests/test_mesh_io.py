"""examples/frame_loop.py mesh readers ("a mesh loader (OBJ/PLY) so real teapot / bunny / Sponza files
can be dropped in when available") — CPU-only round trips; the GPU frame loop itself is in test_gpu_parity.py."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fl():
    spec = importlib.util.spec_from_file_location("frame_loop", os.path.join(ROOT, "examples", "frame_loop.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("binary", [True, False])
def test_ply_round_trip(fl, tmp_path, binary):
    v, i = fl.sphere_mesh(0.4, 7)
    p = tmp_path / "mesh.ply"
    fl.write_ply(str(p), v, i, binary=binary)
    v2, i2 = fl.load_mesh(str(p))
    assert np.array_equal(i, i2)
    assert np.array_equal(v[:, 0:3], v2[:, 0:3]) and np.array_equal(v[:, 4:7], v2[:, 4:7])   # float32 survives both encodings


def test_ply_with_normals_uchar_colours_quads_and_extra_elements(fl, tmp_path):
    p = tmp_path / "quad.ply"
    p.write_text("ply\nformat ascii 1.0\ncomment a quad\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                 "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty float confidence\n"
                 "element face 1\nproperty list uchar int vertex_indices\nelement edge 1\nproperty int a\nproperty int b\nend_header\n"
                 "-0.5 -0.5 0.2 255 0 0 1.0\n0.5 -0.5 0.2 0 255 0 1.0\n0.5 0.5 0.2 0 0 255 0.5\n-0.5 0.5 0.2 51 102 204 0.5\n"
                 "4 0 1 2 3\n0 1\n")
    v, i = fl.load_ply(str(p))
    assert i.tolist() == [0, 1, 2, 0, 2, 3]                                   # fan triangulation
    assert np.allclose(v[:, 0:3], [[-0.5, -0.5, 0.2], [0.5, -0.5, 0.2], [0.5, 0.5, 0.2], [-0.5, 0.5, 0.2]])
    assert np.allclose(v[3, 4:7], [0.2, 0.4, 0.8]) and v[0, 4] == 1.0
    q = tmp_path / "n.ply"
    q.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nproperty float nx\n"
                 "property float ny\nproperty float nz\nelement face 1\nproperty list uchar uint vertex_indices\nend_header\n"
                 "0 0 0 0 0 -1\n1 0 0 0 -1 0\n0 1 0 -1 0 0\n3 0 1 2\n")
    v, i = fl.load_ply(str(q))
    assert i.tolist() == [0, 1, 2] and np.array_equal(v[:, 4:7], np.eye(3, dtype=np.float32)[::-1])   # colour = |normal|
    with pytest.raises(ValueError):
        (tmp_path / "bad.ply").write_text("plx\n")
        fl.load_ply(str(tmp_path / "bad.ply"))
    with pytest.raises(ValueError):                 # a header that never ends must not loop for ever
        (tmp_path / "cut.ply").write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\n")
        fl.load_ply(str(tmp_path / "cut.ply"))


def test_obj_and_ply_agree(fl, tmp_path):
    o = tmp_path / "quad.obj"
    o.write_text("v -0.5 -0.5 0.2\nv 0.5 -0.5 0.2\nv 0.5 0.5 0.2\nv -0.5 0.5 0.2\nvn 0 0 1\nf 1//1 2//1 3//1 4//1\n")
    v, i = fl.load_mesh(str(o))
    p = tmp_path / "quad.ply"
    fl.write_ply(str(p), v, i)
    v2, i2 = fl.load_mesh(str(p))
    assert np.array_equal(v, v2) and np.array_equal(i, i2)
