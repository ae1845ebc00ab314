"""PLY / checkpoint compatibility with the reference's on-disk formats (scene/gaussian_model.py:62-113,220-299)."""
import struct

import numpy as np
import pytest
import torch

from bags_raster.gaussians import GaussianBag
from bags_raster.io import (attribute_names, capture, load_checkpoint, load_ply, read_ply_vertices, restore,
                            save_checkpoint, save_ply)
from bags_raster.synth import synth_scene


def _bag(P=37, deg=3, seed=4):
    return GaussianBag.from_activated(synth_scene(P, seed, 1.0, deg), deg)


def test_attribute_names_follow_the_reference_order():
    # construct_list_of_attributes (scene/gaussian_model.py:220-233) for SH degree 3
    n = attribute_names(3, 45)
    assert n[:6] == ["x", "y", "z", "nx", "ny", "nz"] and n[6:9] == ["f_dc_0", "f_dc_1", "f_dc_2"]
    assert n[9] == "f_rest_0" and n[53] == "f_rest_44" and n[54] == "opacity"
    assert n[55:58] == ["scale_0", "scale_1", "scale_2"] and n[58:] == ["rot_0", "rot_1", "rot_2", "rot_3"] and len(n) == 62


def test_ply_bytes_are_what_plyfile_writes(tmp_path):
    """Header text and body layout of PlyData([PlyElement.describe(elements, 'vertex')]).write(path): binary little endian,
    one float property per attribute, rows of 62 float32; f_dc / f_rest channel-major (transpose(1, 2).flatten(1))."""
    pc = _bag(P=5)
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    save_ply(pc, path)                                    # creates the directories, like mkdir_p(os.path.dirname(path))
    blob = open(path, "rb").read()
    names = attribute_names(3, 45)
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex 5\n" + "".join(f"property float {n}\n" for n in names)
              + "end_header\n").encode()
    assert blob.startswith(header) and len(blob) == len(header) + 5 * 62 * 4
    row0 = struct.unpack("<62f", blob[len(header):len(header) + 62 * 4])
    assert np.allclose(row0[0:3], pc._xyz[0].detach().numpy()) and row0[3:6] == (0.0, 0.0, 0.0)
    assert np.allclose(row0[6:9], pc._features_dc[0, 0].detach().numpy())
    rest = pc._features_rest[0].detach().numpy()          # (15, 3): coefficient-major in memory, channel-major on disk
    assert np.allclose(row0[9:9 + 15], rest[:, 0]) and np.allclose(row0[9 + 15:9 + 30], rest[:, 1])
    assert np.allclose(row0[54], pc._opacity[0, 0].item())
    assert np.allclose(row0[55:58], pc._scaling[0].detach().numpy()) and np.allclose(row0[58:62], pc._rotation[0].detach().numpy())


@pytest.mark.parametrize("deg", [0, 1, 3])
def test_ply_round_trip(tmp_path, deg):
    pc = _bag(P=41, deg=deg)
    path = str(tmp_path / "pc.ply")
    save_ply(pc, path)
    back = load_ply(path, deg)
    for a, b in zip(pc.leaves(), back.leaves()):
        assert a.shape == b.shape and torch.equal(a.detach(), b.detach()) and b.requires_grad
    assert back.active_sh_degree == deg and back.max_radii2D.shape == (41,) and back.denom.shape == (41, 1)
    with pytest.raises(AssertionError):
        load_ply(path, deg + 1)                           # the reference asserts on the f_rest count (:275)


def test_reader_accepts_ascii_big_endian_doubles_and_shuffled_properties(tmp_path):
    """load_ply picks properties by name; files rewritten by other tools (ASCII, doubles, extra columns, another
    property order, a trailing face element) must load to the same leaves."""
    pc = _bag(P=6, deg=1)
    names = attribute_names(3, 9)
    ref_path = str(tmp_path / "ref.ply")
    save_ply(pc, ref_path)
    v = read_ply_vertices(ref_path)
    order = list(reversed(names)) + ["extra"]
    cols = {**v, "extra": np.arange(6, dtype=np.float32)}
    # ASCII
    p1 = str(tmp_path / "ascii.ply")
    with open(p1, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment rewritten\nelement vertex 6\n" + "".join(f"property double {n}\n" for n in order)
                + "element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for i in range(6):
            f.write(" ".join(repr(float(cols[n][i])) for n in order) + "\n")
    # big endian doubles
    p2 = str(tmp_path / "be.ply")
    with open(p2, "wb") as f:
        f.write(("ply\nformat binary_big_endian 1.0\nelement vertex 6\n" + "".join(f"property double {n}\n" for n in order)
                 + "end_header\n").encode())
        f.write(np.stack([cols[n].astype(">f8") for n in order], axis=1).astype(">f8").tobytes())
    for p in (p1, p2):
        back = load_ply(p, 1)
        for a, b in zip(pc.leaves(), back.leaves()):
            assert torch.equal(a.detach(), b.detach())
    with pytest.raises(ValueError):
        open(str(tmp_path / "bad.ply"), "wb").write(open(p2, "rb").read()[:-5])
        load_ply(str(tmp_path / "bad.ply"), 1)


def test_checkpoint_tuple_matches_capture_and_restores(tmp_path):
    pc = _bag(P=9, deg=2)
    pc.max_radii2D = torch.arange(9.0)
    opt = {"state": {}, "param_groups": [{"name": "xyz", "lr": 1e-4}]}
    t = capture(pc, opt, spatial_lr_scale=2.5)
    assert len(t) == 12 and t[0] == pc.active_sh_degree and t[1] is pc._xyz and t[6] is pc._opacity and t[10] is opt and t[11] == 2.5
    path = str(tmp_path / "chkpnt30000.pth")
    save_checkpoint(pc, 30000, path, opt, 2.5)
    back, opt2, scale, it = load_checkpoint(path, 2)
    assert it == 30000 and scale == 2.5 and opt2["param_groups"][0]["name"] == "xyz"
    for a, b in zip(pc.leaves(), back.leaves()):
        assert torch.equal(a.detach(), b.detach())
    assert torch.equal(back.max_radii2D, pc.max_radii2D)
    # the older 15-tuple layout (scene/gaussian_model.py:93-113): three extra slots the reference skips
    old = (t[0], t[1], "mlp", "mlp2", t[2], t[3], t[4], t[5], t[6], "x", t[7], t[8], t[9], t[10], t[11])
    back15, _, _ = restore(old, 2)
    assert torch.equal(back15._features_dc, pc._features_dc) and torch.equal(back15._opacity, pc._opacity)
    with pytest.raises(ValueError):
        restore(t[:5], 2)


def test_reference_checkpoint_with_numpy_lr_scale_loads_without_the_full_unpickler(tmp_path):
    """The reference's capture() stores spatial_lr_scale = scene.cameras_extent, a numpy.float64
    (scene/dataset_readers.py:100, scene/__init__.py:115): the safe loader must take it (weights_only=True rejects numpy
    scalars unless their reconstructors are allow-listed) -- and must still refuse arbitrary pickled objects."""
    pc = _bag(P=5, deg=1)
    opt = {"state": {}, "param_groups": [{"name": "xyz", "lr": 1e-4}]}
    t = list(capture(pc, opt, spatial_lr_scale=0.0))
    t[11] = np.float64(4.25)
    path = str(tmp_path / "chkpnt7000.pth")
    torch.save((tuple(t), 7000), path)
    back, _, scale, it = load_checkpoint(path, 1)
    assert it == 7000 and scale == 4.25 and isinstance(scale, float)
    assert torch.equal(back._xyz.detach(), pc._xyz.detach())

    class Evil:                                              # anything outside tensors / numbers / containers
        pass
    t[11] = Evil()
    import pickle
    bad = str(tmp_path / "bad.pth")
    try:
        torch.save((tuple(t), 1), bad)
    except (pickle.PicklingError, AttributeError):           # a local class may not even pickle: nothing to load then
        return
    with pytest.raises(Exception):
        load_checkpoint(bad, 1)


@pytest.mark.parametrize("writer_numpy", ["numpy.core.multiarray", "numpy._core.multiarray"])
def test_checkpoint_written_under_either_numpy_major_loads(tmp_path, writer_numpy):
    """The module path of the pickled numpy scalar is the WRITER's: the reference's environment (numpy 1.x) writes
    ``numpy.core.multiarray.scalar``, numpy 2 writes ``numpy._core.multiarray.scalar``.  The file is rewritten to carry the
    given spelling (protocol-2 GLOBAL opcodes are newline-terminated text, so a byte replacement is a valid pickle) and must
    load with weights_only=True whichever numpy is installed here."""
    import zipfile
    pc = _bag(P=4, deg=1)
    t = list(capture(pc, {"state": {}, "param_groups": []}, spatial_lr_scale=0.0))
    t[11] = np.float64(2.5)
    src = str(tmp_path / "src.pth")
    torch.save((tuple(t), 30000), src)
    dst = str(tmp_path / "chkpnt30000.pth")
    seen = False
    with zipfile.ZipFile(src) as zi, zipfile.ZipFile(dst, "w", zipfile.ZIP_STORED) as zo:
        for info in zi.infolist():
            data = zi.read(info.filename)
            if info.filename.endswith("data.pkl"):
                for spelling in (b"numpy._core.multiarray", b"numpy.core.multiarray"):
                    if b"c" + spelling + b"\nscalar\n" in data:
                        data = data.replace(b"c" + spelling + b"\nscalar\n", b"c" + writer_numpy.encode() + b"\nscalar\n")
                        seen = True
                        break
            zo.writestr(info.filename, data)
    assert seen, "the checkpoint did not pickle a numpy scalar through a GLOBAL opcode"
    back, _, scale, it = load_checkpoint(dst, 1)
    assert it == 30000 and scale == 2.5 and torch.equal(back._xyz.detach(), pc._xyz.detach())
