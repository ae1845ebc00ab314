"""On-disk formats of a trained scene, as the reference writes and reads them (second half).

* ``point_cloud/iteration_<n>/point_cloud.ply`` -- ``GaussianModel.save_ply`` / ``load_ply``
  (scene/gaussian_model.py:220-299; call sites scene/__init__.py:158,208-210): one ``vertex`` element
  of float32 properties ``x y z nx ny nz f_dc_0..2 f_rest_0..(3(K-1)-1) opacity scale_0..2 rot_0..3`` holding the RAW
  (pre-activation) leaves; ``f_dc`` / ``f_rest`` are stored channel-major (``features.transpose(1, 2).flatten(1)``).
  The reference goes through the ``plyfile`` package, which is not a dependency here: the writer emits the same header
  and ``binary_little_endian`` body byte for byte, the reader accepts binary (either endianness) and ASCII bodies and any
  scalar property type, and picks properties by NAME as ``load_ply`` does.
* ``chkpnt<iter>.pth`` -- ``torch.save((gaussians.capture(), iteration), ...)`` (train.py:487-489): ``capture()`` is the
  12-tuple of scene/gaussian_model.py:62-76; ``restore`` also accepts the 15-tuple of older checkpoints (:78-113).

Everything here is host-side bookkeeping around the hot path: plain numpy / PyTorch, device-agnostic.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .gaussians import GaussianBag

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4) -> List[str]:
    """``construct_list_of_attributes`` (scene/gaussian_model.py:220-233)."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)]
    names += [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)]
    names += [f"rot_{i}" for i in range(n_rot)]
    return names


def save_ply(pc: GaussianBag, path: str) -> None:
    """scene/gaussian_model.py:235-252: raw leaves, normals zero, features channel-major, float32, little endian."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    xyz = pc._xyz.detach().cpu().numpy().astype(np.float32)
    f_dc = pc._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = pc._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, pc._opacity.detach().cpu().numpy(),
            pc._scaling.detach().cpu().numpy(), pc._rotation.detach().cpu().numpy()]
    table = np.ascontiguousarray(np.concatenate([c.reshape(xyz.shape[0], -1) for c in cols], axis=1), dtype="<f4")
    names = attribute_names(f_dc.shape[1], f_rest.shape[1], cols[5].shape[1], cols[6].shape[1])
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {table.shape[0]}\n" + \
             "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


def read_ply_vertices(path: str) -> Dict[str, np.ndarray]:
    """The ``vertex`` element of a PLY file as {property name: 1-D array}.  Scalar properties only (what the reference
    writes); other elements may follow the vertex element and are ignored, list properties are rejected."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements, cur = None, [], None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: header without end_header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = {"name": tok[1], "count": int(tok[2]), "props": []}
                elements.append(cur)
            elif tok[0] == "property":
                if tok[1] == "list":
                    if cur["name"] == "vertex":
                        raise ValueError(f"{path}: list property in the vertex element")
                    cur["props"].append((tok[-1], None))
                else:
                    cur["props"].append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if not elements or elements[0]["name"] != "vertex":
            raise ValueError(f"{path}: the first element must be 'vertex'")
        v = elements[0]
        n, names = v["count"], [p[0] for p in v["props"]]
        if fmt == "ascii":
            rows = np.loadtxt(f, dtype=np.float64, max_rows=n, ndmin=2) if n else np.zeros((0, len(names)))
            if rows.shape != (n, len(names)):
                raise ValueError(f"{path}: expected {n} x {len(names)} vertex values, found {rows.shape}")
            return {nm: rows[:, k].astype(np.dtype(v["props"][k][1])) for k, nm in enumerate(names)}
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unknown PLY format {fmt!r}")
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(nm, order + t) for nm, t in v["props"]])
        raw = f.read(n * dt.itemsize)
        if len(raw) != n * dt.itemsize:
            raise ValueError(f"{path}: truncated vertex data ({len(raw)} of {n * dt.itemsize} bytes)")
        rec = np.frombuffer(raw, dtype=dt, count=n)
        return {nm: np.ascontiguousarray(rec[nm]) for nm in names}


def load_ply(path: str, sh_degree: int, device="cpu", requires_grad: bool = True) -> GaussianBag:
    """scene/gaussian_model.py:259-299, including its checks: properties are taken by name, ``f_rest_*`` / ``scale_*`` /
    ``rot*`` sorted by their numeric suffix, the number of ``f_rest`` properties must match ``sh_degree``."""
    v = read_ply_vertices(path)
    for need in ("x", "y", "z", "opacity", "f_dc_0", "f_dc_1", "f_dc_2"):
        if need not in v:
            raise KeyError(f"{path}: property {need!r} missing")
    P = v["x"].shape[0]
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
    opac = np.asarray(v["opacity"])[:, None]
    f_dc = np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1)[:, :, None]                 # (P, 3, 1)

    def numbered(prefix):
        names = sorted((n for n in v if n.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
        return np.stack([v[n] for n in names], axis=1) if names else np.zeros((P, 0))
    rest = numbered("f_rest_")
    if rest.shape[1] != 3 * (sh_degree + 1) ** 2 - 3:
        raise AssertionError(f"{path}: {rest.shape[1]} f_rest properties, sh_degree {sh_degree} needs {3 * (sh_degree + 1) ** 2 - 3}")
    rest = rest.reshape(P, 3, (sh_degree + 1) ** 2 - 1)                                          # (P, 3, K-1)
    scales, rots = numbered("scale_"), numbered("rot")
    dev = torch.device(device)

    def leaf(a):
        return torch.tensor(np.asarray(a, dtype=np.float32), dtype=torch.float32, device=dev).contiguous().requires_grad_(requires_grad)
    pc = GaussianBag(sh_degree)
    pc._xyz = leaf(xyz)
    pc._features_dc = leaf(np.ascontiguousarray(f_dc.transpose(0, 2, 1)))                        # (P, 1, 3)
    pc._features_rest = leaf(np.ascontiguousarray(rest.transpose(0, 2, 1)))                      # (P, K-1, 3)
    pc._opacity, pc._scaling, pc._rotation = leaf(opac), leaf(scales), leaf(rots)
    pc.max_radii2D = torch.zeros(P, device=dev)
    pc.xyz_gradient_accum = torch.zeros(P, 1, device=dev)
    pc.denom = torch.zeros(P, 1, device=dev)
    pc.active_sh_degree = sh_degree                                                              # :299
    return pc


def capture(pc: GaussianBag, optimizer_state: Optional[dict] = None, spatial_lr_scale: float = 0.0) -> Tuple:
    """The 12-tuple of ``GaussianModel.capture`` (scene/gaussian_model.py:62-76)."""
    return (pc.active_sh_degree, pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity,
            pc.max_radii2D, pc.xyz_gradient_accum, pc.denom, optimizer_state if optimizer_state is not None else {},
            spatial_lr_scale)


def restore(model_args: Tuple, sh_degree: int, device=None) -> Tuple[GaussianBag, dict, float]:
    """``GaussianModel.restore`` (scene/gaussian_model.py:78-113) minus the optimiser set-up, which stays with the
    caller: returns (bag, optimizer state dict, spatial_lr_scale).  Accepts the 12-tuple and the older 15-tuple."""
    if len(model_args) == 12:
        (active, xyz, f_dc, f_rest, scaling, rotation, opacity, max_radii2D, grad_accum, denom, opt_dict, lr_scale) = model_args
    elif len(model_args) == 15:
        (active, xyz, _, _, f_dc, f_rest, scaling, rotation, opacity, _, max_radii2D, grad_accum, denom, opt_dict,
         lr_scale) = model_args
    else:
        raise ValueError(f"checkpoint tuple of length {len(model_args)}: expected 12 or 15 (scene/gaussian_model.py:79,93)")
    pc = GaussianBag(sh_degree)
    mv = (lambda t: t) if device is None else (lambda t: t.to(device) if torch.is_tensor(t) else t)
    pc.active_sh_degree = int(active)
    pc._xyz, pc._features_dc, pc._features_rest = mv(xyz), mv(f_dc), mv(f_rest)
    pc._scaling, pc._rotation, pc._opacity = mv(scaling), mv(rotation), mv(opacity)
    pc.max_radii2D, pc.xyz_gradient_accum, pc.denom = mv(max_radii2D), mv(grad_accum), mv(denom)
    return pc, opt_dict, float(lr_scale)


def save_checkpoint(pc: GaussianBag, iteration: int, path: str, optimizer_state: Optional[dict] = None,
                    spatial_lr_scale: float = 0.0) -> None:
    """``torch.save((gaussians.capture(), iteration), model_path + "/chkpnt<iter>.pth")`` (train.py:487-489)."""
    torch.save((capture(pc, optimizer_state, spatial_lr_scale), iteration), path)


def _numpy_scalar_globals():
    """What the safe unpickler must accept for the reference's own files: ``capture()`` stores ``spatial_lr_scale`` =
    ``scene.cameras_extent`` = ``nerf_normalization['radius']``, a ``numpy.float64`` (scene/dataset_readers.py:100,
    scene/__init__.py:115), which pickles as ``<numpy core>.multiarray.scalar(dtype('f8'), bytes)``.  The module path in the
    file is the WRITER's: ``numpy.core.multiarray`` from the reference's numpy 1.x environment, ``numpy._core.multiarray``
    from numpy 2 -- both spellings are registered (tuple form: object + the name it is pickled under), whichever numpy
    reads the file."""
    import numpy as np
    out = [np.dtype, np.float64, np.float32, np.int64, np.int32, type(np.dtype(np.float64)), type(np.dtype(np.float32)),
           type(np.dtype(np.int64)), type(np.dtype(np.int32))]
    try:
        from numpy._core.multiarray import scalar as _scalar        # numpy >= 2
    except ImportError:                                              # numpy 1.x
        from numpy.core.multiarray import scalar as _scalar
    out += [(_scalar, "numpy.core.multiarray.scalar"), (_scalar, "numpy._core.multiarray.scalar")]
    return out


def _safe_load(path, device):
    allow = _numpy_scalar_globals()
    ser = torch.serialization
    if hasattr(ser, "safe_globals"):                                 # torch >= 2.5: scoped allow-list
        with ser.safe_globals(allow):
            return torch.load(path, map_location=device, weights_only=True)
    if hasattr(ser, "add_safe_globals"):                             # torch 2.4: process-wide allow-list, no tuple form
        ser.add_safe_globals([a[0] if isinstance(a, tuple) else a for a in allow])
    return torch.load(path, map_location=device, weights_only=True)


def load_checkpoint(path: str, sh_degree: int, device=None, trust_pickle: bool = False) -> Tuple[GaussianBag, dict, float, int]:
    """Inverse of ``save_checkpoint`` and of the reference's own checkpoints (train.py:132-135 restores them the same way).

    The tuple holds tensors, numbers (incl. the numpy scalar the reference stores as ``spatial_lr_scale``) and an optimizer
    state dict only, so it is read with ``weights_only=True`` (no code from the file is executed).  A checkpoint that needs the full unpickler (the reference writes ``nn.Parameter`` objects, which
    the safe loader accepts; anything else does not) is only read with ``trust_pickle=True`` -- the reference's own behaviour,
    to be used on files you wrote yourself.  A view-sharded run must call ``DensificationSync.rebase()`` after restoring."""
    try:
        model_args, iteration = _safe_load(path, device)
    except Exception:
        if not trust_pickle:
            raise
        model_args, iteration = torch.load(path, map_location=device, weights_only=False)
    pc, opt, lr_scale = restore(model_args, sh_degree, device)
    return pc, opt, lr_scale, int(iteration)
