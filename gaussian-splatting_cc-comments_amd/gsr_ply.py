"""PLY formats either side of the path, without the `plyfile` dependency.

Two files of the reference:
  * the trained-scene checkpoint `point_cloud/iteration_N/point_cloud.ply`
    (scene/gaussian_model.py:277-308 save_ply, :323-364 load_ply): one `vertex` element, every property
    float32, in the order  x y z  nx ny nz  f_dc_0..2  f_rest_0..(3*(M-1)-1)  opacity  scale_0..2  rot_0..3,
    with f_dc / f_rest stored CHANNEL-major (`_features_rest.transpose(1,2).flatten(1)`), raw leaves
    (log-scales, logit opacity, unnormalised quaternion);
  * the COLMAP / synthetic input cloud `points3D.ply` (scene/dataset_readers.py:123-146 fetchPly/storePly):
    x y z nx ny nz float32 + red green blue uint8.
The reference reads and writes them through plyfile (environment.yml:8, version unpinned), which is not in
this image: the byte layout here follows the PLY 1.0 specification as plyfile's PlyData.write() emits it
for a structured array (binary_little_endian, `property float|uchar <name>` lines, no comments).  Byte
identity of the header with plyfile's is unpinned (no file written by it exists in the reference tree);
the reader accepts ascii / little / big endian and both spellings of the scalar type names.
"""
import os

import numpy as np

_TYPES = {  # PLY scalar type name -> numpy type code (no byte order)
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
    "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8",
}
_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def write_ply(path, columns, element="vertex"):
    """columns: ordered [(name, 1-D array)], equal lengths; written as one binary little-endian element."""
    n = len(columns[0][1]) if columns else 0
    dtype = np.dtype([(name, "<" + np.asarray(col).dtype.str[1:]) for name, col in columns])
    rec = np.empty(n, dtype=dtype)
    for name, col in columns:
        if len(col) != n:
            raise ValueError(f"column {name!r} has {len(col)} rows, expected {n}")
        rec[name] = col
    header = ["ply", "format binary_little_endian 1.0", f"element {element} {n}"]
    header += [f"property {_NAMES[np.dtype(dtype[name]).str[1:]]} {name}" for name, _ in columns]
    header.append("end_header")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)  # utils/system_utils.py mkdir_p
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(rec.tobytes())


def read_ply(path, element="vertex"):
    """-> {property name: 1-D array} of the first element called `element` (insertion order = file order).
    Elements stored before it are skipped (they must have scalar properties only); list properties inside
    `element` are not supported."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt = None
        elements = []  # [name, count, [(type, name)], has_list]
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: end of file inside the header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] in ("comment", "obj_info"):
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append([tok[1], int(tok[2]), [], False])
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1][3] = True
                else:
                    if tok[1] not in _TYPES:
                        raise ValueError(f"{path}: unknown property type {tok[1]!r}")
                    elements[-1][2].append((_TYPES[tok[1]], tok[2]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported format {fmt!r}")
        order = "<" if fmt != "binary_big_endian" else ">"
        for name, count, props, has_list in elements:
            if has_list and (name == element or fmt != "ascii"):
                if name == element:
                    raise ValueError(f"{path}: list properties in element {name!r} are not supported")
                raise ValueError(f"{path}: cannot skip element {name!r} with list properties")
            dtype = np.dtype([(pn, order + pt) for pt, pn in props])
            if fmt == "ascii":
                rows = [f.readline().split() for _ in range(count)]
                if name != element:
                    continue
                rec = np.empty(count, dtype=dtype)
                for j, (pt, pn) in enumerate(props):
                    rec[pn] = np.array([r[j] for r in rows], dtype="f8" if pt[0] == "f" else "i8").astype(pt) if count else []
            else:
                buf = f.read(count * dtype.itemsize)
                if len(buf) != count * dtype.itemsize:
                    raise ValueError(f"{path}: element {name!r} is truncated")
                if name != element:
                    continue
                rec = np.frombuffer(buf, dtype=dtype, count=count)
            return {pn: np.ascontiguousarray(rec[pn]).astype(pt) for pt, pn in props}  # native byte order
    raise ValueError(f"{path}: no element named {element!r}")


# ---- trained-scene checkpoint (scene/gaussian_model.py:262-364) ------------------------------------------
def construct_list_of_attributes(n_dc, n_rest, n_scale=3, n_rot=4):
    """gaussian_model.py:262-275"""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return names


def save_gaussians_ply(path, xyz, features_dc, features_rest, opacity, scaling, rotation):
    """gaussian_model.py:277-295 save_ply.  Arguments are the raw leaves as numpy arrays or torch tensors:
    xyz (P,3), features_dc (P,1,3), features_rest (P,M-1,3), opacity (P,1), scaling (P,3), rotation (P,4)."""
    to_np = lambda t: (t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)).astype(np.float32)
    xyz, features_dc, features_rest, opacity, scaling, rotation = map(to_np, (xyz, features_dc, features_rest, opacity, scaling, rotation))
    P = xyz.shape[0]
    f_dc = features_dc.transpose(0, 2, 1).reshape(P, -1)       # channel-major, like .transpose(1, 2).flatten(start_dim=1)
    f_rest = features_rest.transpose(0, 2, 1).reshape(P, -1)
    attributes = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, opacity.reshape(P, 1), scaling, rotation), axis=1)
    names = construct_list_of_attributes(f_dc.shape[1], f_rest.shape[1], scaling.shape[1], rotation.shape[1])
    assert attributes.shape[1] == len(names)
    write_ply(path, [(n, attributes[:, j]) for j, n in enumerate(names)])


def load_gaussians_ply(path, max_sh_degree=3):
    """gaussian_model.py:323-364 load_ply -> dict of float32 numpy leaves
    {xyz (P,3), features_dc (P,1,3), features_rest (P,(D+1)^2-1,3), opacity (P,1), scaling (P,3), rotation (P,4)}."""
    v = read_ply(path)
    by_index = lambda prefix: sorted((n for n in v if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float32)
    P = xyz.shape[0]
    opacity = np.asarray(v["opacity"], np.float32)[:, None]
    f_dc = np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1).astype(np.float32).reshape(P, 3, 1)
    extra = by_index("f_rest_")
    if len(extra) != 3 * (max_sh_degree + 1) ** 2 - 3:   # the reference asserts the same (gaussian_model.py:337)
        raise ValueError(f"{path}: {len(extra)} f_rest_* properties, expected {3 * (max_sh_degree + 1) ** 2 - 3} for SH degree {max_sh_degree}")
    f_rest = (np.stack([v[n] for n in extra], axis=1) if extra else np.zeros((P, 0))).astype(np.float32)
    f_rest = f_rest.reshape(P, 3, (max_sh_degree + 1) ** 2 - 1)
    scaling = np.stack([v[n] for n in by_index("scale_")], axis=1).astype(np.float32)
    rotation = np.stack([v[n] for n in by_index("rot")], axis=1).astype(np.float32)
    return dict(xyz=xyz, features_dc=np.ascontiguousarray(f_dc.transpose(0, 2, 1)),
                features_rest=np.ascontiguousarray(f_rest.transpose(0, 2, 1)), opacity=opacity, scaling=scaling, rotation=rotation)


# ---- input point cloud (scene/dataset_readers.py:123-146) ------------------------------------------------
def store_ply(path, xyz, rgb):
    """storePly: xyz (N,3) float, rgb (N,3) 0..255."""
    xyz = np.asarray(xyz, np.float32)
    rgb = np.asarray(rgb).astype(np.uint8)
    zeros = np.zeros(xyz.shape[0], np.float32)
    write_ply(path, [("x", xyz[:, 0]), ("y", xyz[:, 1]), ("z", xyz[:, 2]), ("nx", zeros), ("ny", zeros), ("nz", zeros),
                     ("red", rgb[:, 0]), ("green", rgb[:, 1]), ("blue", rgb[:, 2])])


def fetch_ply(path):
    """fetchPly -> (points (N,3), colors (N,3) in [0,1], normals (N,3))  [BasicPointCloud fields]"""
    v = read_ply(path)
    points = np.vstack([v["x"], v["y"], v["z"]]).T
    colors = np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0
    normals = np.vstack([v["nx"], v["ny"], v["nz"]]).T
    return points, colors, normals
