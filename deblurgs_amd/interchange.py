"""Interchange with reference-trained DeblurGS scenes (SURVEY.md 8f, row f4 -- formats only): the Gaussian PLY
written by GaussianModel.save_ply / read by load_ply (scene/gaussian_model.py:206-299), the training checkpoint
tuple of GaussianModel.capture (scene/gaussian_model.py:80-112, train.py:214-216) and the camera-motion file
`cm.pth` (scene/motion.py:337-365).  Pure numpy/torch (the reference depends on `plyfile`, absent here): the PLY
is the little-endian binary layout plyfile produces, one float32 property per attribute in the reference's order.

Fork-specific conventions reproduced: opacity is stored as inverse_sigmoid(activated opacity) and loaded back as
clamp(sigmoid(x)) (the in-memory parameter is the identity-with-clamp activation's argument); scales are stored as
log(activated scale); SH coefficients are stored channel-major (f_rest = [P, 3, M-1] flattened).
"""
import numpy as np
import torch

from .cloud import GaussianCloud


def ply_attributes(n_rest):
    """scene/gaussian_model.py:206-224 (construct_list_of_attributes)."""
    attrs = ['x', 'y', 'z', 'nx', 'ny', 'nz']
    attrs += ['f_dc_{}'.format(i) for i in range(3)]
    attrs += ['f_rest_{}'.format(i) for i in range(n_rest)]
    attrs += ['opacity']
    attrs += ['scale_{}'.format(i) for i in range(3)]
    attrs += ['rot_{}'.format(i) for i in range(4)]
    return attrs


def _inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def save_ply(cloud: GaussianCloud, path: str):
    xyz = cloud._xyz.detach().cpu().numpy()
    normals = np.zeros_like(xyz)
    f_dc = cloud._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = cloud._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    opacities = _inverse_sigmoid(cloud.get_opacity).detach().cpu().numpy().reshape(-1, 1)
    scale = torch.log(cloud.get_scaling).detach().cpu().numpy()
    rotation = cloud._rotation.detach().cpu().numpy()
    attrs = ply_attributes(f_rest.shape[1])
    data = np.concatenate((xyz, normals, f_dc, f_rest, opacities, scale, rotation), axis=1).astype('<f4')
    assert data.shape[1] == len(attrs)
    header = "ply\nformat binary_little_endian 1.0\nelement vertex {}\n".format(data.shape[0])
    header += "".join("property float {}\n".format(a) for a in attrs) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(np.ascontiguousarray(data).tobytes())


def _read_ply(path):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("not a PLY file")
        fmt, count, props = None, None, []
        in_vertex = False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("unterminated PLY header")
            tok = line.decode("ascii").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                props.append((tok[2], tok[1]))
            elif tok[0] == "end_header":
                break
        types = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "uchar": "u1", "uint8": "u1",
                 "int": "i4", "int32": "i4", "uint": "u4", "short": "i2", "ushort": "u2", "char": "i1"}
        if fmt == "binary_little_endian":
            dt = np.dtype([(n, "<" + types[t]) for n, t in props])
            return np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count)
        if fmt == "ascii":
            arr = np.loadtxt(f, max_rows=count, ndmin=2)
            dt = np.dtype([(n, "<f8") for n, _ in props])
            out = np.empty(count, dtype=dt)
            for i, (n, _) in enumerate(props):
                out[n] = arr[:, i]
            return out
        raise ValueError("unsupported PLY format " + str(fmt))


def load_ply(path: str, sh_degree: int = 2, device="cuda", **cloud_kw) -> GaussianCloud:
    """scene/gaussian_model.py:248-299."""
    v = _read_ply(path)
    names = v.dtype.names
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1).astype(np.float32)
    opac = torch.sigmoid(torch.from_numpy(np.array(v["opacity"], np.float32)[..., None])).clamp(0.0, 1.0)
    features_dc = np.zeros((xyz.shape[0], 3, 1), np.float32)
    for c in range(3):
        features_dc[:, c, 0] = v["f_dc_{}".format(c)]
    extra = sorted([n for n in names if n.startswith("f_rest_")], key=lambda x: int(x.split('_')[-1]))
    assert len(extra) == 3 * (sh_degree + 1) ** 2 - 3, "PLY SH coefficient count does not match sh_degree"
    features_extra = np.stack([v[n] for n in extra], axis=1).astype(np.float32) if extra else \
        np.zeros((xyz.shape[0], 0), np.float32)
    features_extra = features_extra.reshape((xyz.shape[0], 3, (sh_degree + 1) ** 2 - 1))
    scale_names = sorted([n for n in names if n.startswith("scale_")], key=lambda x: int(x.split('_')[-1]))
    rot_names = sorted([n for n in names if n.startswith("rot")], key=lambda x: int(x.split('_')[-1]))
    scales = np.stack([v[n] for n in scale_names], axis=1).astype(np.float32)
    rots = np.stack([v[n] for n in rot_names], axis=1).astype(np.float32)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
    cloud = GaussianCloud(t(xyz), t(features_dc).transpose(1, 2).contiguous(), t(features_extra).transpose(1, 2).contiguous(),
                          t(scales), t(rots), opac.float().to(device), sh_degree=sh_degree, active_sh_degree=sh_degree,
                          **cloud_kw)
    return cloud


def load_checkpoint(path: str, device="cuda", **cloud_kw):
    """chkpnt{it}.pth = torch.save((gaussians.capture(), iteration)) (train.py:214-216).  Returns
    (GaussianCloud, iteration, extras) where extras holds max_radii2D / xyz_gradient_accum / denom / the optimiser
    state dict / spatial_lr_scale untouched."""
    model_args, iteration = torch.load(path, map_location="cpu", weights_only=False)
    (active_sh_degree, xyz, f_dc, f_rest, scaling, rotation, opacity, max_radii2D, xyz_gradient_accum, denom, opt_dict,
     spatial_lr_scale) = model_args
    M = f_dc.shape[1] + f_rest.shape[1]
    sh_degree = int(round(M ** 0.5)) - 1
    d = lambda a: a.detach().to(device).float().contiguous()
    cloud = GaussianCloud(d(xyz), d(f_dc), d(f_rest), d(scaling), d(rotation), d(opacity), sh_degree=sh_degree,
                          active_sh_degree=int(active_sh_degree), **cloud_kw)
    extras = dict(max_radii2D=max_radii2D, xyz_gradient_accum=xyz_gradient_accum, denom=denom, optimizer=opt_dict,
                  spatial_lr_scale=spatial_lr_scale)
    return cloud, iteration, extras


def save_checkpoint(cloud: GaussianCloud, iteration: int, path: str):
    """train.py:214-216: torch.save((gaussians.capture(), iteration), model_path + "/chkpnt<it>.pth")."""
    torch.save((cloud.capture(), iteration), path)


def save_camera_motion(module, path: str):
    """scene/motion.py:337-350."""
    assert path.endswith(".pth")
    torch.save({"rot": module._rot.state_dict(), "trans": module._trans.state_dict(), "nu": module._nu}, path)


def load_camera_motion(module, path: str):
    """scene/motion.py:352-365 (a directory means <dir>/cm.pth)."""
    import os
    state_dict_path = path if path.endswith(".pth") else os.path.join(path, "cm.pth")
    sdict = torch.load(state_dict_path, map_location=module.device, weights_only=False)
    module._rot.load_state_dict({k: v for k, v in sdict["rot"].items() if k == "_control_points"}, strict=False)
    module._trans.load_state_dict({k: v for k, v in sdict["trans"].items() if k == "_control_points"}, strict=False)
    module._nu = sdict["nu"]
