"""Field dumps of the reference drivers (plume.py:176-178 YAML echo, :238-310 PNG panels, :311-421 VTK cell data), so
that a `plume.py`-shaped driver can write the same output files from state that lives on the GPU.

The field arithmetic (divergence, centred velocity, centred pressure / density gradients, obstacle masking) runs on the
device through the operator surface (`velocityDivergence`, `getCentered`) and a few slicing ops; only the finished window
is copied to the host.  The writers are dependency-free: `grid_to_vtk` emits the VTK XML rectilinear-grid file pyevtk's
`gridToVTK` writes (pyevtk is not a dependency of this package), `write_png` a plain RGB PNG.
"""
import os
import struct
import zlib

import numpy as np
import torch

from . import fluid


def echo_config(path, sim_conf):
    """plume.py:176-178: the simulation configuration written next to the output (YAML)."""
    import yaml
    with open(path, "w") as outfile:
        yaml.dump(sim_conf, outfile)


def _centered_gradient(field):
    """plume.py:343-363: the difference of a cell-centred scalar (B,1,D,H,W) to its -1 neighbour per direction on the
    interior cells, averaged to the cell centre with `getCentered` applied to that interior block (so the last interior
    column / row is 0, like every border cell).  2D: (B,2,1,H,W); 3D adds the z component the same way."""
    b, _, d, h, w = field.shape
    is3d = d > 1
    nc = 3 if is3d else 2
    zs = slice(1, d - 1) if is3d else slice(0, 1)
    inner = field[:, :, zs, 1:h - 1, 1:w - 1]
    faces = torch.empty((b, nc) + tuple(inner.shape[2:]), dtype=field.dtype, device=field.device)
    faces[:, 0] = (inner - field[:, :, zs, 1:h - 1, 0:w - 2])[:, 0]
    faces[:, 1] = (inner - field[:, :, zs, 0:h - 2, 1:w - 1])[:, 0]
    if is3d:
        faces[:, 2] = (inner - field[:, :, 0:d - 2, 1:h - 1, 1:w - 1])[:, 0]
    out = torch.zeros((b, nc, d, h, w), dtype=field.dtype, device=field.device)
    out[:, :, zs, 1:h - 1, 1:w - 1] = fluid.getCentered(faces.contiguous())[:, 0:nc]
    return out


def vtk_cell_data(batch_dict, window=None):
    """The cell arrays of the reference's VTK dump (plume.py:331-408) for sample 0, as a dict of float32 numpy arrays
    shaped (nx, ny, nz) -- x first, as gridToVTK wants them: density, divergence, pressure, ux, uy[, uz], gradPx, gradPy[, gradPz],
    gradRhox, gradRhoy[, gradRhoz].  Pressure and velocity are NaN inside obstacles.  `window` = (minX, maxX, minY, maxY)
    in cells (default: the whole domain)."""
    U, flags, p, rho = batch_dict["U"], batch_dict["flags"], batch_dict["p"], batch_dict["density"]
    is3d = U.size(1) == 3
    d, h, w = flags.shape[2:]
    minX, maxX, minY, maxY = window if window is not None else (0, w, 0, h)
    div = fluid.velocityDivergence(U.contiguous(), flags.contiguous())
    vel = fluid.getCentered(U)
    grad_rho = _centered_gradient(rho)
    grad_p = _centered_gradient(p)
    nan = torch.full((), float("nan"), dtype=p.dtype, device=p.device)
    obst = flags[0, 0] == float(fluid.CellType.TypeObstacle)

    def host(t, masked=False):                          # (D,H,W) on the device -> (nx,ny,nz) window on the host
        if masked:
            t = torch.where(obst, nan, t)
        return np.ascontiguousarray(t.permute(2, 1, 0)[minX:maxX, minY:maxY].cpu().numpy())

    comps = "xyz" if is3d else "xy"
    out = dict(density=host(rho[0, 0]), divergence=host(div[0, 0]), pressure=host(p[0, 0], True))
    for c, name in enumerate(comps):
        out["u" + name] = host(vel[0, c], True)
        out["gradP" + name] = host(grad_p[0, c])
        out["gradRho" + name] = host(grad_rho[0, c])
    return out


def vtk_coordinates(nx, ny, nz=1):
    """plume.py:317-329: node coordinates of a window of nx x ny cells, y scaled to [0, 1].  2D: z is the single value 0
    (a flat grid, as the reference writes it)."""
    ratio = nx / ny
    lx, ly = ratio, 1.0
    dx, dy = lx / nx, ly / ny
    x = np.arange(0, lx + 0.1 * dx, dx, dtype="float32")
    y = np.arange(0, ly + 0.1 * dy, dy, dtype="float32")
    z = np.zeros(1, dtype="float32") if nz == 1 else np.arange(0, nz + 0.5, 1, dtype="float32") * np.float32(dy)
    return x, y, z


_VTK_TYPES = {np.dtype("float32"): "Float32", np.dtype("float64"): "Float64", np.dtype("int32"): "Int32",
              np.dtype("uint8"): "UInt8", np.dtype("int64"): "Int64"}


def grid_to_vtk(path, x, y, z, cellData):
    """Rectilinear grid with cell data as VTK XML (`<path>.vtr`), the file pyevtk.hl.gridToVTK(path, x, y, z, cellData=...)
    produces: raw appended data, each block prefixed by its byte count (UInt64), arrays in x-fastest order.  Returns the
    file name."""
    x, y, z = (np.ascontiguousarray(a) for a in (x, y, z))
    nx, ny, nz = x.size - 1, y.size - 1, z.size - 1
    ext = f"0 {nx} 0 {ny} 0 {nz}"
    blocks, head, off = [], [], 0

    def entry(name, a, ncomp=1):
        nonlocal off
        a = np.asarray(a)
        line = (f'<DataArray Name="{name}" NumberOfComponents="{ncomp}" type="{_VTK_TYPES[a.dtype]}" '
                f'format="appended" offset="{off}"/>')
        raw = np.asfortranarray(a).tobytes(order="F")
        blocks.append(struct.pack("<Q", len(raw)) + raw)
        off += 8 + len(raw)
        return line

    keys = list(cellData.keys())
    cells = max(nx, 1) * max(ny, 1) * max(nz, 1)
    for k in keys:
        assert np.asarray(cellData[k]).size == cells, f"cell array '{k}' has {np.asarray(cellData[k]).size} values for {cells} cells"
    head.append('<?xml version="1.0"?>')
    head.append('<VTKFile type="RectilinearGrid" version="1.0" byte_order="LittleEndian" header_type="UInt64">')
    head.append(f'<RectilinearGrid WholeExtent="{ext}">')
    head.append(f'<Piece Extent="{ext}">')
    head.append(f'<CellData scalars="{keys[0]}">' if keys else "<CellData>")
    head += [entry(k, cellData[k]) for k in keys]
    head.append("</CellData>")
    head.append("<Coordinates>")
    head += [entry("x_coordinates", x), entry("y_coordinates", y), entry("z_coordinates", z)]
    head.append("</Coordinates>")
    head += ["</Piece>", "</RectilinearGrid>", '<AppendedData encoding="raw">']
    fname = path + ".vtr"
    with open(fname, "wb") as f:
        f.write(("\n".join(head) + "\n_").encode())
        for b in blocks:
            f.write(b)
        f.write(b"\n</AppendedData>\n</VTKFile>\n")
    return fname


def read_vtr(fname):
    """Reads a file written by grid_to_vtk back: (coords dict, cell dict).  For tests and quick inspection."""
    import re
    raw = open(fname, "rb").read()
    cut = raw.index(b'<AppendedData encoding="raw">')
    headtxt = raw[:cut].decode()
    data = raw[raw.index(b"_", cut) + 1:]
    nx, ny, nz = (int(v) for v in re.search(r'WholeExtent="0 (\d+) 0 (\d+) 0 (\d+)"', headtxt).groups())
    inv = {v: k for k, v in _VTK_TYPES.items()}
    arrays = {}
    for name, typ, off in re.findall(r'<DataArray Name="([^"]+)" NumberOfComponents="1" type="(\w+)" format="appended" offset="(\d+)"/>', headtxt):
        off = int(off)
        n = struct.unpack("<Q", data[off:off + 8])[0]
        arrays[name] = np.frombuffer(data[off + 8:off + 8 + n], dtype=inv[typ]).copy()
    coords = {k: arrays.pop(k + "_coordinates") for k in "xyz"}
    shape = (max(nx, 1), max(ny, 1), max(nz, 1))
    return coords, {k: v.reshape(shape, order="F") for k, v in arrays.items()}


def save_vtk(folder, it, batch_dict, window=None):
    """plume.py:311-421: `<folder>/output_<it:05>.vtr` with the reference's cell arrays on its node coordinates."""
    cells = vtk_cell_data(batch_dict, window)
    nx, ny, nz = cells["density"].shape
    x, y, z = vtk_coordinates(nx, ny, nz)
    return grid_to_vtk(os.path.join(folder, "output_{0:05}".format(it)), x, y, z, cells)


# ---- PNG panels ------------------------------------------------------------------------------------------------
def _jet(t):
    """matplotlib's `jet` as piecewise-linear ramps (the drivers' colour map, plume.py:184); t in [0,1] -> uint8 RGB."""
    r = np.clip(np.minimum(4 * t - 1.5, -4 * t + 4.5), 0, 1)
    g = np.clip(np.minimum(4 * t - 0.5, -4 * t + 3.5), 0, 1)
    b = np.clip(np.minimum(4 * t + 0.5, -4 * t + 2.5), 0, 1)
    return (np.stack([r, g, b], -1) * 255.0 + 0.5).astype(np.uint8)


def colorize(a):
    """2D float array -> (H,W,3) uint8: jet over [min,max] of the finite values, NaN (obstacles) grey, row 0 at the
    bottom (imshow origin='lower')."""
    a = np.asarray(a, np.float32)
    fin = np.isfinite(a)
    lo, hi = (float(a[fin].min()), float(a[fin].max())) if fin.any() else (0.0, 1.0)
    t = np.where(fin, (a - lo) / (hi - lo) if hi > lo else 0.5, 0.0)
    rgb = _jet(t)
    rgb[~fin] = 128
    return rgb[::-1]


def write_png(fname, rgb):
    """(H,W,3) uint8 -> PNG file (8-bit RGB, zlib)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[j].tobytes() for j in range(h))

    def chunk(tag, payload):
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(fname, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    return fname


def png_panels(batch_dict, window=None, plane=None):
    """The five fields the reference driver plots (plume.py:238-306): density, x-velocity, y-velocity (centred, NaN in
    obstacles), pressure, divergence -- sample 0, z plane `plane` (default: the middle one), window = (minX, maxX, minY,
    maxY).  Returns a dict name -> 2D float32 array (rows = y)."""
    U, flags = batch_dict["U"], batch_dict["flags"]
    d, h, w = flags.shape[2:]
    k = d // 2 if plane is None else plane
    minX, maxX, minY, maxY = window if window is not None else (0, w, 0, h)
    div = fluid.velocityDivergence(U.contiguous(), flags.contiguous())
    vel = fluid.getCentered(U)
    obst = flags[0, 0, k] == float(fluid.CellType.TypeObstacle)
    nan = torch.full((), float("nan"), dtype=U.dtype, device=U.device)

    def host(t, masked=False):
        if masked:
            t = torch.where(obst, nan, t)
        return t[minY:maxY, minX:maxX].cpu().numpy()

    return {"Density": host(batch_dict["density"][0, 0, k]), "x-velocity": host(vel[0, 0, k], True),
            "y-velocity": host(vel[0, 1, k], True), "pressure": host(batch_dict["p"][0, 0, k]),
            "divergence": host(div[0, 0, k])}


def save_png(folder, it, batch_dict, window=None, plane=None):
    """plume.py:307-309: `<folder>/output_<it:05>.png`, the five panels side by side (2 rows x 3 columns like the
    reference's figure grid, the last cell empty)."""
    panels = list(png_panels(batch_dict, window, plane).values())
    ph, pw = panels[0].shape
    pad = 4
    canvas = np.full((2 * ph + 3 * pad, 3 * pw + 4 * pad, 3), 255, np.uint8)
    for n, a in enumerate(panels):
        r, c = divmod(n, 3)
        y0, x0 = pad + r * (ph + pad), pad + c * (pw + pad)
        canvas[y0:y0 + ph, x0:x0 + pw] = colorize(a)
    return write_png(os.path.join(folder, "output_{0:05}.png".format(it)), canvas)


def save_state(folder, it, batch_dict, window=None, vtk=True, png=True, restart=True):
    """One output event of the driver loop (plume.py:238-424): PNG, VTK and the restart file."""
    from .state_io import save_restart
    files = []
    if png:
        files.append(save_png(folder, it, batch_dict, window))
    if vtk:
        files.append(save_vtk(folder, it, batch_dict, window))
    if restart:
        rf = os.path.join(folder, "restart.pth")
        save_restart(rf, batch_dict, it)
        files.append(rf)
    return files
