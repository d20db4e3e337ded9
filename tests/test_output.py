"""Field dumps (f3): the dependency-free VTK / PNG writers on the CPU, the dumped cell arrays on the GPU against the arrays
the reference driver's own statements produce (tests/golden/dump.npz, plume.py:317-408)."""
import struct
import zlib

import numpy as np
import pytest


def _writers():
    # the writers are plain numpy; importing the module pulls in the operator surface (native extension) like every other
    # product module, so this import also proves the extension loads on the CPU box
    from fluidnet_cxx_amd import output
    return output


def test_vtr_roundtrip(tmp_path, golden):
    out = _writers()
    z = golden("dump")
    cells = {k[4:]: z[k] for k in z.files if k.startswith("vtk_") and k[4:] not in ("x", "y", "z")}
    f = out.grid_to_vtk(str(tmp_path / "output_00007"), z["vtk_x"], z["vtk_y"], z["vtk_z"], cells)
    assert f.endswith("output_00007.vtr")
    head = open(f, "rb").read(400).decode(errors="replace")
    assert 'type="RectilinearGrid"' in head and 'byte_order="LittleEndian"' in head and 'WholeExtent="0 33 0 19 0 0"' in head
    coords, back = out.read_vtr(f)
    for k in "xyz":
        assert np.array_equal(coords[k], z["vtk_" + k])
    assert set(back) == set(cells)
    for k, v in cells.items():
        assert np.array_equal(back[k].view(np.int32), np.ascontiguousarray(v).view(np.int32)), k     # NaNs included
    # x runs fastest in the file: the first appended block is `cells` of the first key in Fortran order
    raw = open(f, "rb").read()
    blob = raw[raw.index(b"_", raw.index(b"<AppendedData")) + 1:]
    n = struct.unpack("<Q", blob[:8])[0]
    first = np.frombuffer(blob[8:8 + n], np.float32)
    k0 = next(iter(cells))
    assert np.array_equal(first.view(np.int32), cells[k0][:, :, 0].T.ravel().view(np.int32))


def test_coordinates_match_reference(golden):
    out = _writers()
    z = golden("dump")
    x, y, zz = out.vtk_coordinates(33, 19)
    for a, k in ((x, "x"), (y, "y"), (zz, "z")):
        assert a.dtype == np.float32 and np.array_equal(a, z["vtk_" + k]), k


def test_png_writer(tmp_path):
    out = _writers()
    a = np.linspace(-1, 2, 7 * 5, dtype=np.float32).reshape(7, 5)
    a[3, 2] = np.nan
    rgb = out.colorize(a)
    assert rgb.shape == (7, 5, 3) and tuple(rgb[7 - 1 - 3, 2]) == (128, 128, 128)      # NaN grey, row 0 at the bottom
    assert tuple(rgb[-1, 0]) == (0, 0, 128) and tuple(rgb[0, -1]) == (128, 0, 0)        # jet end points
    f = out.write_png(str(tmp_path / "t.png"), rgb)
    raw = open(f, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(raw):
        n = struct.unpack(">I", raw[pos:pos + 4])[0]
        tag, payload = raw[pos + 4:pos + 8], raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + payload) & 0xFFFFFFFF
        chunks[tag] = payload; pos += 12 + n
    assert struct.unpack(">IIBBBBB", chunks[b"IHDR"]) == (5, 7, 8, 2, 0, 0, 0)
    px = np.frombuffer(zlib.decompress(chunks[b"IDAT"]), np.uint8).reshape(7, 1 + 5 * 3)
    assert (px[:, 0] == 0).all() and np.array_equal(px[:, 1:].reshape(7, 5, 3), rgb)


def test_yaml_echo(tmp_path):
    import yaml
    out = _writers()
    conf = dict(simMethod="jacobi", jacobiIter=28, gravityVec=dict(x=0.0, y=-1.0, z=0.0), dt=0.1)
    out.echo_config(str(tmp_path / "plumeConfig.yaml"), conf)
    assert yaml.safe_load(open(tmp_path / "plumeConfig.yaml")) == conf


@pytest.mark.gpu
def test_vtk_cell_data_vs_reference(tmp_path, golden):
    """The device-side field arithmetic of the dump (divergence, centred velocity, centred gradients, obstacle masking,
    window, x-first layout) against the arrays the reference driver builds for gridToVTK: bit-exact, NaNs in place."""
    import torch
    out = _writers()
    z = golden("dump")
    dev = torch.device("cuda:0")
    bd = {k: torch.from_numpy(z["in_" + k].copy()).to(dev) for k in ("U", "p", "density", "flags")}
    win = tuple(int(v) for v in z["window"])
    cells = out.vtk_cell_data(bd, win)
    names = dict(density="rho", divergence="divergence", pressure="p", ux="velx", uy="vely", gradPx="gradPx", gradPy="gradPy",
                 gradRhox="gradRhox", gradRhoy="gradRhoy")
    assert set(cells) == set(names)
    for k, ref in names.items():
        a, b = cells[k], z["vtk_" + ref]
        assert a.shape == b.shape and a.dtype == np.float32, k
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        assert np.array_equal(np.nan_to_num(a), np.nan_to_num(b)), f"{k}: max diff {np.nanmax(np.abs(a - b))}"
    assert np.isnan(cells["pressure"]).any()
    files = out.save_state(str(tmp_path), 12, bd, win)
    assert [f.rsplit("/", 1)[1] for f in files] == ["output_00012.png", "output_00012.vtr", "restart.pth"]
    _, back = out.read_vtr(files[1])
    assert np.array_equal(np.nan_to_num(back["gradPy"]), np.nan_to_num(z["vtk_gradPy"]))
    # 3D: same arithmetic per plane + the z component; checked against the 2D routine on a z-constant field
    U3 = torch.zeros(1, 3, 4, 24, 38, device=dev); U3[:, 0:2] = bd["U"].expand(1, 2, 4, 24, 38)
    bd3 = dict(U=U3, p=bd["p"].expand(1, 1, 4, 24, 38).contiguous(), density=bd["density"].expand(1, 1, 4, 24, 38).contiguous(),
               flags=bd["flags"].expand(1, 1, 4, 24, 38).contiguous())
    c3 = out.vtk_cell_data(bd3, win)
    assert c3["density"].shape == (33, 19, 4) and "uz" in c3 and "gradPz" in c3
    for k in ("gradPx", "gradRhoy", "ux"):
        assert np.array_equal(np.nan_to_num(c3[k][:, :, 1]), np.nan_to_num(cells[k][:, :, 0])), k
    assert not np.nan_to_num(c3["gradPz"]).any()


@pytest.mark.gpu
def test_reference_shaped_driver_runs_and_restarts(tmp_path):
    """examples/plume.py (the reference driver's loop on this backend): runs, writes the YAML echo, PNG, VTK and restart
    files of every output event, and a run continued from its restart file ends in the bits of an uninterrupted one."""
    import importlib.util
    import os
    import torch
    import yaml
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("plume_example", os.path.join(repo, "examples", "plume.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    fa, fb = str(tmp_path / "a"), str(tmp_path / "b")
    full, it = mod.main(["--res", "64", "--iters", "12", "--out-iter", "5", "--folder", fa])
    assert it == 12
    for name in ("plumeConfig.yaml", "output_00000.png", "output_00005.vtr", "output_00010.png", "output_00010.vtr", "restart.pth"):
        assert os.path.isfile(os.path.join(fa, name)), name
    assert yaml.safe_load(open(os.path.join(fa, "plumeConfig.yaml")))["jacobiIter"] == 28
    # the restart file holds the state AFTER iteration 10; continue a copy of it to the end
    os.makedirs(fb)
    import shutil
    shutil.copy(os.path.join(fa, "restart.pth"), os.path.join(fb, "restart.pth"))
    # (the driver's loop re-runs the saved iteration number: plume.py increments `it` after saving, like the reference)
    cont, it2 = mod.main(["--res", "64", "--iters", "11", "--out-iter", "50", "--folder", fb, "--restart"])
    assert it2 == 11
    for k in ("p", "U", "density"):
        assert torch.equal(cont[k], full[k]), k
