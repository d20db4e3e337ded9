#!/usr/bin/env python3
"""The reference's `pytorch/plume.py` main loop on this backend -- what a driver looks like after the switch.

    python examples/plume.py [--res 128] [--iters 200] [--out-iter 50] [--method jacobi] [--folder out] [--restart]

Same structure as the reference driver (plume.py:66-178 setup, :231-424 loop): build the batch, `createPlumeBCs`, optional
restart from `<folder>/restart.pth`, echo the configuration as YAML, then `simulate()` per iteration and, every `out-iter`
iterations, the PNG panels, the VTK cell data and the restart file.  Only the import line differs from a reference-side
driver: `lib` -> `fluidnet_cxx_amd`."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidnet_cxx_amd import fluid, simulate, output, load_restart      # noqa: E402   (reference: `import lib, lib.fluid as fluid`)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--out-iter", type=int, default=50)
    ap.add_argument("--method", default="jacobi", choices=["jacobi"])      # 'convnet' needs a trained FluidNet state dict
    ap.add_argument("--folder", default="plume_out")
    ap.add_argument("--restart", action="store_true")
    a = ap.parse_args(argv)
    cuda = torch.device("cuda")
    # plumeConfig.yaml:29-76 (the keys simulate() reads)
    simConf = dict(dt=0.1, maccormackStrength=0.6, sampleOutsideFluid=False, buoyancyScale=0.25, gravityScale=0, viscosity=0,
                   correctScalar=False, gravityVec=dict(x=0.0, y=-1.0, z=0.0), operatingDensity=0.0, pTol=0.0, jacobiIter=28,
                   simMethod=a.method, resX=a.res, resY=a.res, maxIter=a.iters, outputFolder=a.folder)
    os.makedirs(a.folder, exist_ok=True)
    resX = resY = a.res
    # plume.py:131-163
    p = torch.zeros(1, 1, 1, resY, resX, dtype=torch.float, device=cuda)
    U = torch.zeros(1, 2, 1, resY, resX, dtype=torch.float, device=cuda)
    flags = torch.zeros(1, 1, 1, resY, resX, dtype=torch.float, device=cuda)
    density = torch.zeros(1, 1, 1, resY, resX, dtype=torch.float, device=cuda)
    fluid.emptyDomain(flags)
    batch_dict = dict(p=p, U=U, flags=flags, density=density)
    fluid.createPlumeBCs(batch_dict, 0.1, 2, 0.145)
    it = 0
    restart_file = os.path.join(a.folder, "restart.pth")
    if a.restart:                                         # plume.py:168-175
        assert os.path.isfile(restart_file), "Restart file does not exists."
        batch_dict, it = load_restart(restart_file, cuda)
        print("Restarting from checkpoint at it = " + str(it))
    output.echo_config(os.path.join(a.folder, "plumeConfig.yaml"), simConf)      # plume.py:176-178
    while it < a.iters:                                   # plume.py:231-424
        simulate(simConf, batch_dict, None, a.method)
        if it % a.out_iter == 0:
            print("It = " + str(it))
            output.save_state(a.folder, it, batch_dict)
        it += 1
    return batch_dict, it


if __name__ == "__main__":
    main()
