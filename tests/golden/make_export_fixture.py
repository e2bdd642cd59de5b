#!/usr/bin/env python3
"""Record what the REAL reference's save_model writes (build container only): for each of the four scripts' models,
the exported file names, each TorchScript module's forward schema, its state_dict keys/shapes, and -- to pin the
numerics of a reloaded export -- seeded inputs/outputs of every exported module with weights as raw arrays.

    python tests/golden/make_export_fixture.py   ->  tests/golden/export_schema.json, tests/golden/g6_export_io.npz
Only data is written; no reference source travels."""
import json
import os
import pathlib
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_goldens import OUT, load_reference  # noqa: E402

CASES = {"ode01": ("neural_00_ODE_01_no_encode", "ODE_Model", (8, 2, 64)), "ode02": ("neural_00_ODE_02_direct_encode", "ODE_Model", (8, 2, 16)),
         "dae01": ("neural_01_DAE_01_no_encode", "DAE_Model", (8, 2, 2, 2, 64)), "dae02": ("neural_01_DAE_02_direct_encode", "DAE_Model", (8, 2, 2, 2, 16)),
         "dae02_z0": ("neural_01_DAE_02_direct_encode", "DAE_Model", (8, 0, 2, 2, 16))}


def main():
    import importlib
    load_reference()
    schema, io = {}, {}
    for tag, (script, cls, dims) in CASES.items():
        torch.manual_seed(0)
        model = getattr(importlib.import_module(script), cls)(*dims)
        with tempfile.TemporaryDirectory() as d:
            model.save_model(pathlib.Path(d) / "m")
            files = sorted(os.listdir(os.path.join(d, "m")))
            entry = {"dims": list(dims), "files": files, "modules": {}}
            if "dim.txt" in files:
                entry["dim_txt"] = open(os.path.join(d, "m", "dim.txt")).read()
            for f in files:
                if not f.endswith(".pt"):
                    continue
                sm = torch.jit.load(os.path.join(d, "m", f))
                name = f[:-3]
                args = [a.name for a in sm.forward.schema.arguments][1:]
                sd = sm.state_dict()
                entry["modules"][name] = {"args": args, "state": {k: list(v.shape) for k, v in sd.items()}}
                # one seeded call per exported module: inputs [5, width] per forward argument
                g = torch.Generator().manual_seed(len(name))
                first_in = next(v for k, v in sd.items() if k.endswith("0.weight")).shape[1]
                ins = _inputs_for(name, args, dims, first_in, g)
                out = sm(*ins)
                for k, v in sd.items():
                    io[f"{tag}.{name}.state.{k}"] = v.numpy()
                for a, v in zip(args, ins):
                    io[f"{tag}.{name}.in.{a}"] = v.numpy()
                io[f"{tag}.{name}.out"] = out.detach().numpy()
        schema[tag] = entry
    json.dump(schema, open(os.path.join(OUT, "export_schema.json"), "w"), indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, "g6_export_io.npz"), **io)
    print("wrote export_schema.json,", len(io), "arrays")


def _inputs_for(name, args, dims, first_in, g):
    """Input tensors [5, w] per forward argument, widths consistent with the module's first Linear."""
    r = lambda w: 0.3 * torch.randn(5, w, generator=g)
    if len(args) == 1:                       # nn.Sequential encoders / decoders: forward(input)
        return [r(first_in)]
    if name == "init_func":                  # forward(z0, v0, i0)
        z, v, i = (dims[1], dims[2], dims[3])
        return [r(z), r(v), r(i)]
    if name == "de_func":
        n = first_in // 3                    # all_initial width
        if "vt" in args:
            if dims[-1] == 16:               # latent DAE: blocks of H
                H = 16
                z = 0 if dims[1] == 0 else H
                w = {"xt": H, "zt": z, "vt": H, "it": H}
            else:
                w = {"xt": dims[0], "zt": dims[1], "vt": dims[2], "it": dims[3]}
        else:
            w = {"xt": 16, "zt": 16} if dims[-1] == 16 else {"xt": dims[0], "zt": dims[1]}
        return [r(1) if a == "t0" else (r(n) if a == "all_initial" else r(w[a])) for a in args]
    if name == "ae_func":                    # forward(xt, zt, vt, all_initial): in = n + x + z + v
        if dims[-1] == 16:
            H = 16
            z = 0 if dims[1] == 0 else H
            w = {"xt": H, "zt": z, "vt": H}
            n = first_in - (2 * H + z)
        else:
            w = {"xt": dims[0], "zt": dims[1], "vt": dims[2]}
            n = first_in - sum(w.values())
        return [r(n) if a == "all_initial" else r(w[a]) for a in args]
    raise KeyError(name)


if __name__ == "__main__":
    main()
