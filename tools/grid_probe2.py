import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gnss_sdr_rs_amd import _lib, acquisition as A, distributed as Dm, synth
_lib.init(0)
ca = A.ca_code_table(); b1i = A.b1i_codes(range(1, 23))
sc = synth.cfg4_grid_scene(ca, b1i)
fams = Dm.baseline_grid_families(sc, b1i)
d_x = torch.from_numpy(synth.to_i8_iq(sc["x"])).cuda()
D = sc["D"]
ptrs = {0: d_x.data_ptr(), 1: d_x.data_ptr(), 2: d_x.data_ptr()}
import os
for world, r in ((1, 0), (8, 2), (8, 3), (8, 6)):
    g = Dm.MixedGrid(fams, world, r, stream=(torch.cuda.Stream().cuda_stream if os.environ.get("PROBE_EXT") else None))
    for trial in range(3):
        blk = g.search_dev(ptrs, A.FMT_I8_IQ)
        torch.cuda.synchronize()
        b = blk.cpu().numpy().reshape(3, g.pmax, D)
        # the engines' own view after a full sync
        for p in g.parts:
            met = p["met"].cpu().numpy().reshape(3, p["cnt"], D)
            mine = b[:, p["row"]:p["row"] + p["cnt"], :]
            z = np.argwhere(mine[0] == 0)
            print("world", world, "rank", r, "trial", trial, "family", fams[p["fi"]].name, "rows", p["cnt"], "block==met:", bool((mine == met).all()),
                  "zero maxima in block:", len(z), "in met:", int((met[0] == 0).sum()), "bins", sorted(set(z[:, 1].tolist()))[:12])
    g.close()
