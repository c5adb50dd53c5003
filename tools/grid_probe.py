"""Diagnostic for tests/test_gpu_mixed_grid.py: which words of the 8-shard grid differ from the 1-shard grid."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gnss_sdr_rs_amd import _lib, acquisition as A, distributed as Dm, synth
_lib.init(0)
ca = A.ca_code_table(); b1i = A.b1i_codes(range(1, 23))
sc = synth.cfg4_grid_scene(ca, b1i)
fams = Dm.baseline_grid_families(sc, b1i)
d_x = torch.from_numpy(synth.to_i8_iq(sc["x"])).cuda()
from test_gpu_mixed_grid import _gather_world
D = sc["D"]
g1, grid1 = _gather_world(fams, 1, d_x)
g1b, grid1b = _gather_world(fams, 1, d_x)
print("world-1 repeat equal:", bool((g1.cpu() == g1b.cpu()).all()))
g8, grid8 = _gather_world(fams, 8, d_x)
a1 = Dm.grid_assemble(g1.cpu().numpy(), fams, 1, D); a8 = Dm.grid_assemble(g8.cpu().numpy(), fams, 8, D)
for fi, f in enumerate(fams):
    d = np.argwhere(a1[fi] != a8[fi])
    print(f.name, "differing words", len(d), "per plane", [int((d[:, 0] == q).sum()) for q in range(3)], "rows", sorted(set(d[:, 1].tolist()))[:40])
    for q, p, b in d[:6]:
        x1, x8 = a1[fi][q, p, b], a8[fi][q, p, b]
        print("   plane", q, "row", p, "bin", b, x1.view(np.float32) if q != 1 else x1, x8.view(np.float32) if q != 1 else x8)
for g in (grid1, grid1b, grid8):
    g.close()
