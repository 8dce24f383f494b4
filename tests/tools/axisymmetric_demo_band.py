"""iteration counts of the REFERENCE ALGORITHM (oracle) on the axisymmetric notebook's problem (64 x 64, BiCGSTAB, tol 1e-7)
under random summation orders of its dot products"""
import sys, warnings
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pyapes_oracle as O
warnings.simplefilter("ignore")
n = [64, 64]
mo = O.OMesh([0.0, 0.0], [1.0, 2.0], n, "double", "rz")
R, Z = mo.grid[0], mo.grid[1]
def face_vals(face):
    # dirichlet values in boolean-mask gather order of the face plane
    if face == "ru": return torch.zeros(n[1], dtype=torch.float64)
    if face == "zl": return 1 - R[:, 0] ** 2
    if face == "zu": return torch.exp(-2.0 * Z[:, -1]) * (1 - R[:, -1])
cfg = O.mixed_cfg([0.0, None, None, None], ["neumann", "dirichlet", "dirichlet", "dirichlet"], O.FACES_RZ)
for c in cfg:
    if c["bc_type"] == "dirichlet":
        c["bc_val"] = face_vals(c["bc_face"])
rhs = torch.zeros((1, *n), dtype=torch.float64)
rhs[0] = -4.0 * R ** 2 * torch.exp(-2.0 * Z)
orig = torch.sum
def run(seed):
    gen = torch.Generator().manual_seed(1000 + seed)
    def fsum(t, dim=None, **kw):
        f = t.contiguous().flatten() if dim is None else t.contiguous().flatten(1)
        perm = torch.randperm(f.shape[-1], generator=gen)
        nb = int(torch.randint(2, 64, (1,), generator=gen))
        acc = None
        for c in f[..., perm].chunk(nb, dim=-1):
            s = orig(c, dim=-1)
            acc = s if acc is None else acc + s
        return acc
    if seed >= 0: torch.sum = fsum
    try:
        x, r = O.solve_poisson(mo, cfg, rhs.clone(), method="bicgstab", tol=1e-7, max_it=1000)
    finally:
        torch.sum = orig
    return r["itr"], r["tol"]
print("plain order:", run(-1))
its = [run(s)[0] for s in range(40)]
print(sorted(its), min(its), max(its))
