import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, cases, oracle_lib as O, indep_statements as I
from gvpm_amd import abi
c = cases.make_case("cbox_phong1", 20, 16, 20000, 4.0)
print(c.bsdfs.size, c.bsdfs["distribution"], c.bsdfs["exponent"])
pt = c.ph.flags & 3
gl = pt == abi.GVPM_PARENT_SURFACE_BSDF
print(gl.sum(), np.unique(c.ph.parent_g[gl], return_counts=True), np.unique(c.ph.flags[gl] >> 16, return_counts=True), np.unique((c.ph.flags[gl]>>2)&7, return_counts=True))
idx=c.ph.parent_g[gl].astype(int); ct=(c.ph.flags[gl]>>16)
print("even idx -> comp types", np.unique(ct[idx%2==0]), "odd", np.unique(ct[idx%2==1]))
g = np.flatnonzero(gl)[:400]
d = c.ph.pos[g].astype(np.float64) - c.ph.parent_pos[g]
ln = np.linalg.norm(d, axis=1); wo = d/ln[:,None]
I.set_bsdfs(c.bsdfs)
f, pdf, known = I.phong_world(c.ph.parent_scat[g].astype(np.float64), c.ph.parent_g[g].astype(np.int64), c.ph.parent_n[g].astype(np.float64), c.ph.parent_wi[g].astype(np.float64), wo)
print("pdf ok", np.allclose(pdf, c.ph.parent_pdf[g]*ln*ln, rtol=2e-4), np.abs(pdf/(c.ph.parent_pdf[g]*ln*ln)-1).max())
tr = np.exp(-float(c.m.sigma_t[0]) * ln)
want = c.ph.prefix_w[g] * (f / pdf[:, None]) * c.ph.parent_rr[g][:, None] * (tr / c.ph.edge_pdf[g])[:, None]
print("flux ok", np.allclose(c.ph.flux[g], want, rtol=4e-4), np.abs(c.ph.flux[g]/want-1).max())
ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64)
print(cnt)
