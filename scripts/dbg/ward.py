import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, cases, oracle_lib as O, indep_statements as I
from gvpm_amd import abi
for scene in ("cbox_ward", "cbox_ward_duer"):
    c = cases.make_case(scene, 20, 16, 20000, 4.0)
    print(scene, c.bsdfs.size, c.bsdfs["kind"], c.bsdfs["exponent"], c.bsdfs["sample_visible"], c.bsdfs["specular_sampling_weight"])
    gl = (c.ph.flags & 3) == abi.GVPM_PARENT_SURFACE_BSDF
    print(" glossy", gl.sum(), np.unique(c.ph.parent_g[gl], return_counts=True), np.unique(c.ph.flags[gl] >> 16, return_counts=True))
    g = np.flatnonzero(gl)[:400]
    d = c.ph.pos[g].astype(np.float64) - c.ph.parent_pos[g]
    ln = np.linalg.norm(d, axis=1); wo = d/ln[:,None]
    I.set_bsdfs(c.bsdfs)
    f, pdf, known = I.phong_world(c.ph.parent_scat[g].astype(np.float64), c.ph.parent_g[g].astype(np.int64), c.ph.parent_n[g].astype(np.float64), c.ph.parent_wi[g].astype(np.float64), wo)
    print(" pdf", np.abs(pdf/(c.ph.parent_pdf[g]*ln*ln)-1).max())
    tr = np.exp(-float(c.m.sigma_t[0]) * ln)
    want = c.ph.prefix_w[g] * (f / pdf[:, None]) * c.ph.parent_rr[g][:, None] * (tr / c.ph.edge_pdf[g])[:, None]
    print(" flux", np.abs(c.ph.flux[g]/want-1).max())
    for k in range(0, 400, 80):
        fo, po = O.bsdf_eval_pdf(c.bsdfs[int(c.ph.parent_g[g][k])], c.ph.parent_scat[g][k], c.ph.parent_n[g][k], c.ph.parent_wi[g][k], wo[k])
        print("   oracle vs numpy", np.abs(fo - f[k]).max(), abs(po - pdf[k]))
    ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64)
    print(" ", cnt)
