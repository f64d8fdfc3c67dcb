"""Glossy surface parents (include/gvpm_hip.h GVPM_PARENT_SURFACE_BSDF, gvpm_upload_bsdfs): photons and
beams whose parent vertex lies on a Phong wall are re-connected through the wall's whole BSDF (diffuseReconnection,
shift_diffuse.cpp:25-47 with src/bsdfs/phong.cpp:121-186) instead of failing their shift.  CPU side: the oracle's
restatement against the independent numpy statement for the three techniques that reconnect, the synthetic host's Phong
walls against both, and the sampling routine against its pdf."""
import numpy as np
import pytest

import cases
import indep_statements as I
import oracle_lib as O
from gvpm_amd import abi
from test_indep_statements import compare, compare_beams
from test_oracle_beams import make_beam_case, TECHS
from test_oracle_vpm import make_vpm_case


def test_the_scene_has_photons_behind_glossy_walls_and_a_table_for_them():
    c = cases.make_case("cbox_phong", 20, 16, 20000, 4.0)
    assert c.bsdfs.size == 2 and (c.bsdfs["kind"] == abi.GVPM_BSDF_PHONG).all()
    pt = c.ph.flags & 3
    gl = pt == abi.GVPM_PARENT_SURFACE_BSDF
    assert gl.sum() > 500 and (pt == abi.GVPM_PARENT_SURFACE).sum() > 500
    # the table index rides in parent_g; such photons classify as diffuse reconnections (gvpm_struct.h:66-100)
    idx = c.ph.parent_g[gl]
    assert set(np.unique(idx)) == {0.0, 1.0}
    assert (((c.ph.flags[gl] >> 2) & 7) == 1).all()
    # componentType of the parent = the lobe that was sampled: EGlossyReflection or EDiffuseReflection
    assert set(np.unique(c.ph.flags[gl] >> 16)) == {0x2, 0x8}
    # energy conservation of the two materials (ensureEnergyConservation, phong.cpp:86-91): kd + ks <= 1
    kd = np.array([c.ph.parent_scat[gl][idx == k][0] for k in (0, 1)])
    assert (kd + c.bsdfs["specular"] <= 1.0).all()
    # m_specularSamplingWeight = lum(ks) / (lum(kd) + lum(ks)), phong.cpp:93-97 (Spectrum::getLuminance, RGB)
    lum = lambda v: v @ np.array([0.212671, 0.715160, 0.072169])
    assert np.allclose(c.bsdfs["specular_sampling_weight"], lum(c.bsdfs["specular"]) / (lum(kd) + lum(c.bsdfs["specular"])), rtol=1e-6)


def test_a_phong_wall_below_roughness_005_is_met_one_component_at_a_time():
    """Round 5.  PathVertex::sampleNext picks ONE component of a Phong surface below roughness 0.05
    (vertex.cpp:160-173 with Phong::sampleComponent, phong.cpp:308-329: exponent > 798) and diffuseReconnection evaluates the
    component the vertex was sampled through (shift_diffuse.cpp:31-44: bRec.component = sampledComponentIndex,
    pdf * pdfComponent).  S-cbox-phong1 (exponents 1500 / 900): every such wall has TWO table entries -- met through its
    specular lobe, met through its diffuse one -- and the photon names the one of its parent's sampled component."""
    c = cases.make_case("cbox_phong1", 20, 16, 20000, 4.0)
    assert c.bsdfs.size == 4 and (c.bsdfs["kind"] == abi.GVPM_BSDF_PHONG).all()
    assert list(c.bsdfs["distribution"]) == [1, 2, 1, 2] and (np.sqrt(2.0 / (2.0 + c.bsdfs["exponent"])) < 0.05).all()
    gl = (c.ph.flags & 3) == abi.GVPM_PARENT_SURFACE_BSDF
    idx = c.ph.parent_g[gl].astype(np.int64)
    assert gl.sum() > 500 and set(np.unique(idx)) == {0, 1, 2, 3}
    # the entry's component is the lobe that was sampled: EGlossyReflection (0x8) through entry 2k, EDiffuseReflection (0x2)
    # through 2k + 1; both classify as diffuse reconnections (roughness 0.037 / 0.047 and infinity > bounceRoughness)
    ctype = c.ph.flags[gl] >> 16
    assert (ctype[idx % 2 == 0] == 0x8).all() and (ctype[idx % 2 == 1] == 0x2).all()
    assert (((c.ph.flags[gl] >> 2) & 7) == 1).all()
    # the component is picked with the specular sampling weight (Phong::sampleComponent): about w of the bounces off a wall
    for k in (0, 2):
        n0, n1 = (idx == k).sum(), (idx == k + 1).sum()
        w = float(c.bsdfs["specular_sampling_weight"][k])
        # (a specular sample under the horizon is lost and the diffuse photons scatter more widely: a loose band)
        assert 0.3 * w < n0 / (n0 + n1) < 1.5 * w + 0.1, (k, n0, n1, w)


@pytest.mark.parametrize("scene", ["cbox_phong", "cbox_phong1"])
def test_the_hosts_phong_bounce_is_weight_times_pdf_equals_eval(scene):
    """A photon stored right behind a Phong bounce: flux = prefix * (f cos / pdf) * rr * (Tr / edgePdf)  (gvpm_accel.h:134-148
    with vertex.cpp:165-171: weight = bsdf->sample() = eval / pdf), where pdf in solid angle = the stored area pdf * len^2
    (vertex.cpp:315-329; a medium successor has no cosine).  Checked with the INDEPENDENT statement of the BRDF.  phong1:
    one component per bounce -- weight = eval_c / (pdf_c pdfComponent), pdf = pdf_c pdfComponent (vertex.cpp:169-173)."""
    c = cases.make_case(scene, 20, 16, 20000, 4.0)
    I.set_bsdfs(c.bsdfs)
    gl = np.flatnonzero((c.ph.flags & 3) == abi.GVPM_PARENT_SURFACE_BSDF)[:400]
    d = c.ph.pos[gl].astype(np.float64) - c.ph.parent_pos[gl]
    ln = np.linalg.norm(d, axis=1)
    wo = d / ln[:, None]
    f, pdf, known = I.phong_world(c.ph.parent_scat[gl].astype(np.float64), c.ph.parent_g[gl].astype(np.int64),
                                  c.ph.parent_n[gl].astype(np.float64), c.ph.parent_wi[gl].astype(np.float64), wo)
    assert known.all()
    # (len and direction from fp32 positions; an exponent of 1500 multiplies the direction's rounding)
    assert np.allclose(pdf, c.ph.parent_pdf[gl] * ln * ln, rtol=2e-4 if scene == "cbox_phong" else 2e-3)
    tr = np.exp(-float(c.m.sigma_t[0]) * ln)
    want = c.ph.prefix_w[gl] * (f / pdf[:, None]) * c.ph.parent_rr[gl][:, None] * (tr / c.ph.edge_pdf[gl])[:, None]
    assert np.allclose(c.ph.flux[gl], want, rtol=4e-4)
    # and the oracle's local-frame restatement says the same as the world-space one
    for k in range(0, 400, 40):
        fo, po = O.phong_eval_pdf(c.bsdfs[int(c.ph.parent_g[gl][k])], c.ph.parent_scat[gl][k], c.ph.parent_n[gl][k],
                                  c.ph.parent_wi[gl][k], wo[k])
        assert np.allclose(fo, f[k], rtol=1e-12) and abs(po - pdf[k]) < 1e-12 * pdf[k]


@pytest.mark.parametrize("scene", ["cbox_phong", "cbox_phong_hg", "cbox_conductor", "cbox_phong1"])
def test_bre3d_all_27_accumulators(scene):
    c = cases.make_case(scene, 20, 16, 6000 if scene != "cbox_conductor" else 20000, 4.0)
    cnt = compare(c)
    assert cnt["diffuse_shifts"] > 300
    # without the table the same photons fail their shifts -- the state of affairs before round 4
    O.set_bsdfs(c.bsdfs[:0])
    _, cnt0, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    cases.use_bsdfs(c)
    assert cnt0["failed_shifts"] - cnt["failed_shifts"] == cnt["diffuse_shifts"] - cnt0["diffuse_shifts"] > 100


@pytest.mark.parametrize("kw", [dict(use_mis=0), dict(power_heuristic=1), dict(use_shift_null=0), dict(bsdf_interaction_mode=0x8),
                                dict(bsdf_interaction_mode=0x2)])
def test_bre3d_flags(kw):
    # (bsdfInteractionMode = glossy keeps only the photons whose parent sampled its glossy lobe: a larger map for that one)
    c = cases.make_case("cbox_phong", 16, 12, 60000 if kw.get("bsdf_interaction_mode") == 0x8 else 5000, 4.0, **kw)
    compare(c)


@pytest.mark.parametrize("tech", TECHS)
@pytest.mark.parametrize("scene", ["cbox_phong", "cbox_conductor", "cbox_phong1"])
def test_beams_all_27_accumulators(tech, scene):
    c = make_beam_case(scene, 12, 10, 1500 if scene == "cbox_conductor" else 600, 5.0, technique=tech)
    assert ((c.beams.flags & 3) == abi.GVPM_PARENT_SURFACE_BSDF).sum() > 30
    compare_beams(c)


@pytest.mark.parametrize("scene", ["cbox_phong", "cbox_conductor", "cbox_phong1"])
def test_vpm_all_27_accumulators(scene):
    c = make_vpm_case(scene, 12, 10, 20000 if scene == "cbox_conductor" else 6000, 8.0, 6)
    ref, rsv, rnv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=True)
    acc, icnt, mvol = I.vpm_full(c)
    assert cnt["evaluations"] > 300
    for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert icnt[k] == cnt[k], (k, icnt, cnt)
    lum = ref[..., 0:3].mean()
    for j in range(9):
        assert np.abs(acc[..., 3 * j:3 * j + 3] - ref[..., 3 * j:3 * j + 3]).max() / lum < 1e-9, j


@pytest.mark.parametrize("component", [-1, 0, 1])
def test_phong_sampling_matches_its_pdf_chi_square(component):
    """src/tests/test_chisquare.cpp (test01_BSDF: every component of the BSDF, then all of them) for the "phong" instance of
    data/tests/test_bsdf.xml (diffuse 0.2, specular 0.4, the default exponent 30): for 10 incident directions the histogram
    of Phong::sample over 10 x 20 (theta, phi) bins against the integral of Phong::pdf over the bins; bins with an expected
    frequency below 5 are pooled (libcore/chisquare.cpp), significance 0.01 with the Sidak correction.  One component: the
    table entry's pdf carries pdfComponent (w or 1 - w), the sampler's density is the entry's pdf over it."""
    from scipy import stats
    b = np.zeros(1, abi.BSDF_DTYPE)
    b["kind"], b["specular"], b["exponent"] = abi.GVPM_BSDF_PHONG, 0.4, 30.0
    b["specular_sampling_weight"] = 0.4 / 0.6
    b["distribution"] = component + 1
    pdf_component = {-1: 1.0, 0: 0.4 / 0.6, 1: 1.0 - 0.4 / 0.6}[component]
    kd = np.full(3, 0.2)
    n = np.array([0.0, 0.0, 1.0])
    rng = np.random.default_rng(11)
    theta_bins, phi_bins, n_wi, n_samples = 10, 20, 10, 40000
    alpha = 1.0 - (1.0 - 0.01) ** (1.0 / n_wi)
    for _ in range(n_wi):
        z = 0.05 + 0.95 * rng.random()
        ph = 2 * np.pi * rng.random()
        wi = np.array([np.sqrt(1 - z * z) * np.cos(ph), np.sqrt(1 - z * z) * np.sin(ph), z])
        wos = [O.phong_sample(b[0], n, wi, *rng.random(2)) for _ in range(n_samples)]
        below = sum(w is None for w in wos)  # a lobe sample under the horizon: Phong::sample returns 0 (weight lost)
        wo = np.array([w for w in wos if w is not None])
        theta = np.arccos(np.clip(wo[:, 2], -1, 1))
        phi = np.arctan2(wo[:, 1], wo[:, 0]) % (2 * np.pi)
        obs, _, _ = np.histogram2d(theta, phi, bins=[theta_bins, phi_bins], range=[[0, np.pi], [0, 2 * np.pi]])
        sub = 16
        th = (np.arange(theta_bins * sub) + 0.5) * (np.pi / (theta_bins * sub))
        phs = (np.arange(phi_bins * sub) + 0.5) * (2 * np.pi / (phi_bins * sub))
        T, Pm = np.meshgrid(th, phs, indexing="ij")
        dirs = np.stack([np.sin(T) * np.cos(Pm), np.sin(T) * np.sin(Pm), np.cos(T)], -1).reshape(-1, 3)
        I.set_bsdfs(b)  # (the independent statement reads its module table; the oracle's pdf is spot-checked against it)
        _, pdf, _ = I.phong_world(kd[None, :], np.zeros(len(dirs), np.int64), np.broadcast_to(n, dirs.shape),
                                  np.broadcast_to(wi, dirs.shape), dirs)
        for k in (37, 5000, 20011):
            assert abs(pdf[k] - O.phong_eval_pdf(b[0], kd, n, wi, dirs[k])[1]) < 1e-12 + 1e-12 * pdf[k]
        pdf = pdf.reshape(theta_bins * sub, phi_bins * sub) / pdf_component
        cell = np.sin(T) * (np.pi / (theta_bins * sub)) * (2 * np.pi / (phi_bins * sub))
        exp_ = (pdf * cell).reshape(theta_bins, sub, phi_bins, sub).sum((1, 3)) * n_samples
        # the pdf integrates to 1 minus the part of the lobe under the horizon, which the sampler loses
        assert abs(exp_.sum() - (n_samples - below)) < 5 * np.sqrt(n_samples) + 0.01 * n_samples
        o, e = obs.ravel(), exp_.ravel()
        order = np.argsort(e)
        o, e = o[order], e[order]
        cum = np.cumsum(e)
        k = int(np.searchsorted(cum, 5.0)) + 1
        o = np.concatenate([[o[:k].sum()], o[k:]])
        e = np.concatenate([[e[:k].sum()], e[k:]])
        chi2 = ((o - e) ** 2 / e).sum()
        pval = 1 - stats.chi2.cdf(chi2, df=e.size - 1)
        assert pval > alpha, (wi, chi2, pval)


# ---- the table's second kind: the rough conductor (src/bsdfs/roughconductor.cpp + microfacet.h, isotropic Beckmann / GGX) ----
def test_the_conductor_scene_and_its_table():
    c = cases.make_case("cbox_conductor", 20, 16, 20000, 4.0)
    assert c.bsdfs.size == 2 and (c.bsdfs["kind"] == abi.GVPM_BSDF_ROUGHCONDUCTOR).all()
    assert list(c.bsdfs["distribution"]) == [abi.GVPM_MICROFACET_BECKMANN, abi.GVPM_MICROFACET_GGX] and not c.bsdfs["sample_visible"].any()
    gl = (c.ph.flags & 3) == abi.GVPM_PARENT_SURFACE_BSDF
    assert gl.sum() > 800 and set(np.unique(c.ph.parent_g[gl])) == {0.0, 1.0}
    assert (((c.ph.flags[gl] >> 2) & 7) == 1).all()          # roughness alpha = 0.3 / 0.2 > bounceRoughness: diffuse shifts
    assert set(np.unique(c.ph.flags[gl] >> 16)) == {0x8}     # one component: EGlossyReflection


def test_the_hosts_conductor_bounce_is_weight_times_pdf_equals_eval():
    """as for Phong above: flux = prefix * (f cos / pdf) * rr * (Tr / edgePdf) with RoughConductor::sample's weight = eval / pdf
    (roughconductor.cpp:321-365, sampleVisible = false) -- against the INDEPENDENT statement (textbook D and G, complex Fresnel)"""
    c = cases.make_case("cbox_conductor", 20, 16, 20000, 4.0)
    gl = np.flatnonzero((c.ph.flags & 3) == abi.GVPM_PARENT_SURFACE_BSDF)[:600]
    d = c.ph.pos[gl].astype(np.float64) - c.ph.parent_pos[gl]
    ln = np.linalg.norm(d, axis=1)
    wo = d / ln[:, None]
    f, pdf, known = I.phong_world(c.ph.parent_scat[gl].astype(np.float64), c.ph.parent_g[gl].astype(np.int64),
                                  c.ph.parent_n[gl].astype(np.float64), c.ph.parent_wi[gl].astype(np.float64), wo)
    assert known.all() and (pdf > 0).all()
    assert np.allclose(pdf, c.ph.parent_pdf[gl] * ln * ln, rtol=4e-4)
    tr = np.exp(-float(c.m.sigma_t[0]) * ln)
    want = c.ph.prefix_w[gl] * (f / pdf[:, None]) * c.ph.parent_rr[gl][:, None] * (tr / c.ph.edge_pdf[gl])[:, None]
    assert np.allclose(c.ph.flux[gl], want, rtol=6e-4)
    for k in range(0, 600, 30):
        fo, po = O.bsdf_eval_pdf(c.bsdfs[int(c.ph.parent_g[gl][k])], c.ph.parent_scat[gl][k], c.ph.parent_n[gl][k],
                                 c.ph.parent_wi[gl][k], wo[k])
        assert np.allclose(fo, f[k], rtol=1e-11) and abs(po - pdf[k]) < 1e-11 * pdf[k]


@pytest.mark.parametrize("distribution", [abi.GVPM_MICROFACET_BECKMANN, abi.GVPM_MICROFACET_GGX])
@pytest.mark.parametrize("visible", [0, 1])
def test_conductor_statements_agree_over_the_hemisphere(distribution, visible):
    """oracle restatement (local frame, the reference's operations) == independent statement (world space, textbook forms,
    complex Fresnel) for random directions, both pdf forms, grazing angles included"""
    b = np.zeros(1, abi.BSDF_DTYPE)
    b["kind"], b["specular"], b["exponent"] = abi.GVPM_BSDF_ROUGHCONDUCTOR, (0.9, 0.8, 1.0), 0.15
    b["distribution"], b["sample_visible"], b["eta"], b["k"] = distribution, visible, (0.2, 0.92, 1.1), (3.9, 2.45, 2.14)
    I.set_bsdfs(b)
    rng = np.random.default_rng(3)
    nrm = rng.normal(size=3)
    nrm /= np.linalg.norm(nrm)
    worst = 0.0
    for _ in range(400):
        v = rng.normal(size=(2, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        v = np.where((v @ nrm)[:, None] < 0, -v, v) if rng.random() < 0.9 else v   # (some pairs below the horizon)
        f, pdf, _ = I.phong_world(np.zeros((1, 3)), np.zeros(1, np.int64), nrm[None, :], v[0][None, :], v[1][None, :])
        fo, po = O.bsdf_eval_pdf(b[0], np.zeros(3), nrm, v[0], v[1])
        assert np.allclose(fo, f[0], rtol=1e-10, atol=1e-300) and abs(po - pdf[0]) <= 1e-10 * abs(pdf[0])
        worst = max(worst, float(pdf[0]))
    assert worst > 1.0  # (the lobe was met)


@pytest.mark.parametrize("distribution", [abi.GVPM_MICROFACET_BECKMANN, abi.GVPM_MICROFACET_GGX])
def test_conductor_sampling_matches_its_pdf_chi_square(distribution):
    """src/tests/test_chisquare.cpp (test01_BSDF) as it runs for the roughconductor instances of data/tests/test_bsdf.xml, here
    with sampleVisible = false: the histogram of RoughConductor::sample over 10 x 20 (theta, phi) bins against the integral of
    RoughConductor::pdf; and the returned weight IS eval / pdf."""
    from scipy import stats
    b = np.zeros(1, abi.BSDF_DTYPE)
    b["kind"], b["specular"], b["exponent"] = abi.GVPM_BSDF_ROUGHCONDUCTOR, 1.0, 0.25
    b["distribution"], b["eta"], b["k"] = distribution, (0.2, 0.92, 1.1), (3.9, 2.45, 2.14)
    I.set_bsdfs(b)
    n = np.array([0.0, 0.0, 1.0])
    rng = np.random.default_rng(17)
    theta_bins, phi_bins, n_wi, n_samples = 10, 20, 6, 30000
    alpha = 1.0 - (1.0 - 0.01) ** (1.0 / n_wi)
    for _ in range(n_wi):
        z = 0.1 + 0.9 * rng.random()
        ph = 2 * np.pi * rng.random()
        wi = np.array([np.sqrt(1 - z * z) * np.cos(ph), np.sqrt(1 - z * z) * np.sin(ph), z])
        res = [O.roughconductor_sample(b[0], n, wi, *rng.random(2)) for _ in range(n_samples)]
        lost = sum(r is None for r in res)
        wo = np.array([r[0] for r in res if r is not None])
        for r in [r for r in res if r is not None][:50]:
            fo, po = O.bsdf_eval_pdf(b[0], np.zeros(3), n, wi, r[0])
            assert abs(po - r[2]) < 1e-9 * po and np.allclose(fo / po, r[1], rtol=1e-9)
        theta = np.arccos(np.clip(wo[:, 2], -1, 1))
        phi = np.arctan2(wo[:, 1], wo[:, 0]) % (2 * np.pi)
        obs, _, _ = np.histogram2d(theta, phi, bins=[theta_bins, phi_bins], range=[[0, np.pi], [0, 2 * np.pi]])
        sub = 16
        th = (np.arange(theta_bins * sub) + 0.5) * (np.pi / (theta_bins * sub))
        phs = (np.arange(phi_bins * sub) + 0.5) * (2 * np.pi / (phi_bins * sub))
        T, Pm = np.meshgrid(th, phs, indexing="ij")
        dirs = np.stack([np.sin(T) * np.cos(Pm), np.sin(T) * np.sin(Pm), np.cos(T)], -1).reshape(-1, 3)
        _, pdf, _ = I.phong_world(np.zeros((1, 3)), np.zeros(len(dirs), np.int64), np.broadcast_to(n, dirs.shape),
                                  np.broadcast_to(wi, dirs.shape), dirs)
        pdf = pdf.reshape(theta_bins * sub, phi_bins * sub)
        cell = np.sin(T) * (np.pi / (theta_bins * sub)) * (2 * np.pi / (phi_bins * sub))
        exp_ = (pdf * cell).reshape(theta_bins, sub, phi_bins, sub).sum((1, 3)) * n_samples
        assert abs(exp_.sum() - (n_samples - lost)) < 5 * np.sqrt(n_samples) + 0.01 * n_samples
        o, e = obs.ravel(), exp_.ravel()
        order = np.argsort(e)
        o, e = o[order], e[order]
        k = int(np.searchsorted(np.cumsum(e), 5.0)) + 1
        o = np.concatenate([[o[:k].sum()], o[k:]])
        e = np.concatenate([[e[:k].sum()], e[k:]])
        chi2 = ((o - e) ** 2 / e).sum()
        pval = 1 - stats.chi2.cdf(chi2, df=e.size - 1)
        assert pval > alpha, (wi, chi2, pval)


# ---- Ward (round 5; a material of the bathroom scene BASELINE configs[3] is named after) ------------------------------------
def test_the_ward_scene_and_its_table():
    c = cases.make_case("cbox_ward", 20, 16, 20000, 4.0)
    assert c.bsdfs.size == 2 and (c.bsdfs["kind"] == abi.GVPM_BSDF_WARD).all()
    assert list(c.bsdfs["sample_visible"]) == [abi.GVPM_WARD_BALANCED, abi.GVPM_WARD_WARD] and (c.bsdfs["exponent"] >= 0.05).all()
    gl = (c.ph.flags & 3) == abi.GVPM_PARENT_SURFACE_BSDF
    assert gl.sum() > 500 and set(np.unique(c.ph.parent_g[gl])) == {0.0, 1.0}
    assert (((c.ph.flags[gl] >> 2) & 7) == 1).all() and set(np.unique(c.ph.flags[gl] >> 16)) == {0x2, 0x8}
    d = cases.make_case("cbox_ward_duer", 20, 16, 2000, 4.0)
    assert (d.bsdfs["sample_visible"] == abi.GVPM_WARD_DUER).all()


@pytest.mark.parametrize("scene", ["cbox_ward", "cbox_ward_duer"])
def test_the_hosts_ward_bounce_is_weight_times_pdf_equals_eval(scene):
    """as for Phong: flux = prefix * (f cos / pdf) * rr * (Tr / edgePdf) with Ward::sample's weight = eval / pdf (ward.cpp:321-326),
    against the INDEPENDENT world-space statement; the oracle's local-frame restatement says the same."""
    c = cases.make_case(scene, 20, 16, 20000, 4.0)
    I.set_bsdfs(c.bsdfs)
    gl = np.flatnonzero((c.ph.flags & 3) == abi.GVPM_PARENT_SURFACE_BSDF)[:400]
    d = c.ph.pos[gl].astype(np.float64) - c.ph.parent_pos[gl]
    ln = np.linalg.norm(d, axis=1)
    wo = d / ln[:, None]
    f, pdf, known = I.phong_world(c.ph.parent_scat[gl].astype(np.float64), c.ph.parent_g[gl].astype(np.int64),
                                  c.ph.parent_n[gl].astype(np.float64), c.ph.parent_wi[gl].astype(np.float64), wo)
    assert known.all()
    assert np.allclose(pdf, c.ph.parent_pdf[gl] * ln * ln, rtol=5e-4)
    tr = np.exp(-float(c.m.sigma_t[0]) * ln)
    want = c.ph.prefix_w[gl] * (f / pdf[:, None]) * c.ph.parent_rr[gl][:, None] * (tr / c.ph.edge_pdf[gl])[:, None]
    assert np.allclose(c.ph.flux[gl], want, rtol=4e-4)
    for k in range(0, 400, 40):
        fo, po = O.bsdf_eval_pdf(c.bsdfs[int(c.ph.parent_g[gl][k])], c.ph.parent_scat[gl][k], c.ph.parent_n[gl][k],
                                 c.ph.parent_wi[gl][k], wo[k])
        assert np.allclose(fo, f[k], rtol=1e-11, atol=1e-15) and abs(po - pdf[k]) < 1e-11 * pdf[k]


@pytest.mark.parametrize("scene", ["cbox_ward", "cbox_ward_duer"])
def test_ward_bre3d_vpm_and_beams_against_the_numpy_statements(scene):
    c = cases.make_case(scene, 20, 16, 20000, 4.0)
    compare(c)
    cb = make_beam_case(scene, 12, 10, 600, 5.0)
    assert ((cb.beams.flags & 3) == abi.GVPM_PARENT_SURFACE_BSDF).sum() > 30
    compare_beams(cb)
    cv = make_vpm_case(scene, 12, 10, 6000, 8.0, 6)
    ref, rsv, rnv, cnt, _ = O.gather_vpm(cv.p, cv.m, cv.tris, cv.ph, cv.rays, cv.samples, 64, use_accel=True)
    acc, icnt, mvol = I.vpm_full(cv)
    for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert icnt[k] == cnt[k], (k, icnt, cnt)
    lum = ref[..., 0:3].mean()
    for j in range(9):
        assert np.abs(acc[..., 3 * j:3 * j + 3] - ref[..., 3 * j:3 * j + 3]).max() / lum < 1e-9, j


@pytest.mark.parametrize("variant", [abi.GVPM_WARD_WARD, abi.GVPM_WARD_DUER, abi.GVPM_WARD_BALANCED])
def test_ward_sampling_matches_its_pdf_chi_square(variant):
    """src/tests/test_chisquare.cpp (test01_BSDF) for a Ward instance (diffuse 0.2, specular 0.4, alpha 0.2): the histogram of
    Ward::sample over 10 x 20 (theta, phi) bins against the integral of Ward::pdf (the pdf does not depend on the variant; the
    sampler neither: the variants differ in eval only -- the test also pins that)."""
    from scipy import stats
    b = np.zeros(1, abi.BSDF_DTYPE)
    b["kind"], b["specular"], b["exponent"], b["sample_visible"] = abi.GVPM_BSDF_WARD, 0.4, 0.2, variant
    b["specular_sampling_weight"] = 0.4 / 0.6
    I.set_bsdfs(b)
    kd = np.full(3, 0.2)
    n = np.array([0.0, 0.0, 1.0])
    rng = np.random.default_rng(23)
    theta_bins, phi_bins, n_wi, n_samples = 10, 20, 6, 30000
    alpha = 1.0 - (1.0 - 0.01) ** (1.0 / n_wi)
    for _ in range(n_wi):
        z = 0.1 + 0.9 * rng.random()
        ph = 2 * np.pi * rng.random()
        wi = np.array([np.sqrt(1 - z * z) * np.cos(ph), np.sqrt(1 - z * z) * np.sin(ph), z])
        wos = [O.ward_sample(b[0], n, wi, *rng.random(2)) for _ in range(n_samples)]
        lost = sum(w is None for w in wos)
        wo = np.array([w for w in wos if w is not None])
        theta = np.arccos(np.clip(wo[:, 2], -1, 1))
        phi = np.arctan2(wo[:, 1], wo[:, 0]) % (2 * np.pi)
        obs, _, _ = np.histogram2d(theta, phi, bins=[theta_bins, phi_bins], range=[[0, np.pi], [0, 2 * np.pi]])
        sub = 16
        th = (np.arange(theta_bins * sub) + 0.5) * (np.pi / (theta_bins * sub))
        phs = (np.arange(phi_bins * sub) + 0.5) * (2 * np.pi / (phi_bins * sub))
        T, Pm = np.meshgrid(th, phs, indexing="ij")
        dirs = np.stack([np.sin(T) * np.cos(Pm), np.sin(T) * np.sin(Pm), np.cos(T)], -1).reshape(-1, 3)
        f, pdf, _ = I.phong_world(kd[None, :], np.zeros(len(dirs), np.int64), np.broadcast_to(n, dirs.shape),
                                  np.broadcast_to(wi, dirs.shape), dirs)
        for k in (37, 5000, 20011):
            fo, po = O.bsdf_eval_pdf(b[0], kd, n, wi, dirs[k])
            assert abs(pdf[k] - po) < 1e-12 + 1e-11 * pdf[k] and np.allclose(fo, f[k], rtol=1e-11, atol=1e-15)
        pdf = pdf.reshape(theta_bins * sub, phi_bins * sub)
        cell = np.sin(T) * (np.pi / (theta_bins * sub)) * (2 * np.pi / (phi_bins * sub))
        exp_ = (pdf * cell).reshape(theta_bins, sub, phi_bins, sub).sum((1, 3)) * n_samples
        assert abs(exp_.sum() - (n_samples - lost)) < 5 * np.sqrt(n_samples) + 0.01 * n_samples
        o, e = obs.ravel(), exp_.ravel()
        order = np.argsort(e)
        o, e = o[order], e[order]
        k = int(np.searchsorted(np.cumsum(e), 5.0)) + 1
        o = np.concatenate([[o[:k].sum()], o[k:]])
        e = np.concatenate([[e[:k].sum()], e[k:]])
        chi2 = ((o - e) ** 2 / e).sum()
        pval = 1 - stats.chi2.cdf(chi2, df=e.size - 1)
        assert pval > alpha, (wi, chi2, pval)
