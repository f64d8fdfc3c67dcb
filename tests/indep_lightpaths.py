"""A second, independent implementation (pure Python, float64) of the closed-form scenes' photon shooting: the light-path
random walk with the conventions of Path::randomWalk(EImportance) (src/libbidir/path.cpp:471-501, vertex.cpp:35-332,
edge.cpp:27-84) and the flattening of GPhotonMap::tryAppend (gvpm/gvpm_accel.h:119-199), written from
those and from the stand-in's stream layout below -- it shares no code with gvpm_amd/host/synth_core.h, which both the
host generator and the device generator (synth_device.hip) compile.  tests/test_devgen_gpu.py compares the DEVICE
output with it; tests/test_host.py the host output.

Stream layout of the stand-in (its own specification, not the reference's): path k of iteration `it` draws from
Philox4x32-10 with key (seed, 0x11ff), counter (block, it, k, 0), 24-bit floats, in this order:
  emitter position u1 u2; then per vertex i = 1 .. maxDepth-1:
    emitter sample: two draws, the SECOND cosine-hemisphere argument first;  surface / medium vertex: two draws (a, b)
    Russian roulette (i >= rrDepth): one draw;  free-flight distance: one draw."""
import math

import numpy as np

EPS = 1e-4
MAT_LAMBERT, MAT_NULL, MAT_MIRROR = 0, 1, 2


class Philox:
    def __init__(self, seed, stream, a, b, c=0):
        self.key = (seed & 0xFFFFFFFF, stream & 0xFFFFFFFF)
        self.ctr = [0, a & 0xFFFFFFFF, b & 0xFFFFFFFF, c & 0xFFFFFFFF]
        self.buf = []

    def _refill(self):
        c0, c1, c2, c3 = self.ctr
        k0, k1 = self.key
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
            c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
            k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
        self.buf = [c0, c1, c2, c3]
        self.ctr[0] = (self.ctr[0] + 1) & 0xFFFFFFFF

    def next(self):
        if not self.buf:
            self._refill()
        return (self.buf.pop(0) >> 8) / 16777216.0


class Scene:
    """Read out of the gvpm_devgen_scene the host library exports (arrays of doubles / ints)."""

    def __init__(self, d):
        import ctypes as C
        n, nm = d.n_tris, d.n_mats
        t = np.array((C.c_double * (12 * n)).from_address(d.tris)).reshape(n, 4, 3)
        self.v0, self.e1, self.e2, self.n = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
        self.tri_mat = np.array((C.c_int32 * n).from_address(d.tri_mat))
        self.mat_kind = np.array((C.c_int32 * nm).from_address(d.mat_kind))
        self.mat_albedo = np.array((C.c_double * (3 * nm)).from_address(d.mat_albedo)).reshape(nm, 3)
        self.light_c, self.light_u, self.light_v, self.light_n = (np.array(list(v)) for v in (d.light_c, d.light_u, d.light_v, d.light_n))
        self.radiance, self.light_area = np.array(list(d.radiance)), d.light_area
        self.sig_t, self.sig_s = float(d.medium.sigma_t[1]), np.array([float(x) for x in d.medium.sigma_s])
        self.g, self.msw = float(d.medium.g), float(d.medium.medium_sampling_weight)
        self.cam = np.array(list(d.cam_pos))
        self.seed, self.max_depth, self.rr_depth, self.min_depth = d.seed, d.max_depth, d.rr_depth, d.min_depth
        self.camera_sphere = d.camera_sphere

    def closest(self, o, d, mint):
        p = np.cross(d, self.e2)
        det = (self.e1 * p).sum(1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o - self.v0
            u = (tv * p).sum(1) * inv
            q = np.cross(tv, self.e1)
            v = (q @ d) * inv
            t = (self.e2 * q).sum(1) * inv
        ok = (det != 0) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > mint)
        if not ok.any():
            return None
        t = np.where(ok, t, np.inf)
        k = int(np.argmin(t))   # (the first of equal distances, like a sequential `<` scan)
        return float(t[k]), k


def frame(n):
    """coordinateSystem, src/libcore/util.cpp:487-505: (s, t) with s = t x n"""
    if abs(n[0]) > abs(n[1]):
        inv = 1.0 / math.sqrt(n[0] * n[0] + n[2] * n[2])
        c = np.array([n[2] * inv, 0.0, -n[0] * inv])
    else:
        inv = 1.0 / math.sqrt(n[1] * n[1] + n[2] * n[2])
        c = np.array([0.0, n[2] * inv, -n[1] * inv])
    return np.cross(c, n), c


def to_world(n, loc):
    s, t = frame(n)
    return s * loc[0] + t * loc[1] + n * loc[2]


def hg(g, cos):
    temp = 1 + g * g + 2 * g * cos
    return (1 / (4 * math.pi)) * (1 - g * g) / (temp * math.sqrt(temp))


def walk(sc, it, k):
    """Vertices of light path k: dicts {type, pos, n, weight, rr, pdf, e_weight, e_pdf, e_medium, albedo, kind}."""
    rng = Philox(sc.seed, 0x11FF, it, k & 0xFFFFFFFF, k >> 32)
    V = [dict(type="super", pos=None, n=None, weight=sc.radiance * (math.pi * sc.light_area), rr=1.0, pdf=1 / sc.light_area,
              e_weight=np.ones(3), e_pdf=1.0, e_medium=True, kind=-1, albedo=np.zeros(3))]
    u1, u2 = rng.next(), rng.next()
    V.append(dict(type="emitter", pos=sc.light_c + sc.light_u * (u1 - 0.5) + sc.light_v * (u2 - 0.5), n=sc.light_n, kind=-1,
                  albedo=np.zeros(3), weight=np.zeros(3), rr=1.0, pdf=0.0, e_weight=np.ones(3), e_pdf=1.0, e_medium=False))
    thr = np.ones(3)
    for i in range(1, sc.max_depth):
        cur = V[i]
        mint, solid = EPS, True
        if cur["type"] == "emitter":
            e2, e1 = rng.next(), rng.next()
            r, phi = math.sqrt(e1), 2 * math.pi * e2
            loc = np.array([r * math.cos(phi), r * math.sin(phi), math.sqrt(max(0.0, 1 - e1))])
            wo = to_world(cur["n"], loc)
            cur["weight"], cur["pdf"] = np.ones(3), loc[2] / math.pi        # area light: cosine lobe (area.cpp:132-150)
            if cur["pdf"] <= 0:
                break
        elif cur["type"] == "surface":
            wi = V[i - 1]["pos"] - cur["pos"]
            wi = wi / np.linalg.norm(wi)
            a, b = rng.next(), rng.next()
            if cur["kind"] == MAT_NULL:
                break
            if cur["n"] @ wi <= 0:
                break
            if cur["kind"] == MAT_MIRROR:
                wo = cur["n"] * (2 * (cur["n"] @ wi)) - wi
                cur["weight"], cur["pdf"], solid = cur["albedo"].copy(), 1.0, False
                if cur["weight"].max() <= 0:
                    break
            else:
                r, phi = math.sqrt(a), 2 * math.pi * b
                loc = np.array([r * math.cos(phi), r * math.sin(phi), math.sqrt(max(0.0, 1 - a))])
                wo = to_world(cur["n"], loc)
                cur["weight"], cur["pdf"] = cur["albedo"].copy(), loc[2] / math.pi   # f cos / pdf = albedo (diffuse.cpp)
                if loc[2] <= 0 or cur["weight"].max() <= 0:
                    break
        else:
            wi = V[i - 1]["pos"] - cur["pos"]
            wi = wi / np.linalg.norm(wi)
            a, b = rng.next(), rng.next()
            if abs(sc.g) < EPS:
                z = 1 - 2 * a
                rr_ = math.sqrt(max(0.0, 1 - z * z))
                wo = np.array([rr_ * math.cos(2 * math.pi * b), rr_ * math.sin(2 * math.pi * b), z])
                cur["pdf"] = 1 / (4 * math.pi)
            else:
                g = sc.g
                sq = (1 - g * g) / (1 - g + 2 * g * a)
                ct = (1 + g * g - sq * sq) / (2 * g)
                st = math.sqrt(max(0.0, 1 - ct * ct))
                wo = to_world(-wi, np.array([st * math.cos(2 * math.pi * b), st * math.sin(2 * math.pi * b), ct]))
                cur["pdf"] = hg(g, float(wi @ wo))
            cur["weight"] = sc.sig_s.copy()                                 # sigma_s * phase->sample() (= 1)
            mint = 0.0
        thr = thr * cur["weight"]
        cur["rr"] = 1.0
        if sc.rr_depth != -1 and i >= sc.rr_depth:
            q = min(float(thr.max()), 0.95)
            if rng.next() > q:
                break
            cur["rr"] = 1 / q
            thr = thr * cur["rr"]
        hit = sc.closest(cur["pos"], wo, mint)
        dist_surf = hit[0] if hit else math.inf
        u = rng.next()
        sampled = -math.log(1 - u / sc.msw) / sc.sig_t if u < sc.msw else math.inf
        if sampled < dist_surf:
            ln = sampled
            succ = dict(type="medium", pos=cur["pos"] + wo * ln, n=np.zeros(3), kind=-1, albedo=np.zeros(3))
        elif hit:
            ln = hit[0]
            m = sc.tri_mat[hit[1]]
            succ = dict(type="surface", pos=cur["pos"] + wo * ln, n=sc.n[hit[1]], kind=int(sc.mat_kind[m]), albedo=sc.mat_albedo[m])
        else:
            break
        if ln == 0:
            break
        tr = math.exp(-sc.sig_t * ln)
        if tr < 1e-20:
            break
        cur["e_medium"] = True
        cur["e_pdf"] = sc.sig_t * tr * sc.msw if succ["type"] == "medium" else tr * sc.msw + (1 - sc.msw)
        cur["e_weight"] = np.full(3, tr / cur["e_pdf"])
        thr = thr * cur["e_weight"]
        if solid:                                                           # solid angle -> area (vertex.cpp:315-329)
            cur["pdf"] /= ln * ln
            if succ["type"] == "surface":
                cur["pdf"] *= abs(float(wo @ succ["n"]))
        succ.update(weight=np.zeros(3), rr=1.0, pdf=0.0, e_weight=np.ones(3), e_pdf=1.0, e_medium=False)
        V.append(succ)
    return V


def is_diffuse(sc, v):
    """VertexClassifier::type, gvpm_struct.h:66-79"""
    return v["type"] == "emitter" or (v["type"] == "surface" and v["kind"] == MAT_LAMBERT) or (v["type"] == "medium" and not sc.g > 0.5)


def type_shift(sc, V, c):
    """getTypeShift, shift_utilities.h:112-136 -> 0 invalid, 1 diffuse, 2 medium, 3 manifold"""
    b = -1
    i = c - 1
    while i > 0 and b == -1:
        if is_diffuse(sc, V[i]):
            b = i
        i -= 1
    if b == -1:
        return 0
    if b + 1 == c:
        return 1
    return 2 if V[c - 1]["type"] == "medium" else 3


def camera_hit(sc, a, b):
    """isIntersectedPoint, src/integrators/volume_utils.h:154-169"""
    if sc.camera_sphere == 0:
        return False
    beam = b - a
    l2 = float(beam @ beam)
    if l2 == 0:
        return False
    t = min(1.0, max(0.0, float((sc.cam - a) @ beam) / l2))
    v = (a + beam * t) - sc.cam
    return sc.camera_sphere ** 2 > float(v @ v)


def photons_of(sc, V):
    """GPhotonMap::tryAppend for a volume map, gvpm_accel.h:119-199 -> list of records (dicts of the SoA fields)"""
    out = []
    start = max(2, sc.min_depth + 1)
    if any(V[i]["pdf"] == 0 for i in range(1, len(V) - 1)):     # generatePath() rejects the path, gvpm_proc.cpp:138-143
        return out
    if len(V) <= start:
        return out
    w = np.ones(3)
    for i in range(start - 1):
        w = w * V[i]["weight"] * V[i]["rr"] * V[i]["e_weight"]
    for i in range(start, len(V)):
        prefix = w
        w = w * V[i - 1]["weight"] * V[i - 1]["rr"] * V[i - 1]["e_weight"]
        if V[i]["type"] != "medium" or camera_hit(sc, V[i - 1]["pos"], V[i]["pos"]):
            continue
        par = V[i - 1]
        wi = par["pos"] - V[i]["pos"]
        rec = dict(pos=V[i]["pos"], wi=wi / np.linalg.norm(wi), flux=w, parent_pos=par["pos"], parent_n=par["n"], prefix_w=prefix,
                   parent_scat=np.zeros(3), parent_wi=np.array([1.0, 0, 0]), parent_pdf=par["pdf"], edge_pdf=par["e_pdf"],
                   parent_rr=par["rr"], parent_g=sc.g)
        ptype, comp = 0, 0x00002
        if par["type"] in ("surface", "medium"):
            ptype = 1 if par["type"] == "surface" else 2
            rec["parent_scat"] = par["albedo"] if par["type"] == "surface" else sc.sig_s
            if par["type"] == "surface" and par["kind"] == MAT_MIRROR:
                comp = 0x00008
            pw = V[i - 2]["pos"] - par["pos"]
            rec["parent_wi"] = pw / np.linalg.norm(pw)
        rec["flags"] = ptype | (type_shift(sc, V, i) << 2) | ((1 if par["e_medium"] else 0) << 5) | ((i - 1) << 8) | (comp << 16)
        out.append(rec)
    return out


def shoot_photons(sc, it, capacity):
    """The sequential loop of gvpm_proc.cpp:278-350: paths 0, 1, 2 ... until `capacity` photons are stored.
    -> (dict of arrays, nb_paths)"""
    recs, ids = [], []
    nb_paths = added = k = 0
    while len(recs) < capacity:
        ph = photons_of(sc, walk(sc, it, k))
        k += 1
        nb_paths += 1                              # paths that store nothing count as shot, gvpm_proc.cpp:302-307
        n_app = 0
        for r in ph:
            if len(recs) >= capacity:
                break
            recs.append(r)
            ids.append(added)
            n_app += 1
        if n_app:
            added += 1
    out = {key: np.array([r[key] for r in recs], np.float32) for key in
           ("pos", "wi", "flux", "parent_pos", "parent_n", "prefix_w", "parent_scat", "parent_wi", "parent_pdf", "edge_pdf", "parent_rr", "parent_g")}
    out["flags"] = np.array([r["flags"] for r in recs], np.uint32)
    out["path_id"] = np.array(ids, np.uint32)
    return out, nb_paths
