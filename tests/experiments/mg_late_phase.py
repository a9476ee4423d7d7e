#!/usr/bin/env python3
"""Design experiment (numpy, CPU; not product code, not part of the test-suite): the V-cycle of csrc/kernels_mg.hip on a LATE pressure system
of the 256^3 drop (the settled pool: 1.4 M unknowns in a 254 x 100 x 254 box, dumped on the GPU box with the flags of step 445), where the
product needs 31 iterations per solve against 20 in free fall.  Variants of the smoother near the boundary and of the coarse-cell typing.

Run: python tests/experiments/mg_late_phase.py state445.npz [variant ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib.util
spec = importlib.util.spec_from_file_location("mgp", os.path.join(os.path.dirname(os.path.abspath(__file__)), "mg_prototype.py"))

n = 256
d = np.load(sys.argv[1])
unk = np.unpackbits(d["unk"])[:n**3].reshape(n, n, n).astype(bool)
solid = np.unpackbits(d["solid"])[:n**3].reshape(n, n, n).astype(bool)
fluid = np.unpackbits(d["fluid"])[:n**3].reshape(n, n, n).astype(bool)
b = np.zeros((n, n, n)); b[unk] = d["b"].astype(np.float64)
typ = np.where(solid, 0, np.where(fluid, 2, 1)).astype(np.int8)
w = np.argwhere(unk); lo = np.maximum(w.min(0) - 1, 0); hi = np.minimum(w.max(0) + 2, n)
if os.environ.get("MG_TIGHT"):   # the product's box starts at the first cell inside the walls (its level-0 pairs are (1,2), (3,4), ...)
    lo = np.array([1, 1, 1]); hi = np.minimum(w.max(0) + 1 + int(os.environ["MG_TIGHT"]), n - 1)
t0 = typ[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].copy(); b0 = b[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]].copy()

def pad(a, v=0):
    return np.pad(a, 1, constant_values=v)

class Level:
    def __init__(self, typ, scale):
        self.typ = typ; self.scale = scale
        f = (typ == 2)
        ns = pad((typ != 0).astype(np.float32), 0.0)   # beyond the box: solid, like the arrays of csrc/kernels_mg.hip (type 0 outside the level)
        cnt = ns[:-2,1:-1,1:-1]+ns[2:,1:-1,1:-1]+ns[1:-1,:-2,1:-1]+ns[1:-1,2:,1:-1]+ns[1:-1,1:-1,:-2]+ns[1:-1,1:-1,2:]
        self.f = f & (cnt > 0)
        self.diag = np.where(self.f, cnt * scale, 1.0)
        self.inv = np.where(self.f, 1.0 / self.diag, 0.0)
        # band: unknowns within `width` cells (6-neighbourhood steps) of a non-unknown
        self._band = {}
    def band(self, width):
        if width not in self._band:
            from scipy import ndimage
            self._band[width] = self.f & ~ndimage.binary_erosion(pad(self.f, False), iterations=width)[1:-1,1:-1,1:-1]
        return self._band[width]
    def nbsum(self, u):
        p = pad(u * self.f)
        return p[:-2,1:-1,1:-1]+p[2:,1:-1,1:-1]+p[1:-1,:-2,1:-1]+p[1:-1,2:,1:-1]+p[1:-1,1:-1,:-2]+p[1:-1,1:-1,2:]
    def A(self, u):
        return np.where(self.f, self.diag * u - self.scale * self.nbsum(u), 0.0)

def coarsen_type(t, rule="any_air"):
    s = [(dd + 1) // 2 * 2 for dd in t.shape]
    tp = np.zeros(s, dtype=np.int8); tp[:t.shape[0], :t.shape[1], :t.shape[2]] = t
    c = tp.reshape(s[0]//2, 2, s[1]//2, 2, s[2]//2, 2)
    all_solid = (c == 0).all(axis=(1, 3, 5))
    nair = (c == 1).sum(axis=(1, 3, 5)); nfl = (c == 2).sum(axis=(1, 3, 5))
    if rule == "any_air": air = nair > 0
    elif rule == "half_air": air = nair >= 4
    elif rule == "no_fluid": air = (nfl == 0) & ~all_solid
    return np.where(all_solid, 0, np.where(air, 1, 2)).astype(np.int8)

W = np.array([0.25, 0.75, 0.75, 0.25])
def restrict(r, cshape):
    s = [2 * dd for dd in cshape]
    rp = np.zeros([dd + 2 for dd in s]); rp[1:1+r.shape[0], 1:1+r.shape[1], 1:1+r.shape[2]] = r
    out = np.zeros(cshape)
    for a in range(4):
        for bb in range(4):
            for c in range(4):
                out += W[a]*W[bb]*W[c] * rp[a:a+s[0]:2, bb:bb+s[1]:2, c:c+s[2]:2]
    return out / 8.0
def prolong(e, fshape):
    s = [2 * dd for dd in e.shape]
    out = np.zeros([dd + 2 for dd in s])
    for a in range(4):
        for bb in range(4):
            for c in range(4):
                out[a:a+s[0]:2, bb:bb+s[1]:2, c:c+s[2]:2] += W[a]*W[bb]*W[c] * e
    return out[1:1+fshape[0], 1:1+fshape[1], 1:1+fshape[2]]

def jac(L, u, rhs, w, mask=None):
    du = w * L.inv * (rhs - L.A(u))
    if mask is not None: du = du * mask
    u += du

def vcycle(levels, l, rhs, cfg):
    L = levels[l]
    u = np.zeros_like(rhs)
    if l == len(levels) - 1 or max(L.typ.shape) <= 8:
        for _ in range(30): jac(L, u, rhs, 0.8)
        return u
    wts = cfg.get("wts", (0.56, 1.39))
    bw, bn = cfg.get("band", (0, 0))          # band width, extra sweeps on the band
    maxl = cfg.get("band_levels", 1)
    if bn and l < maxl:
        m = L.band(bw)
        for _ in range(bn): jac(L, u, rhs, cfg.get("band_w", 0.8), m)
    for wq in wts: jac(L, u, rhs, wq)
    res = np.where(L.f, rhs - L.A(u), 0.0)
    C = levels[l + 1]
    ec = vcycle(levels, l + 1, restrict(res, C.typ.shape) * C.f, cfg)
    wc = cfg.get("wc", (1.25, 1.1, 1.0))
    u += wc[min(l, len(wc) - 1)] * prolong(ec, rhs.shape) * L.f
    for wq in reversed(wts): jac(L, u, rhs, wq)
    if bn and l < maxl:
        m = L.band(bw)
        for _ in range(bn): jac(L, u, rhs, cfg.get("band_w", 0.8), m)
    return u

def pcg(L, b, M, tol=2.220446049250313e-16, maxit=80):
    x = np.zeros_like(b); r = b.copy(); bb = (b*b).sum(); thr = tol*tol*bb
    z = M(r); p = z.copy(); rz = (r*z).sum(); it = 0
    while it < maxit:
        q = L.A(p); al = rz / (p*q).sum(); x += al*p; r -= al*q
        rr = (r*r).sum()
        if rr < thr: break
        z = M(r); rzn = (r*z).sum(); p = z + (rzn/rz)*p; rz = rzn; it += 1
    return x, it, np.sqrt(rr / bb)

def build(rule):
    levels = [Level(t0, float(d["dt"]))]
    while max(levels[-1].typ.shape) > 8:
        levels.append(Level(coarsen_type(levels[-1].typ, rule), levels[-1].scale / 4))
    return levels

variants = {
    "base": dict(),
    "band2x1": dict(band=(2, 1)),
    "band2x2": dict(band=(2, 2)),
    "band3x2": dict(band=(3, 2)),
    "band3x3": dict(band=(3, 3)),
    "band2x2_l2": dict(band=(2, 2), band_levels=2),
    "v33": dict(wts=(0.52, 0.8, 1.7)),
    "wc14": dict(wc=(1.4, 1.2, 1.0)),
}
if __name__ == "__main__" and not os.environ.get("MG_GALERKIN") and not os.environ.get("MG_MOD"):
    names = sys.argv[2:] or ["base"]
    rule = os.environ.get("MG_RULE", "any_air")
    levels = build(rule)
    L = levels[0]
    print(f"box {t0.shape} unknowns {int(L.f.sum())} levels {[l.typ.shape for l in levels]} unknowns per level {[int(l.f.sum()) for l in levels]}", flush=True)
    for nm in names:
        cfg = variants[nm]
        t = time.time()
        x, it, rel = pcg(L, b0, lambda r: vcycle(levels, 0, r, cfg))
        print(f"{nm:12s} {cfg}: iterations {it} (rel {rel:.1e}) in {time.time()-t:.0f} s", flush=True)


# ---- variant: Galerkin coarse operators by 2x2x2 aggregation (piecewise-constant P over the unknown children, R = P^T): the free surface
# stays where it is on every level (a coarse cell is an unknown if ANY child is), at the price of per-cell coefficients
class GLevel:
    def __init__(self, f, diag, wx, wy, wz):
        self.f, self.diag, self.wx, self.wy, self.wz = f, diag, wx, wy, wz     # w?: weight of the face to the +? neighbour (0 if none)
        self.inv = np.where(f, 1.0 / np.where(f, diag, 1.0), 0.0)
    def A(self, u):
        out = self.diag * u
        out[:-1] -= self.wx[:-1] * u[1:];   out[1:] -= self.wx[:-1] * u[:-1]
        out[:, :-1] -= self.wy[:, :-1] * u[:, 1:];   out[:, 1:] -= self.wy[:, :-1] * u[:, :-1]
        out[:, :, :-1] -= self.wz[:, :, :-1] * u[:, :, 1:];   out[:, :, 1:] -= self.wz[:, :, :-1] * u[:, :, :-1]
        return out * self.f
    def coarser(self):
        s = [(dd + 1) // 2 * 2 for dd in self.f.shape]
        def padto(a):
            o = np.zeros(s, a.dtype); o[:a.shape[0], :a.shape[1], :a.shape[2]] = a; return o
        f, dg, wx, wy, wz = map(padto, (self.f, self.diag * self.f, self.wx, self.wy, self.wz))
        blk = lambda a: a.reshape(s[0]//2, 2, s[1]//2, 2, s[2]//2, 2)
        fc = blk(f).any(axis=(1, 3, 5))
        # faces inside an aggregate: x faces of the children with even x (their +x neighbour is in the same block), etc.
        inner = blk(wx)[:, 0].sum(axis=(2, 4)) + blk(wy)[:, :, :, 0].sum(axis=(1, 4)) + blk(wz)[:, :, :, :, :, 0].sum(axis=(1, 3))
        dc = blk(dg).sum(axis=(1, 3, 5)) - 2.0 * inner
        cx = blk(wx)[:, 1].sum(axis=(2, 4)); cy = blk(wy)[:, :, :, 1].sum(axis=(1, 4)); cz = blk(wz)[:, :, :, :, :, 1].sum(axis=(1, 3))
        return GLevel(fc, np.where(fc, dc, 1.0), cx, cy, cz)

def g_restrict(r, cshape):
    s = [2 * dd for dd in cshape]
    rp = np.zeros(s); rp[:r.shape[0], :r.shape[1], :r.shape[2]] = r
    return rp.reshape(cshape[0], 2, cshape[1], 2, cshape[2], 2).sum(axis=(1, 3, 5))
def g_prolong(e, fshape):
    return np.repeat(np.repeat(np.repeat(e, 2, 0), 2, 1), 2, 2)[:fshape[0], :fshape[1], :fshape[2]]

def g_vcycle(levels, l, rhs, cfg):
    L = levels[l]
    u = np.zeros_like(rhs)
    if l == len(levels) - 1:
        for _ in range(40): u += 0.8 * L.inv * (rhs - L.A(u))
        return u
    wts = cfg.get("wts", (0.56, 1.39))
    for wq in wts: u += wq * L.inv * (rhs - L.A(u))
    res = (rhs - L.A(u)) * L.f
    C = levels[l + 1]
    ec = g_vcycle(levels, l + 1, g_restrict(res, C.f.shape) * C.f, cfg)
    if cfg.get("cycle", "V") == "W" and l + 1 < len(levels) - 1:
        pass
    u += cfg.get("wc", 1.0) * g_prolong(ec, rhs.shape) * L.f
    for wq in reversed(wts): u += wq * L.inv * (rhs - L.A(u))
    return u

if __name__ == "__main__" and os.environ.get("MG_GALERKIN"):
    L0 = Level(t0, float(d["dt"]))
    f = L0.f
    sc = L0.scale
    wx = np.zeros(f.shape); wx[:-1] = sc * (f[:-1] & f[1:])
    wy = np.zeros(f.shape); wy[:, :-1] = sc * (f[:, :-1] & f[:, 1:])
    wz = np.zeros(f.shape); wz[:, :, :-1] = sc * (f[:, :, :-1] & f[:, :, 1:])
    gl = [GLevel(f, np.where(f, L0.diag, 1.0), wx, wy, wz)]
    while max(gl[-1].f.shape) > 8: gl.append(gl[-1].coarser())
    print("galerkin levels", [g.f.shape for g in gl], [int(g.f.sum()) for g in gl], flush=True)
    chk = np.random.default_rng(0).standard_normal(f.shape) * f
    print("operator check |A_g u - A u| =", np.abs(gl[0].A(chk) - L0.A(chk)).max())
    for wc in [float(a) for a in os.environ["MG_GALERKIN"].split(",")]:
        t = time.time()
        x, it, rel = pcg(gl[0], b0, lambda r: g_vcycle(gl, 0, r, dict(wc=wc)))
        print(f"galerkin aggregation wc {wc}: iterations {it} (rel {rel:.1e}) in {time.time()-t:.0f} s", flush=True)


# ---- variant: rediscretised coarse operators (uniform off-diagonals, trilinear transfers — the product's legs as they are) on NON-eroded coarse
# domains (a coarse cell is an unknown if any child is fluid), the free surface inside a partly filled coarse cell accounted for by a larger
# diagonal: diag = scale (non-solid neighbours + extra(air children))
class MLevel(Level):
    def __init__(self, typ, scale, nair, formula, beta):
        Level.__init__(self, typ, scale)
        k = nair.astype(np.float64)
        extra = beta * k if formula == "lin" else beta * k / np.maximum(8.0 - k, 1.0)
        self.diag = np.where(self.f, self.diag + extra * scale, 1.0)
        self.inv = np.where(self.f, 1.0 / self.diag, 0.0)

def coarsen_mod(t):
    s = [(dd + 1) // 2 * 2 for dd in t.shape]
    tp = np.zeros(s, dtype=np.int8); tp[:t.shape[0], :t.shape[1], :t.shape[2]] = t
    c = tp.reshape(s[0]//2, 2, s[1]//2, 2, s[2]//2, 2)
    nfl = (c == 2).sum(axis=(1, 3, 5)); nair = (c == 1).sum(axis=(1, 3, 5)); all_solid = (c == 0).all(axis=(1, 3, 5))
    return np.where(all_solid, 0, np.where(nfl > 0, 2, 1)).astype(np.int8), nair

if __name__ == "__main__" and os.environ.get("MG_MOD"):
    formula, betas = os.environ["MG_MOD"].split(":")
    for beta in [float(a) for a in betas.split(",")]:
        levels = [Level(t0, float(d["dt"]))]
        nair_acc = None
        while max(levels[-1].typ.shape) > 8:
            tc, nair = coarsen_mod(levels[-1].typ)
            levels.append(MLevel(tc, levels[-1].scale / 4, nair, formula, beta))
        t = time.time()
        x, it, rel = pcg(levels[0], b0, lambda r: vcycle(levels, 0, r, dict(wc=tuple(float(a) for a in os.environ.get("MG_WC", "1.25,1.1,1.0").split(",")))))
        print(f"non-eroded + diag extra {formula} beta {beta}: unknowns per level {[int(l.f.sum()) for l in levels]} iterations {it} (rel {rel:.1e}) in {time.time()-t:.0f} s", flush=True)
