#!/usr/bin/env python3
"""Design prototype (numpy, CPU) of the multigrid-preconditioned CG in csrc/kernels_mg.hip.

Not product code and not part of the test-suite: it is the experiment that fixed the design
(coarse-cell typing, trilinear transfer, damped-Jacobi sweeps and their weights, per-slab variant)
before the HIP kernels were written.  It assembles real pressure systems with the oracle
(test infrastructure) and prints PCG iteration counts:

  256^3 scene (700k unknowns): Jacobi 562 | V(2,2) Jacobi w=(2/3,2/3) 27 | w=(2/3,1.2) 22 | RBGS nu=2 19
  per-slab V-cycle (block preconditioner, multi-GPU): 83 / 118 / 133 iterations for 2 / 4 / 8 slabs

Run: python tests/experiments/mg_prototype.py [n]   (n = 128 by default; 256 takes a few minutes)
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as entry
fs = entry.load_package(); oracle = entry.load_oracle()

def build(n, ppc, steps=0, seed=0):
    pos = fs.water_cube_drop(n, ppc, seed)
    o = oracle.Oracle(n=n); o.set_particles(pos, np.random.default_rng(1).standard_normal(pos.shape))
    for _ in range(steps): o.step()
    o.p2g(); o.flags_index(); o.rhs_div(); o.build_matrix()
    idx = o.field(4); solid = o.field(9) != 0
    b = o.field(6).astype(np.float64)
    typ = np.where(solid, 0, np.where(idx >= 0, 2, 1)).astype(np.int8)   # 0 solid 1 air 2 fluid
    return typ, b * (typ == 2), o.dt

def pad(a, v=0):
    return np.pad(a, 1, constant_values=v)

class Level:
    def __init__(self, typ, scale):
        self.typ = typ; self.scale = scale
        f = (typ == 2)
        ns = pad((typ != 0).astype(np.float64), 1.0)   # off-grid = not solid (background)
        cnt = ns[:-2,1:-1,1:-1]+ns[2:,1:-1,1:-1]+ns[1:-1,:-2,1:-1]+ns[1:-1,2:,1:-1]+ns[1:-1,1:-1,:-2]+ns[1:-1,1:-1,2:]
        self.f = f & (cnt > 0)
        self.diag = np.where(self.f, cnt * scale, 1.0)
        self.inv = np.where(self.f, 1.0 / self.diag, 0.0)
        x, y, z = np.indices(typ.shape)
        self.red = ((x + y + z) & 1) == 0
    def nbsum(self, u):
        p = pad(u * self.f)
        return p[:-2,1:-1,1:-1]+p[2:,1:-1,1:-1]+p[1:-1,:-2,1:-1]+p[1:-1,2:,1:-1]+p[1:-1,1:-1,:-2]+p[1:-1,1:-1,2:]
    def A(self, u):
        return np.where(self.f, self.diag * u - self.scale * self.nbsum(u), 0.0)
    def gs_half(self, u, rhs, color):
        m = self.f & (self.red == color)
        u[m] = ((rhs + self.scale * self.nbsum(u)) * self.inv)[m]
    def jacobi(self, u, rhs, w=2/3):
        u += w * self.inv * (rhs - self.A(u))

def coarsen_type(t):
    s = [(d + 1) // 2 * 2 for d in t.shape]
    tp = np.zeros(s, dtype=np.int8); tp[:t.shape[0], :t.shape[1], :t.shape[2]] = t   # pad with solid
    c = tp.reshape(s[0]//2, 2, s[1]//2, 2, s[2]//2, 2)
    any_air = (c == 1).any(axis=(1, 3, 5)); all_solid = (c == 0).all(axis=(1, 3, 5))
    return np.where(any_air, 1, np.where(all_solid, 0, 2)).astype(np.int8)

W = np.array([0.25, 0.75, 0.75, 0.25])
def restrict(r, cshape):     # R = (1/8) P^T, P trilinear cell-centred
    s = [2 * d for d in cshape]
    rp = np.zeros([d + 2 for d in s]); rp[1:1+r.shape[0], 1:1+r.shape[1], 1:1+r.shape[2]] = r
    out = np.zeros(cshape)
    for a in range(4):
        for b in range(4):
            for c in range(4):
                out += W[a]*W[b]*W[c] * rp[a:a+s[0]:2, b:b+s[1]:2, c:c+s[2]:2]
    return out / 8.0
def prolong(e, fshape):
    ep = pad(e)
    s = [2 * d for d in e.shape]
    out = np.zeros([d + 2 for d in s])
    for a in range(4):
        for b in range(4):
            for c in range(4):
                out[a:a+s[0]:2, b:b+s[1]:2, c:c+s[2]:2] += W[a]*W[b]*W[c] * e
    return out[1:1+fshape[0], 1:1+fshape[1], 1:1+fshape[2]]
def prolong_const(e, fshape):
    return np.repeat(np.repeat(np.repeat(e, 2, 0), 2, 1), 2, 2)[:fshape[0], :fshape[1], :fshape[2]]
def restrict_const(r, cshape):
    s = [2 * d for d in cshape]
    rp = np.zeros(s); rp[:r.shape[0], :r.shape[1], :r.shape[2]] = r
    return rp.reshape(cshape[0], 2, cshape[1], 2, cshape[2], 2).sum(axis=(1, 3, 5)) / 8.0 * 2   # overcorrection x2? tune

def vcycle(levels, l, rhs, nu=1, smoother="gs", transfer="tri"):
    L = levels[l]
    u = np.zeros_like(rhs)
    if l == len(levels) - 1:
        for _ in range(20):
            L.gs_half(u, rhs, True); L.gs_half(u, rhs, False)
        for _ in range(20):
            L.gs_half(u, rhs, False); L.gs_half(u, rhs, True)
        return u
    for _ in range(nu):
        if smoother == "gs": L.gs_half(u, rhs, True); L.gs_half(u, rhs, False)
        else: L.jacobi(u, rhs)
    res = np.where(L.f, rhs - L.A(u), 0.0)
    C = levels[l + 1]
    rc = (restrict if transfer == "tri" else restrict_const)(res, C.typ.shape) * C.f
    ec = vcycle(levels, l + 1, rc, nu, smoother, transfer)
    u += (prolong if transfer == "tri" else prolong_const)(ec, rhs.shape) * L.f
    for _ in range(nu):
        if smoother == "gs": L.gs_half(u, rhs, False); L.gs_half(u, rhs, True)
        else: L.jacobi(u, rhs)
    return u

def pcg(L, b, M, tol=2.220446049250313e-16, maxit=2000):
    x = np.zeros_like(b); r = b.copy(); bb = (b*b).sum(); thr = tol*tol*bb
    z = M(r); p = z.copy(); rz = (r*z).sum(); it = 0
    hist = []
    while it < maxit:
        q = L.A(p); al = rz / (p*q).sum(); x += al*p; r -= al*q
        rr = (r*r).sum(); hist.append(np.sqrt(rr/bb))
        if rr < thr: break
        z = M(r); rzn = (r*z).sum(); p = z + (rzn/rz)*p; rz = rzn; it += 1
    return x, it, hist




def jac_w(L, u, rhs, w):
    u += w * L.inv * (rhs - L.A(u))


def vcycle_w(levels, l, rhs, wts, csweeps=12):
    L = levels[l]
    u = np.zeros_like(rhs)
    if l == len(levels) - 1:
        for _ in range(csweeps):
            L.gs_half(u, rhs, True); L.gs_half(u, rhs, False)
        for _ in range(csweeps):
            L.gs_half(u, rhs, False); L.gs_half(u, rhs, True)
        return u
    for w in wts:
        jac_w(L, u, rhs, w)
    res = np.where(L.f, rhs - L.A(u), 0.0)
    C = levels[l + 1]
    ec = vcycle_w(levels, l + 1, restrict(res, C.typ.shape) * C.f, wts, csweeps)
    u += prolong(ec, rhs.shape) * L.f
    for w in reversed(wts):
        jac_w(L, u, rhs, w)
    return u


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    typ, b, dt = build(n, 2 if n <= 128 else 1, 0)
    w = np.argwhere(typ == 2); lo = np.maximum(w.min(0) - 1, 0); hi = np.minimum(w.max(0) + 2, n)
    t0 = typ[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]; b0 = b[lo[0]:hi[0], lo[1]:hi[1], lo[2]:hi[2]]
    levels = [Level(t0, dt)]
    while max(levels[-1].typ.shape) > 8:
        levels.append(Level(coarsen_type(levels[-1].typ), levels[-1].scale / 4))
    L = levels[0]
    print(f"n={n} box={t0.shape} unknowns={int(L.f.sum())} levels={[l.typ.shape for l in levels]}")
    xj, itj, _ = pcg(L, b0, lambda r: r * L.inv)
    print("Jacobi-CG iterations", itj)
    for wts in ((2 / 3, 2 / 3), (2 / 3, 1.2)):
        xm, itm, _ = pcg(L, b0, lambda r: vcycle_w(levels, 0, r, wts))
        print(f"MG-PCG V(2,2) damped Jacobi weights {wts}: iterations {itm}, |x - x_jacobi|/|x| = {np.linalg.norm(xm - xj) / np.linalg.norm(xj):.1e}")
