"""Evaluations per binned pair for (G_i x G_j) particle-group culling on a C2-like frame sorted the way
rdf_cell_sort_kernel sorts (columns a x a, thin layers, serpentine)."""
import numpy as np
rng = np.random.default_rng(0)
N, L, R = 32768, 68.94, 15.0
pos = rng.random((N, 3)) * L
a = (64 * L**3 / N) ** (1 / 3)
nc0 = nc1 = max(int(round(L / a)), 1)
nc2 = max(int(round(16 * L / a)), 1)
c = np.minimum((pos / L * [nc0, nc1, nc2]).astype(int), [nc0 - 1, nc1 - 1, nc2 - 1])
cx, cy, cz = c[:, 0], c[:, 1].copy(), c[:, 2].copy()
cy = np.where(cx & 1, nc1 - 1 - cy, cy)
col = cx * nc1 + cy
cz = np.where(col & 1, nc2 - 1 - cz, cz)
key = col * nc2 + cz
order = np.argsort(key, kind="stable")
P = pos[order]

def boxes(G):
    g = P.reshape(N // G, G, 3)
    return g.min(1), g.max(1)

def gap2(lo_i, hi_i, lo_j, hi_j):
    # minimum-image gap between boxes (centre difference folded)
    ci, hi_ = 0.5 * (lo_i + hi_i), 0.5 * (hi_i - lo_i)
    cj, hj_ = 0.5 * (lo_j + hi_j), 0.5 * (hi_j - lo_j)
    d = cj[None] - ci[:, None]
    d -= L * np.rint(d / L)
    g = np.maximum(0.0, np.abs(d) - hi_[:, None] - hj_[None])
    return (g * g).sum(-1)

# binned (ordered) pairs per particle: 4/3 pi R^3 rho
binned_per_i = 4 / 3 * np.pi * R**3 * N / L**3
for Gi, Gj in [(64, 1), (64, 8), (32, 1), (8, 8), (16, 4), (8, 4), (4, 4), (16, 8)]:
    li, hi = boxes(Gi)
    lj, hj = boxes(Gj)
    sel = rng.choice(len(li), 64, replace=False)
    g2 = gap2(li[sel], hi[sel], lj, hj)
    surv = (g2 <= R * R).sum(1).mean()          # j groups per i group
    evals_per_i = surv * Gj                       # each i particle evaluated against surv*Gj j particles
    print(f"G_i={Gi:3d} G_j={Gj:3d}: evaluations per binned pair {evals_per_i / binned_per_i:.2f}   "
          f"(trips of 64 lanes per frame, unordered: {N * evals_per_i / 64 / 2:.3e})")

print("--- Morton order on fine cells")
def morton_order(cells_per_axis):
    c = np.minimum((pos / L * cells_per_axis).astype(np.int64), cells_per_axis - 1)
    def spread(v):
        out = np.zeros_like(v)
        for b in range(10):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    return np.argsort(spread(c[:, 0]) | (spread(c[:, 1]) << 1) | (spread(c[:, 2]) << 2), kind="stable")
for cpa in (16, 32, 64):
    P = pos[morton_order(cpa)]
    for Gi, Gj in [(64, 1), (8, 8), (8, 4), (16, 4), (4, 4), (32, 2)]:
        li, hi = boxes(Gi); lj, hj = boxes(Gj)
        sel = rng.choice(len(li), 64, replace=False)
        surv = (gap2(li[sel], hi[sel], lj, hj) <= R * R).sum(1).mean()
        print(f"cells/axis {cpa:3d} G_i={Gi:3d} G_j={Gj:3d}: evaluations per binned pair {surv * Gj / binned_per_i:.2f}")


# ---------------------------------------------------------------------------------------------
# Second model: the two 32-lane halves of a wave take different j rows where a row is needed by one
# 32-particle quarter only (rows paired up within a tile visit).
def quarter_model():
    rng = np.random.default_rng(1)
    N, L, R = 32768, 68.94, 15.0
    pos = rng.random((N, 3)) * L
    a = (64 * L**3 / N) ** (1 / 3)
    nc0 = nc1 = max(int(round(L / a)), 1); nc2 = max(int(round(16 * L / a)), 1)
    c = np.minimum((pos / L * [nc0, nc1, nc2]).astype(int), [nc0 - 1, nc1 - 1, nc2 - 1])
    cx, cy, cz = c[:, 0], c[:, 1].copy(), c[:, 2].copy()
    cy = np.where(cx & 1, nc1 - 1 - cy, cy); col = cx * nc1 + cy
    cz = np.where(col & 1, nc2 - 1 - cz, cz)
    P = pos[np.argsort(col * nc2 + cz, kind="stable")]

    def box(g):
        return g.min(0), g.max(0)
    def reach(lo, hi, pts):
        """rows (points) within R of the box, minimum image per component"""
        cen, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
        d = pts - cen; d -= L * np.rint(d / L)
        g = np.maximum(0.0, np.abs(d) - half)
        return (g * g).sum(-1) <= R * R

    old = new = new_union = vis = 0
    tiles = rng.choice(N // 64, 48, replace=False)
    for t in tiles:
        I = P[t * 64:(t + 1) * 64]
        lo, hi = box(I)
        (loA, hiA), (loB, hiB) = box(I[:32]), box(I[32:])
        H = reach(lo, hi, P).reshape(-1, 64)
        A = reach(loA, hiA, P).reshape(-1, 64)
        B = reach(loB, hiB, P).reshape(-1, 64)
        for j in range(N // 64):
            h = H[j].sum()
            if not h:
                continue
            vis += 1
            old += h
            both = (A[j] & B[j]).sum(); xa = (A[j] & ~B[j]).sum(); xb = (B[j] & ~A[j]).sum()
            new += both + max(xa, xb)
            new_union += (A[j] | B[j]).sum()
    print(f"visits per (64-half): {vis / len(tiles):.1f}; rows per visit old {old / vis:.1f}")
    print(f"trips: old {old / len(tiles):.0f} per half-tile; union of quarters {new_union / len(tiles):.0f} "
          f"({new_union / old:.3f}); paired {new / len(tiles):.0f} ({new / old:.3f})")


quarter_model()
