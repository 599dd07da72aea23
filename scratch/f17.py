import numpy as np
rng = np.random.default_rng(12)
F, N, L = 24, 3000, 31.0
pos = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.3, (F, N, 3)), axis=0)
pos = np.mod(pos, L).astype(np.float32)
f = 17
box = np.array([L + 0.01 * f, L, L - 0.02 * f], dtype=np.float32)
p = pos[f]
print("box", box, "max", p.max(0), "min", p.min(0))
for k in range(3):
    print(k, "outside:", (p[:, k] >= box[k]).sum(), "near0:", np.sort(p[:, k])[:3], "nearL:", np.sort(p[:, k])[-3:])
pd = p.astype(np.float64); Ld = box.astype(np.float64)
w = pd - Ld * np.floor(pd * (1.0 / Ld))
wf = w.astype(np.float32)
print("wrapped range", wf.min(0), wf.max(0), (wf >= box).sum(0), (wf < 0).sum(0))
d = wf[None] - wf[:, None]
d = d - box * np.rint(d * (np.float32(1) / box).astype(np.float32))
r2 = (d.astype(np.float32) ** 2).sum(-1)
np.fill_diagonal(r2, 9)
i, j = np.nonzero(r2 < 0.01)
print(i, j, r2[i, j])
# cells
n = N
cell = np.cbrt(8.0 * Ld.prod() / n)
nc = np.clip(2 * np.rint(0.5 * Ld / cell).astype(int), 2, 64)
print("nc", nc, "cell", cell)
