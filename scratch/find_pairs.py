import numpy as np
rng = np.random.default_rng(12)
F, N, L = 24, 3000, 31.0
pos = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.3, (F, N, 3)), axis=0)
pos = np.mod(pos, L).astype(np.float32)
lengths = np.array([[L + 0.01 * f, L, L - 0.02 * f] for f in range(F)], dtype=np.float32)
for f in range(F):
    p = pos[f].astype(np.float64); Lf = lengths[f].astype(np.float64)
    d = p[None] - p[:, None]
    d -= Lf * np.round(d / Lf)
    r = np.sqrt((d * d).sum(-1))
    i, j = np.nonzero(np.triu(r < 0.08, 1))
    for a, b in zip(i, j):
        print(f, a, b, r[a, b], pos[f, a], pos[f, b], lengths[f])
