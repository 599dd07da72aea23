import numpy as np, sys
d = np.load(sys.argv[1])
pw, po, box = d["pw"], d["po"], d["box"][0]
n_pad = len(pw); L = box[:3]
tags = pw[:, 3].view(np.int32)
print("valid", (tags >= 0).sum(), "unique tags", len(np.unique(tags[tags >= 0])))
# duplicates / missing
cnt = np.bincount(tags[tags >= 0], minlength=3000)
print("dup tags", np.nonzero(cnt > 1)[0], "missing", np.nonzero(cnt == 0)[0])
xyz = pw[:, :3]
# all pairs float32 per-pair image
bad = []
valid = tags >= 0
idx = np.nonzero(valid)[0]
X = xyz[idx]
dd = X[None] - X[:, None]
invL = (np.float32(1) / L).astype(np.float32)
dd = dd - L * np.rint(dd * invL)
r2 = (dd * dd).sum(-1)
np.fill_diagonal(r2, 9)
i, j = np.nonzero(r2 < 0.0064)
print("close pairs (sorted idx):", [(idx[a], idx[b], tags[idx[a]], tags[idx[b]], r2[a, b]) for a, b in zip(i, j)])
# where do padding rows sit
print("padding rows", np.nonzero(~valid)[0][:10], "...", (~valid).sum())
print("nan rows", np.nonzero(np.isnan(xyz[:, 0]))[0][:5], np.isnan(xyz[:, 0]).sum())
# wrapped vs original consistency
w = po[:, :3].astype(np.float64) - L.astype(np.float64) * np.floor(po[:, :3].astype(np.float64) / L.astype(np.float64))
print("max |pw - wrap(po)|", np.nanmax(np.abs(w[valid] - xyz[valid])))
