import numpy as np, sys
d = np.load(sys.argv[1])
pw, po, box = d["pw"], d["po"], d["box"][0]
f32 = np.float32
L = box[:3].astype(f32); invL = (f32(1) / L).astype(f32)
n_pad = len(pw); T = n_pad // 64
xyz = pw[:, :3]; tags = pw[:, 3].view(np.int32)
lo = np.full((T, 3), np.inf, f32); hi = np.full((T, 3), -np.inf, f32)
for t in range(T):
    v = xyz[t*64:(t+1)*64]; ok = ~np.isnan(v[:, 0])
    if ok.any():
        lo[t] = v[ok].min(0); hi[t] = v[ok].max(0)
cut = f32(np.sqrt(f32(144.0)) + 1e-3)   # approx
hits = []
with np.errstate(invalid="ignore"):
  for I in range(n_pad // 128):
    loI = np.minimum(lo[2*I], lo[2*I+1]); hiI = np.maximum(hi[2*I], hi[2*I+1])
    cI = f32(0.5) * (loI + hiI); hI = f32(0.5) * (hiI - loI)
    P = xyz[I*128:(I+1)*128]; tI = tags[I*128:(I+1)*128]
    for J in range(2*I, T):
        cJ = f32(0.5) * (lo[J] + hi[J]); hJ = f32(0.5) * (hi[J] - lo[J])
        dd = cJ - cI
        sft = np.rint(dd * invL)
        dd = dd - sft * L
        ext = hI + hJ
        reach = np.abs(dd) + ext
        gap = np.maximum(f32(0), np.abs(dd) - ext)
        gap = np.where(np.isnan(gap), f32(0), gap)
        g2 = (gap * gap).sum()
        halfL = f32(0.4999) * L
        general = bool((~(reach < halfL) & ~((cut < halfL) & (reach < L - cut - f32(1e-4) * L))).any())
        if not g2 <= cut * cut:
            continue
        Q = xyz[J*64:(J+1)*64].copy(); tJ = tags[J*64:(J+1)*64]
        if not general:
            Q = Q - sft * L
            D = Q[None] - P[:, None]
        else:
            D = Q[None] - P[:, None]
            D = D - np.rint(D * invL) * L
        r2 = (D * D).sum(-1)
        diag = J <= 2*I + 1
        m = r2 < 0.0064
        if diag:
            m &= tI[:, None] != tJ[None]
        ii, jj = np.nonzero(m)
        for a, b in zip(ii, jj):
            hits.append((I, J, general, sft.tolist(), I*128+a, J*64+b, tI[a], tJ[b], r2[a, b]))
print(len(hits)); [print(h) for h in hits]
