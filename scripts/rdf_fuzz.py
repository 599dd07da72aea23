"""Randomised parity run of the RDF engine against the C restatement (test infrastructure):
random particle counts, cells (orthorhombic and triclinic), ranges, bin counts, exclusions,
self and cross histograms, clustered and uniform configurations.  Run through gpurun:
    python scripts/rdf_fuzz.py [seconds] [seed]
Prints one line per mismatch and a summary; exit code 1 on any mismatch."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from mdhelper_amd import _core
from oracle.cbind import c_radial_histogram

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = bad = 0
while time.time() < t_end:
    cases += 1
    n1 = int(rng.choice([1, 2, 3, 17, 128, 129, 500, 1000, 3000, 6000]))
    same = rng.random() < 0.6
    n2 = n1 if same else int(rng.choice([1, 5, 127, 800, 2500]))
    L = rng.uniform(8.0, 60.0, 3).astype(np.float32)
    tri = rng.random() < 0.3
    ang = rng.uniform(60.0, 120.0, 3).astype(np.float32) if tri else np.full(3, 90.0, np.float32)
    if tri and not (ang.sum() < 355 and all(2 * ang[i] < ang.sum() for i in range(3))):
        ang = np.array([80.0, 95.0, 105.0], np.float32)
    box = np.concatenate([L, ang]).astype(np.float32)
    F = int(rng.integers(1, 4))
    style = rng.integers(0, 3)
    if style == 0:
        pos = rng.random((F, n1 + n2, 3)) * L
    elif style == 1:      # clustered: many pairs at the same and at tiny distances
        centres = rng.random((F, max(1, (n1 + n2) // 20), 3)) * L
        pos = centres[:, rng.integers(0, centres.shape[1], n1 + n2)] + rng.normal(0, 0.05, (F, n1 + n2, 3))
    else:                 # lattice + jitter, coordinates outside the cell too
        pos = (rng.integers(-3, 40, (F, n1 + n2, 3)) * 0.5 + rng.normal(0, 1e-3, (F, n1 + n2, 3)))
    pos = pos.astype(np.float32)
    r_hi = float(rng.uniform(0.5, 0.5 * float(L.min()) * (0.8 if tri else 1.0)))
    if not tri and rng.random() < 0.3:   # up to half the cell: tile pairs that straddle L/2 (per-pair image fold)
        r_hi = 0.5 * float(L.min()) * float(rng.choice([1.0, 0.999, rng.uniform(0.8, 1.0)]))
    r_lo = float(rng.choice([0.0, 0.0, rng.uniform(0, r_hi * 0.5)]))
    n_bins = int(rng.choice([1, 2, 7, 50, 201, 777, 3000]))
    excl = None
    if rng.random() < 0.5:
        excl = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
    p1 = pos[:, :n1]
    p2 = None if same else pos[:, n1:]
    want = np.zeros(n_bins, dtype=np.int64)
    for f in range(F):
        want += c_radial_histogram(p1[f], p1[f] if same else p2[f], n_bins, (r_lo, r_hi), box, exclusion=excl)
    edges = np.linspace(r_lo, r_hi, n_bins + 1)
    for algo in ("auto", "filter", "cell"):
        eng = _core.RdfEngine(edges, excl, algo=algo)
        eng.accumulate(p1, p2, np.tile(box, (F, 1)))
        got = eng.counts()
        eng.close()
        if not np.array_equal(got, want):
            bad += 1
            print(f"MISMATCH case {cases} seed {seed} algo {algo}: n1={n1} n2={n2} same={same} box={box.tolist()} "
                  f"F={F} style={style} range=({r_lo},{r_hi}) n_bins={n_bins} excl={excl} "
                  f"diff={np.abs(got - want).sum()} at {np.flatnonzero(got != want)[:5].tolist()}", flush=True)
    if cases % 200 == 0:
        print(f"{cases} cases, {bad} mismatches", flush=True)
print(f"done: {cases} cases, {bad} mismatches (seed {seed})", flush=True)
sys.exit(1 if bad else 0)
