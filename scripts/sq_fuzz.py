"""Randomised check of the S(q) / ISF engines' wavevector-set handling (test infrastructure): random
grid subsets (cubic and non-cubic cells, negative indices, shuffled, duplicated, perturbed at the
1e-13 level, single planes and lines), random non-lattice sets; every set through the default
kernels and through the general fp64 sincos kernels, and against numpy on a small system.
    python scripts/sq_fuzz.py [seconds] [seed]"""
import itertools
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from mdhelper_amd import _core

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = bad = 0


def engine_result(kind, q, sizes, pairs, pos, general, n_lags=None):
    for k in ("MDX_SQ_NO_LATTICE",):
        os.environ.pop(k, None)
    if general:
        os.environ["MDX_SQ_NO_LATTICE"] = "1"
    if kind == "sq":
        eng = _core.SqEngine(q, sizes, pairs)
        eng.accumulate(pos)
        out = [eng.result()]
    else:
        eng = _core.IsfEngine(q, sizes, pairs, n_lags, True)
        eng.accumulate(pos)
        out = list(eng.result())
    eng.close()
    os.environ.pop("MDX_SQ_NO_LATTICE", None)
    return out


while time.time() < t_end:
    cases += 1
    L = rng.uniform(8.0, 40.0, 3) if rng.random() < 0.6 else np.full(3, rng.uniform(8.0, 40.0))
    lo = [int(rng.integers(-12, 1)) for _ in range(3)]
    if rng.random() < 0.45:
        # the reference's own grids (n = arange(n_points), structure.py:1376-1381) and q_max-filtered subsets of them:
        # m >= 0, the sets the aligned-block (regular) quad items serve
        lo = [0, 0, 0] if rng.random() < 0.8 else [int(rng.integers(0, 4)) for _ in range(3)]
    hi = [int(l + rng.integers(1, 14)) for l in lo]
    axes = [2 * np.pi * np.arange(l, h) / x for l, h, x in zip(lo, hi, L)]
    q = np.stack(np.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 3)
    style = int(rng.integers(0, 7))
    if style == 1:
        q = q[rng.random(len(q)) < rng.uniform(0.05, 0.9)]
    elif style == 2:
        q = q[np.linalg.norm(q, axis=1) <= rng.uniform(0.3, 1.0) * np.abs(q).max()]
    elif style == 3:
        q = np.concatenate([q, q[rng.integers(0, len(q), max(1, len(q) // 10))]])      # duplicates
    elif style == 4:
        q = q * (1.0 + rng.uniform(-1e-13, 1e-13, q.shape))                            # round-off noise
    elif style == 5:
        q = rng.normal(0, 1.0, (int(rng.integers(1, 300)), 3))                          # no lattice at all
    elif style == 6:
        q = q[:, :] * np.array([1.0, 1.0, 0.0]) if rng.random() < 0.5 else q[q[:, 0] == q[0, 0]]
    if len(q) == 0:
        q = np.zeros((1, 3))
    if len(q) > 6000:
        q = q[rng.choice(len(q), 6000, replace=False)]
    q = q[rng.permutation(len(q))]
    n_groups = int(rng.integers(1, 4))
    sizes = [int(rng.integers(1, 900)) for _ in range(n_groups)]
    N = sum(sizes)
    mode = rng.choice(["none", "partial"]) if n_groups > 1 else "none"
    pairs = ((-1, -1),) if mode == "none" else tuple(itertools.combinations_with_replacement(range(n_groups), 2))
    F = int(rng.integers(1, 6))
    pos = (rng.uniform(-0.5, 1.5, (F, N, 3)) * L).astype(np.float32)
    kind = "sq" if rng.random() < 0.7 else "isf"
    n_lags = int(rng.integers(1, F + 1))
    fast = engine_result(kind, q, sizes if mode != "none" else [N], pairs, pos, False, n_lags)
    slow = engine_result(kind, q, sizes if mode != "none" else [N], pairs, pos, True, n_lags)
    ok = all(np.allclose(a, b, rtol=1e-9, atol=1e-9 * max(np.abs(b).max(), 1e-300)) for a, b in zip(fast, slow))
    if ok and kind == "sq" and N * len(q) < 2e6:
        # numpy: sum over frames of |rho|^2 (mode none) or the pair terms
        ref = np.zeros((len(pairs), len(q)))
        edges = np.concatenate(([0], np.cumsum(sizes)))
        for f in range(F):
            p = pos[f].astype(np.float64)
            if mode == "none":
                rho = np.exp(1j * q @ p.T).sum(axis=1)
                ref[0] += np.abs(rho) ** 2
            else:
                rhos = [np.exp(1j * q @ p[edges[g]:edges[g + 1]].T).sum(axis=1) for g in range(n_groups)]
                for k, (i, j) in enumerate(pairs):
                    ref[k] += np.abs(rhos[i]) ** 2 if i == j else 2 * (rhos[i] * rhos[j].conj()).real
        ok = np.allclose(fast[0], ref, rtol=1e-6, atol=1e-8 * np.abs(ref).max())
    if not ok:
        bad += 1
        print(f"MISMATCH case {cases} seed {seed}: kind={kind} style={style} n_q={len(q)} L={L.tolist()} lo={lo} hi={hi} "
              f"sizes={sizes} mode={mode} F={F} n_lags={n_lags}", flush=True)
    if cases % 100 == 0:
        print(f"{cases} cases, {bad} mismatches", flush=True)
print(f"done: {cases} cases, {bad} mismatches (seed {seed})", flush=True)
sys.exit(1 if bad else 0)
