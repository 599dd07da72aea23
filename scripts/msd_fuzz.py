"""Randomised check of the MSD engine against the scipy-FFT restatement (test infrastructure):
random block lengths around every transform-shape boundary, block counts, particle counts,
zeroed dimensions, two groups.    python scripts/msd_fuzz.py [seconds] [seed]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from mdhelper_amd import _core
from oracle import correlation as oc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
t_end = time.time() + budget
cases = bad = 0
edges = [2, 3, 17, 200, 201, 400, 401, 800, 801, 1000, 1600, 1601, 2047, 2048, 2049, 2050, 3200, 3201, 4095, 4096, 4097, 6400, 6401, 8191, 8192, 8193, 12800, 12801, 16383, 16384, 16385,
         25600, 25601, 32767, 32768, 32769, 40000, 51200, 51201, 65536, 65537]
while time.time() < t_end:
    cases += 1
    t_block = int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(2, 70000))
    n_blocks = int(rng.integers(1, 5))
    n_atoms = int(rng.integers(1, 24))
    if t_block <= 800 and rng.random() < 0.7:
        # the single-pass kernel (<= 800 frames per block): many blocks, several pair groups per block of the grid
        n_blocks = int(rng.integers(1, 400))
        n_atoms = int(rng.integers(1, 400))
        if n_blocks * t_block * n_atoms > 6_000_000:
            n_atoms = max(1, 6_000_000 // (n_blocks * t_block))
    if rng.random() < 0.08:
        # the long shapes (round 5: 1024-point rows through msd_fft_rows1024_power_kernel; the sums kernel of 2^18 .. 2^20):
        # 409 600 = 400 x 1024 (131 073 .. 204 800 frames), 2^18, 2^19, 2^20 — one block, a few particles
        t_block = int(rng.choice([102401, 131072, 131073, 150000, 204800, 204801, 262144, 262145, 300000]))
        n_blocks, n_atoms = 1, int(rng.integers(1, 7))
    elif rng.random() < 0.3:
        # rows of whole 128-byte lines (16 | 3 n_atoms): pushes that start inside a line enter their chunk early
        # (the `head` of msd_fft_cols400_fused_kernel, blocks of 32 769 ... 102 400 frames)
        n_atoms = int(rng.choice([16, 32]))
        if rng.random() < 0.7:
            t_block, n_blocks = int(rng.integers(32769, 60000)), 1
    zero_dims = int(rng.choice([0, 0, 1, 2, 4, 5]))
    T = t_block * n_blocks + int(rng.integers(0, 3))
    pos = rng.uniform(0, 50, (1, n_atoms, 3)) + np.cumsum(rng.normal(0, 0.3, (T, n_atoms, 3)), axis=0)
    eng = _core.MsdEngine(t_block, n_blocks, 2)
    first = int(rng.integers(0, n_atoms))
    eng.push(0, pos, 0, n_atoms, zero_dims)
    eng.push(1, pos, first, n_atoms - first, zero_dims)
    msd, traj = eng.result()
    acf = eng.result_acf()
    n_fft = eng.n_fft
    eng.close()
    p = pos[:t_block * n_blocks].reshape(n_blocks, t_block, n_atoms, 3).copy()
    for k in range(3):
        if (zero_dims >> k) & 1:
            p[..., k] = 0
    ok = True
    for g, sl in enumerate((slice(0, n_atoms), slice(first, n_atoms))):
        ref = oc.msd_fft_ref(p[:, :, sl], axis=1, average=False).sum(axis=-1)
        scale = max(np.abs(ref).max(), 1e-300)
        # MSD_m = S_m - 2 A_m cancels ~sum(r^2) / (T - m) at the last lags in the reference too:
        # the absolute floor is a few ulps of that term
        floor = 1e-13 * float((p[:, :, sl] ** 2).sum(axis=(1, 2, 3)).max())
        ok &= np.allclose(msd[g], ref, rtol=1e-7, atol=1e-9 * scale + floor)
        ok &= np.allclose(traj[g], p[:, :, sl].sum(axis=2), rtol=1e-12, atol=1e-9)
        a_ref = oc.correlation_fft_ref(p[:, :, sl], axis=1, vector=True).sum(axis=-1) * (t_block - np.arange(t_block))
        ok &= np.allclose(acf[g], a_ref, rtol=1e-9, atol=1e-9 * max(np.abs(a_ref).max(), 1e-300))
    if not ok:
        bad += 1
        print(f"MISMATCH case {cases} seed {seed}: t_block={t_block} n_blocks={n_blocks} n_atoms={n_atoms} "
              f"zero_dims={zero_dims} first={first} n_fft={n_fft}", flush=True)
    if cases % 50 == 0:
        print(f"{cases} cases, {bad} mismatches", flush=True)
print(f"done: {cases} cases, {bad} mismatches (seed {seed})", flush=True)
sys.exit(1 if bad else 0)
