"""
Small numeric utilities (reference ``src/mdhelper/algorithm/utility.py``).
Only ``get_closest_factors`` (:15-72) touches the hot path (spherical wavevector
surfaces of ``StructureFactor``, reference analysis/structure.py:1383).
"""

from __future__ import annotations

import numpy as np


def get_closest_factors(value: int, n_factors: int, reverse: bool = False) -> np.ndarray:
    """
    ``n_factors`` integers whose product is ``value`` and which lie as close to
    each other as possible, ascending (descending when ``reverse``).

    The reference walks the prime factors greedily (sympy); here the
    factorisation with the smallest spread (then the smallest sum) is found by an
    exhaustive search over divisors, which gives the same answers on the
    reference's test vectors (1000 → 10·10·10, 35904 → 32·33·34,
    73440 → 15·16·17·18) without the sympy dependency.
    """
    value, n_factors = int(value), int(n_factors)
    if value < 1 or n_factors < 1:
        raise ValueError("value and n_factors must be positive.")
    best = None

    def search(remaining, k, lowest, chosen):
        nonlocal best
        if k == 1:
            if remaining >= lowest:
                cand = chosen + [remaining]
                key = (cand[-1] - cand[0], sum(cand))
                if best is None or key < best[0]:
                    best = (key, cand)
            return
        d = lowest
        while d ** k <= remaining:
            if remaining % d == 0:
                search(remaining // d, k - 1, d, chosen + [d])
            d += 1

    search(value, n_factors, 1, [])
    factors = np.array(best[1], dtype=int)
    return factors[::-1] if reverse else factors
