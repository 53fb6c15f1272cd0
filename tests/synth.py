"""Seeded synthetic inputs shared by tests and bench (no files, no reference needed).

Alignment generator "T" of SURVEY.md 8(d): taxon 0 uniform over ACGT, taxon i copies taxon
(i-1)//2 with a 10 % per-site substitution; generator "U": i.i.d. uniform.
"""
from __future__ import annotations

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def treelike_rows(n: int, m: int, seed: int, sub_rate: float = 0.10) -> list[bytes]:
    rng = np.random.default_rng(seed)
    mat = np.empty((n, m), dtype=np.uint8)
    mat[0] = rng.integers(0, 4, size=m, dtype=np.uint8)
    for i in range(1, n):
        src = mat[(i - 1) // 2]
        flip = rng.random(m) < sub_rate
        shift = rng.integers(1, 4, size=m, dtype=np.uint8)
        mat[i] = np.where(flip, (src + shift) % 4, src)
    return [ACGT[mat[i]].tobytes() for i in range(n)]


def uniform_rows(n: int, m: int, seed: int) -> list[bytes]:
    rng = np.random.default_rng(seed)
    return [ACGT[rng.integers(0, 4, size=m)].tobytes() for _ in range(n)]


IUPAC = np.frombuffer(b"ACGTUYRWSKMBDHVNX?-", dtype=np.uint8)


def iupac_rows(n: int, m: int, seed: int) -> list[bytes]:
    """Rows that use every symbol DNAToBinary accepts."""
    rng = np.random.default_rng(seed)
    return [IUPAC[rng.integers(0, len(IUPAC), size=m)].tobytes() for _ in range(n)]
