"""Loading the committed golden vectors (tests/golden/vectors/*.npz; made by gen_golden.py)."""
from __future__ import annotations

import zlib
from pathlib import Path

import numpy as np

VECTORS = Path(__file__).resolve().parent / "golden" / "vectors"


def crc(a: np.ndarray) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def names(max_words: int | None = None) -> list[str]:
    out = []
    for f in sorted(VECTORS.glob("*.npz")):
        if max_words is not None:
            with np.load(f) as z:
                if int(z["nwords"]) > max_words:
                    continue
        out.append(f.stem)
    return out


class Golden:
    def __init__(self, name: str):
        self.name = name
        z = np.load(VECTORS / f"{name}.npz")
        self.z = z
        self.n, self.m, self.nwords = int(z["n"]), int(z["m"]), int(z["nwords"])
        self.nb = 2 * self.n - 3
        self.min_len_tree = int(z["min_len_tree"])
        self.cases = len(z["kind"])

    def rows(self) -> list[bytes]:
        """Alignment text after the reference's constant-column cut."""
        if "text" in self.z:
            return [bytes(r) for r in self.z["text"]]
        from tests import synth
        n, m, seed = (int(x) for x in self.z["synth"])
        full = synth.treelike_rows(n, m, seed)
        # the stored dims are post-cut: redo the cut the way matchange does
        mat = np.stack([np.frombuffer(r, dtype=np.uint8) for r in full])
        keep = (mat != mat[0]).any(axis=0)
        return [mat[i][keep].tobytes() for i in range(n)]

    def enc(self, encoder) -> np.ndarray:
        e = self.z["enc"] if "enc" in self.z else encoder(self.rows())
        assert crc(e) == int(self.z["enc_crc"]), "encoded alignment differs from the reference's"
        return e

    def case(self, k: int) -> dict:
        z = self.z
        d = {key: z[key][k] for key in ("kind", "base", "left", "right", "root", "dirty", "length", "changes",
                                         "sets_crc", "threads_length")}
        if "sets" in z:
            d["sets"] = z["sets"][k]
        return d
