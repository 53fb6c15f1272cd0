"""ctypes doors to the two CPU checkers.  TEST INFRASTRUCTURE ONLY.

* ``load_oracle()``  -> oracle/liblvboracle.so  (our C restatement, oracle/fitch_oracle.c)
* ``load_ref()``     -> oracle/_ref/liblvbref.so (the real reference + oracle/ref_harness.cpp),
                        or ``None`` when it has not been built (no /root/reference and no
                        prebuilt copy).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg import this module.
Nothing under lvb_amd/ does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "liblvboracle.so"
REF_SO = HERE / "_ref" / "liblvbref.so"
REF_BIN = HERE / "_ref" / "lvb_ref"
DROPIN_BIN = HERE / "_ref" / "lvb_dropin"

_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_longp = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_intp = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(target: str = "oracle") -> None:
    subprocess.run(["make", "-s", "-C", str(HERE), target], check=True)


class Node(C.Structure):
    """40-byte node record (reference LVB.h:121-128 / oracle lvbo_node)."""

    _fields_ = [
        ("parent", C.c_long),
        ("left", C.c_long),
        ("right", C.c_long),
        ("changes", C.c_long),
        ("sitestate", C.POINTER(C.c_uint64)),
    ]


_oracle = None


def load_oracle() -> C.CDLL:
    global _oracle
    if _oracle is not None:
        return _oracle
    src_mtime = max((HERE / "fitch_oracle.c").stat().st_mtime, (HERE / "fitch_oracle.h").stat().st_mtime)
    if not ORACLE_SO.exists() or ORACLE_SO.stat().st_mtime < src_mtime:
        build("oracle")
    lib = C.CDLL(str(ORACLE_SO))
    lib.lvbo_words_per_row.restype = C.c_long
    lib.lvbo_words_per_row.argtypes = [C.c_long]
    lib.lvbo_encode_char.restype = C.c_int
    lib.lvbo_encode_char.argtypes = [C.c_char]
    lib.lvbo_encode_row.restype = C.c_int
    lib.lvbo_encode_row.argtypes = [C.c_char_p, C.c_long, C.c_long, _u64p]
    lib.lvbo_variable_columns.restype = C.c_long
    lib.lvbo_variable_columns.argtypes = [C.c_long, C.c_long, C.POINTER(C.c_char_p), C.c_void_p]
    lib.lvbo_min_tree_length.restype = C.c_long
    lib.lvbo_min_tree_length.argtypes = [C.c_long, C.c_long, C.POINTER(C.c_char_p)]
    lib.lvbo_combine.restype = C.c_uint64
    lib.lvbo_combine.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_long)]
    lib.lvbo_tree_bytes.restype = C.c_long
    lib.lvbo_tree_bytes.argtypes = [C.c_long, C.c_long]
    lib.lvbo_treealloc.restype = C.POINTER(Node)
    lib.lvbo_treealloc.argtypes = [C.c_long, C.c_long]
    lib.lvbo_tree_set_topology.restype = None
    lib.lvbo_tree_set_topology.argtypes = [C.POINTER(Node), C.c_long, _longp, _longp, _longp]
    lib.lvbo_ss_init.restype = None
    lib.lvbo_ss_init.argtypes = [C.POINTER(Node), C.c_long, C.c_long, C.c_long, _u64p]
    lib.lvbo_mark_dirty.restype = None
    lib.lvbo_mark_dirty.argtypes = [C.POINTER(Node), C.c_long]
    lib.lvbo_make_dirty_below.restype = None
    lib.lvbo_make_dirty_below.argtypes = [C.POINTER(Node), C.c_long]
    lib.lvbo_treecopy.restype = None
    lib.lvbo_treecopy.argtypes = [C.POINTER(Node), C.POINTER(Node), C.c_long, C.c_long]
    lib.lvbo_getplen.restype = C.c_long
    lib.lvbo_getplen.argtypes = [C.POINTER(Node), C.c_long, C.c_long, C.c_long, C.c_long, _longp]
    lib.lvbo_getplen_sliced.restype = C.c_long
    lib.lvbo_getplen_sliced.argtypes = [C.POINTER(Node), C.c_long, C.c_long, C.c_long, C.c_long, _longp,
                                        C.c_int, C.c_long]
    lib.lvbo_fitch_length_plain.restype = C.c_long
    lib.lvbo_fitch_length_plain.argtypes = [C.c_long, C.c_long, _u64p, _longp, _longp, C.c_long]
    _oracle = lib
    return lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def encode_rows(rows: list[str | bytes]) -> np.ndarray:
    """Encode n text rows (already upper-case, equal length) -> uint64 [n, nwords]."""
    lib = load_oracle()
    rows_b = [r.encode() if isinstance(r, str) else bytes(r) for r in rows]
    m = len(rows_b[0])
    nwords = lib.lvbo_words_per_row(m)
    out = np.zeros((len(rows_b), nwords), dtype=np.uint64)
    for i, r in enumerate(rows_b):
        if len(r) != m:
            raise ValueError("ragged alignment")
        if lib.lvbo_encode_row(r, m, nwords, out[i]) != 0:
            raise ValueError(f"bad base symbol in row {i}")
    return out


def cut_constant_columns(rows: list[bytes]) -> list[bytes]:
    lib = load_oracle()
    n, m = len(rows), len(rows[0])
    arr = (C.c_char_p * n)(*rows)
    keep = np.zeros(m, dtype=np.uint8)
    lib.lvbo_variable_columns(n, m, arr, keep.ctypes.data)
    idx = np.nonzero(keep)[0]
    return [bytes(np.frombuffer(r, dtype=np.uint8)[idx]) for r in rows]


def min_tree_length(rows: list[bytes]) -> int:
    lib = load_oracle()
    n, m = len(rows), len(rows[0])
    arr = (C.c_char_p * n)(*rows)
    return int(lib.lvbo_min_tree_length(n, m, arr))


class OracleTree:
    """A reference-layout tree block driven by the oracle's getplen."""

    def __init__(self, n: int, nwords: int, enc: np.ndarray | None = None):
        self.lib = load_oracle()
        self.n, self.nwords = n, nwords
        self.nbranches = 2 * n - 3
        self.ptr = self.lib.lvbo_treealloc(self.nbranches, nwords)
        if not self.ptr:
            raise MemoryError
        self.todo = np.zeros(max(self.nbranches - n, 1), dtype=np.int64)
        self.root = 0
        if enc is not None:
            self.ss_init(enc)

    def __del__(self):
        if getattr(self, "ptr", None):
            _libc.free(C.cast(self.ptr, C.c_void_p))
            self.ptr = None

    def ss_init(self, enc: np.ndarray) -> None:
        enc = np.ascontiguousarray(enc, dtype=np.uint64)
        assert enc.shape == (self.n, self.nwords)
        self.lib.lvbo_ss_init(self.ptr, self.n, self.nbranches, self.nwords, enc)

    def set_topology(self, parent, left, right, root: int) -> None:
        p = np.ascontiguousarray(parent, dtype=np.int64)
        l = np.ascontiguousarray(left, dtype=np.int64)
        r = np.ascontiguousarray(right, dtype=np.int64)
        self.lib.lvbo_tree_set_topology(self.ptr, self.nbranches, p, l, r)
        self.root = int(root)

    def mark_all_dirty(self) -> None:
        for i in range(self.n, self.nbranches):
            self.lib.lvbo_mark_dirty(self.ptr, i)

    def mark_dirty(self, nodes) -> None:
        for i in nodes:
            self.lib.lvbo_mark_dirty(self.ptr, int(i))

    def getplen(self) -> int:
        return int(self.lib.lvbo_getplen(self.ptr, self.n, self.nbranches, self.nwords, self.root, self.todo))

    def getplen_sliced(self, nslices: int, slice_words: int) -> int:
        return int(self.lib.lvbo_getplen_sliced(self.ptr, self.n, self.nbranches, self.nwords, self.root,
                                                self.todo, nslices, slice_words))

    def copy_from(self, other: "OracleTree") -> None:
        self.lib.lvbo_treecopy(self.ptr, other.ptr, self.nbranches, self.nwords)
        self.root = other.root

    def topology(self):
        nb = self.nbranches
        parent = np.array([self.ptr[i].parent for i in range(nb)], dtype=np.int64)
        left = np.array([self.ptr[i].left for i in range(nb)], dtype=np.int64)
        right = np.array([self.ptr[i].right for i in range(nb)], dtype=np.int64)
        return parent, left, right

    def changes(self) -> np.ndarray:
        return np.array([self.ptr[i].changes for i in range(self.nbranches)], dtype=np.int64)

    def dirty(self) -> np.ndarray:
        return np.array([self.ptr[i].sitestate[0] == 0 for i in range(self.nbranches)], dtype=bool)

    def sets(self, node: int) -> np.ndarray:
        return np.ctypeslib.as_array(self.ptr[node].sitestate, shape=(self.nwords,)).copy()

    def all_sets(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.ptr[0].sitestate, shape=(self.nbranches, self.nwords)).copy()


# --------------------------------------------------------------------------- real reference

_ref = None


def have_ref() -> bool:
    return REF_SO.exists()


def load_ref():
    """The real reference behind oracle/ref_harness.cpp, or None if it is not built."""
    global _ref
    if _ref is not None:
        return _ref
    if not REF_SO.exists():
        if Path("/root/reference/src/TreeEvaluation.c").exists():
            build("ref")
        if not REF_SO.exists():
            return None
    lib = C.CDLL(str(REF_SO))
    vp = C.c_void_p
    lib.refh_new_from_rows.restype = vp
    lib.refh_new_from_rows.argtypes = [C.c_long, C.c_long, C.POINTER(C.c_char_p), C.c_int, C.c_int]
    lib.refh_new_from_file.restype = vp
    lib.refh_new_from_file.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int]
    lib.refh_free.argtypes = [vp]
    lib.refh_dims.argtypes = [vp, _longp]
    lib.refh_set_threads.argtypes = [vp, C.c_int]
    lib.refh_row_text.argtypes = [vp, C.c_long, C.c_char_p]
    lib.refh_enc_row.argtypes = [vp, C.c_long, _u64p]
    lib.refh_reseed.argtypes = [vp, C.c_int]
    lib.refh_random_tree.argtypes = [vp]
    lib.refh_uni.restype = C.c_double
    lib.refh_uni.argtypes = []
    lib.refh_randpint.restype = C.c_long
    lib.refh_randpint.argtypes = [C.c_long]
    if hasattr(lib, "refh_set_topology"):
        lib.refh_set_topology.argtypes = [vp, C.c_int, _longp, _longp, _longp, C.c_long]
    lib.refh_root.restype = C.c_long
    lib.refh_root.argtypes = [vp, C.c_int]
    lib.refh_getplen.restype = C.c_long
    lib.refh_getplen.argtypes = [vp, C.c_int]
    lib.refh_mutate.argtypes = [vp, C.c_int]
    lib.refh_swap.argtypes = [vp]
    lib.refh_arbreroot.restype = C.c_long
    lib.refh_arbreroot.argtypes = [vp]
    lib.refh_get_tree.argtypes = [vp, C.c_int, _longp, _longp, _longp, _longp, _intp]
    lib.refh_get_sets.argtypes = [vp, C.c_int, C.c_long, _u64p]
    lib.refh_tree_block.restype = vp
    lib.refh_tree_block.argtypes = [vp, C.c_int]
    lib.refh_msa.restype = vp
    lib.refh_msa.argtypes = [vp]
    lib.refh_tree_bytes.restype = C.c_long
    lib.refh_tree_bytes.argtypes = [vp]
    lib.refh_layout.argtypes = [_longp]
    lib.refh_treeprint.restype = C.c_int
    lib.refh_treeprint.argtypes = [vp, C.c_int, C.c_char_p]
    lib.refh_row_title.argtypes = [vp, C.c_long, C.c_char_p, C.c_long]
    lib.refh_time_proposals.argtypes = [vp, C.c_int, C.c_long, C.c_long, C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.refh_time_full.restype = C.c_double
    lib.refh_time_full.argtypes = [vp, C.c_long, C.POINTER(C.c_long)]
    _ref = lib
    _prime_reference_statics(lib)
    return lib


PRIME_TAXA = 4096


def _prime_reference_statics(lib) -> None:
    """The reference keeps function-local statics sized by the FIRST alignment it sees
    (`oldparent` in lvb_reroot, TreeOperations.c:585-592, and in mutate_tbr, :360/:445): a later,
    larger alignment in the same process would overrun them.  Touch both once with more taxa
    than any test uses, so one process can host many shapes."""
    n, m = PRIME_TAXA, 4
    rows = [bytes(b"ACGT"[(i >> (2 * k)) & 3] for k in range(m)) for i in range(n)]
    arr = (C.c_char_p * n)(*rows)
    h = lib.refh_new_from_rows(n, m, arr, 1, 1)
    lib.refh_getplen(h, 0)
    lib.refh_arbreroot(h)
    for _ in range(8):
        lib.refh_mutate(h, 2)
    lib.refh_free(h)


class RefRun:
    """One reference data structure + current/proposed trees (oracle/ref_harness.cpp)."""

    NNI, SPR, TBR = 0, 1, 2

    def __init__(self, rows: list[bytes] | None = None, path: str | None = None, fmt: int = 0,
                 seed: int = 1, nproc: int = 1):
        self.lib = load_ref()
        if self.lib is None:
            raise RuntimeError("oracle/_ref/liblvbref.so not available")
        if rows is not None:
            n, m = len(rows), len(rows[0])
            arr = (C.c_char_p * n)(*rows)
            self.h = self.lib.refh_new_from_rows(n, m, arr, seed, nproc)
        else:
            self.h = self.lib.refh_new_from_file(os.fsencode(path), fmt, seed, nproc)
        d = np.zeros(8, dtype=np.int64)
        self.lib.refh_dims(self.h, d)
        (self.n, self.m, self.nwords, self.nbranches, self.min_len, self.nthreads, self.slice,
         self.original_m) = (int(x) for x in d)

    def close(self):
        if self.h:
            self.lib.refh_free(self.h)
            self.h = None

    def set_threads(self, k: int) -> None:
        self.lib.refh_set_threads(self.h, k)
        d = np.zeros(8, dtype=np.int64)
        self.lib.refh_dims(self.h, d)
        self.nthreads, self.slice = int(d[5]), int(d[6])

    def rows(self) -> list[bytes]:
        out = []
        for i in range(self.n):
            buf = C.create_string_buffer(self.m)
            self.lib.refh_row_text(self.h, i, buf)
            out.append(buf.raw[: self.m])
        return out

    def enc(self) -> np.ndarray:
        out = np.zeros((self.n, self.nwords), dtype=np.uint64)
        for i in range(self.n):
            self.lib.refh_enc_row(self.h, i, out[i])
        return out

    def reseed(self, seed: int) -> None:
        self.lib.refh_reseed(self.h, seed)

    def random_tree(self) -> None:
        self.lib.refh_random_tree(self.h)

    def uni(self) -> float:
        return float(self.lib.refh_uni())

    def randpint(self, upper: int) -> int:
        return int(self.lib.refh_randpint(int(upper)))

    def set_topology(self, parent, left, right, root: int, which: int = 0) -> None:
        """Tree `which` := the given arrays (all internal nodes dirty: the next getplen is a full evaluation)."""
        p, l, r = (np.ascontiguousarray(a, dtype=np.int64) for a in (parent, left, right))
        self.lib.refh_set_topology(self.h, which, p, l, r, int(root))

    def root(self, which: int = 0) -> int:
        return int(self.lib.refh_root(self.h, which))

    def getplen(self, which: int = 0) -> int:
        return int(self.lib.refh_getplen(self.h, which))

    def mutate(self, kind: int) -> None:
        self.lib.refh_mutate(self.h, kind)

    def swap(self) -> None:
        self.lib.refh_swap(self.h)

    def arbreroot(self) -> int:
        return int(self.lib.refh_arbreroot(self.h))

    def tree(self, which: int = 0):
        nb = self.nbranches
        p = np.zeros(nb, dtype=np.int64)
        l = np.zeros(nb, dtype=np.int64)
        r = np.zeros(nb, dtype=np.int64)
        ch = np.zeros(nb, dtype=np.int64)
        d = np.zeros(nb, dtype=np.int32)
        self.lib.refh_get_tree(self.h, which, p, l, r, ch, d)
        return p, l, r, ch, d.astype(bool)

    def sets(self, which: int, node: int) -> np.ndarray:
        out = np.zeros(self.nwords, dtype=np.uint64)
        self.lib.refh_get_sets(self.h, which, node, out)
        return out

    def all_sets(self, which: int) -> np.ndarray:
        return np.stack([self.sets(which, i) for i in range(self.nbranches)])

    def tree_block(self, which: int):
        return C.cast(self.lib.refh_tree_block(self.h, which), C.POINTER(Node))

    def titles(self) -> list[bytes]:
        out = []
        for i in range(self.n):
            buf = C.create_string_buffer(256)
            self.lib.refh_row_title(self.h, i, buf, 256)
            out.append(buf.value)
        return out

    def treeprint(self, which: int = 0) -> str:
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".tre") as f:
            assert self.lib.refh_treeprint(self.h, which, os.fsencode(f.name)) == 0
            return Path(f.name).read_text()

    def time_proposals(self, kind: int, reps: int, accept_every: int = 4):
        tg, tm = C.c_double(), C.c_double()
        cs, ds = C.c_long(), C.c_long()
        self.lib.refh_time_proposals(self.h, kind, reps, accept_every, C.byref(tg), C.byref(tm), C.byref(cs),
                                     C.byref(ds))
        return tg.value, tm.value, cs.value, ds.value

    def time_full(self, reps: int):
        cs = C.c_long()
        t = self.lib.refh_time_full(self.h, reps, C.byref(cs))
        return t, cs.value


def layout() -> dict:
    lib = load_ref()
    out = np.zeros(16, dtype=np.int64)
    lib.refh_layout(out)
    keys = ["node_size", "off_parent", "off_left", "off_right", "off_changes", "off_sitestate", "msa_size",
            "off_nthreads", "off_slice", "off_n", "off_nbranches", "off_nwords", "params_size", "off_bytes",
            "off_m"]
    return {k: int(v) for k, v in zip(keys, out)}
