"""Python face of include/lvbhost.h: topology object, NNI/SPR/TBR proposals as edits, batched SA.

The library (lvb_amd/liblvbhost.so) links liblvbgpu.so; it holds no scoring code.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import api

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "liblvbhost.so"

NNI, SPR, TBR = 0, 1, 2
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class AnnealParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("algorithm", C.c_int32), ("cooling_schedule", C.c_int32),
                ("batch", C.c_int32), ("reroot_interval", C.c_int32), ("t0", C.c_double),
                ("maxaccept", C.c_int64), ("maxpropose", C.c_int64), ("maxfail", C.c_int64),
                ("min_len_tree", C.c_int64), ("max_proposals", C.c_int64), ("max_seconds", C.c_double),
                ("max_device_steps", C.c_int64), ("sync_every", C.c_int32), ("log_cap", C.c_int32),
                ("device_proposals", C.c_int32), ("run_levels", C.c_int32), ("lanes", C.c_int32)]


class AnnealResult(C.Structure):
    _fields_ = [("start_length", C.c_int64), ("best_length", C.c_int64), ("final_length", C.c_int64),
                ("global_best_length", C.c_int64), ("scored", C.c_int64), ("consumed", C.c_int64),
                ("accepted", C.c_int64), ("topologies", C.c_int64), ("device_steps", C.c_int64), ("reroots", C.c_int64),
                ("dirty_nodes", C.c_int64), ("temperatures", C.c_int64), ("t_final", C.c_double),
                ("seconds", C.c_double), ("seconds_device", C.c_double), ("n_log", C.c_int32),
                ("frozen", C.c_int32), ("seconds_done", C.c_double), ("seconds_busy", C.c_double), ("scored_busy", C.c_int64),
                ("host_steps", C.c_int64)]


class RefSearchParams(C.Structure):
    _fields_ = [("seed", C.c_int32), ("algorithm", C.c_int32), ("cooling_schedule", C.c_int32),
                ("max_batch", C.c_int32), ("min_len_tree", C.c_int64), ("max_trees", C.c_int64),
                ("maxaccept", C.c_int64), ("maxpropose", C.c_int64), ("maxfail", C.c_int64),
                ("device_moves_min", C.c_int64), ("reserved", C.c_int64 * 3)]


class RefSearchResult(C.Structure):
    _fields_ = [("t0", C.c_double), ("rearrangements", C.c_int64), ("best_length", C.c_int64), ("trees", C.c_int64),
                ("start_length", C.c_int64), ("final_length", C.c_int64), ("accepted_moves", C.c_int64),
                ("reroots", C.c_int64), ("temperatures", C.c_int64), ("st_rearrangements", C.c_int64),
                ("scored", C.c_int64), ("device_steps", C.c_int64), ("st_scored", C.c_int64),
                ("st_device_steps", C.c_int64), ("device_move_steps", C.c_int64), ("t_final", C.c_double),
                ("seconds", C.c_double),
                ("seconds_device", C.c_double)]


SIGNATURES = {
    "lvbhost_tree_random": (C.c_void_p, [C.c_int32, C.c_uint64]),
    "lvbhost_tree_from_arrays": (C.c_void_p, [C.c_int32, _i32p, _i32p, C.c_int32, C.c_uint64]),
    "lvbhost_tree_free": (None, [C.c_void_p]),
    "lvbhost_tree_n": (C.c_int32, [C.c_void_p]),
    "lvbhost_tree_root": (C.c_int32, [C.c_void_p]),
    "lvbhost_tree_arrays": (None, [C.c_void_p, _i32p, _i32p, _i32p]),
    "lvbhost_tree_reseed": (None, [C.c_void_p, C.c_uint64]),
    "lvbhost_propose": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int32]),
    "lvbhost_propose_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int32, _i32p, C.c_void_p, C.c_int32]),
    "lvbhost_nni_edits": (C.c_int, [C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_int32]),
    "lvbhost_spr_edits": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "lvbhost_tbr_edits": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    "lvbhost_reroot_edits": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    "lvbhost_tree_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "lvbhost_tree_topology_hash": (C.c_uint64, [C.c_void_p]),
    "lvbhost_tree_canonical": (C.c_int32, [C.c_void_p, _i32p, C.c_int32]),
    "lvbhost_tree_best_count": (C.c_int32, [C.c_void_p]),
    "lvbhost_tree_best_kept": (C.c_int32, [C.c_void_p]),
    "lvbhost_tree_best_get": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p, C.POINTER(C.c_int32)]),
    "lvbhost_program": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.POINTER(C.c_int32), C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "lvbhost_alignment_read_phylip": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_int32]),
    "lvbhost_alignment_read": (C.c_void_p, [C.c_char_p, C.c_int, C.c_char_p, C.c_int32]),
    "lvbhost_alignment_free": (None, [C.c_void_p]),
    "lvbhost_alignment_n": (C.c_int64, [C.c_void_p]),
    "lvbhost_alignment_m": (C.c_int64, [C.c_void_p]),
    "lvbhost_alignment_row": (C.c_char_p, [C.c_void_p, C.c_int64]),
    "lvbhost_alignment_name": (C.c_char_p, [C.c_void_p, C.c_int64]),
    "lvbhost_tree_newick": (C.c_int64, [C.c_void_p, C.POINTER(C.c_char_p), C.c_char_p, C.c_int64]),
    "lvbhost_variable_columns": (C.c_int64, [C.c_int64, C.c_int64, C.POINTER(C.c_char_p), C.c_void_p]),
    "lvbhost_min_tree_length": (C.c_int64, [C.c_int64, C.c_int64, C.POINTER(C.c_char_p)]),
    "lvbhost_anneal_defaults": (None, [C.POINTER(AnnealParams)]),
    "lvbhost_anneal": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(AnnealParams), C.POINTER(AnnealResult),
                                 C.c_void_p, C.c_void_p]),
    "lvbhost_anneal_chains": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(AnnealParams),
                                        C.POINTER(AnnealResult), C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "lvbhost_starting_temperature": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(AnnealParams),
                                               C.POINTER(C.c_double)]),
    "lvbhost_tree_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "lvbhost_refrng_new": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32]),
    "lvbhost_refrng_free": (None, [C.c_void_p]),
    "lvbhost_refrng_uni": (C.c_double, [C.c_void_p]),
    "lvbhost_refrng_randpint": (C.c_int64, [C.c_void_p, C.c_int64]),
    "lvbhost_ref_random_tree": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p]),
    "lvbhost_ref_propose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lvbhost_ref_draw_move": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "lvbhost_move_edits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lvbhost_ref_arbreroot": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32)]),
    "lvbhost_refsearch_defaults": (None, [C.POINTER(RefSearchParams)]),
    "lvbhost_reference_search": (C.c_int, [C.c_void_p, C.POINTER(RefSearchParams), C.POINTER(RefSearchResult),
                                           C.POINTER(C.c_void_p)]),
}

_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    api.load_library()  # liblvbhost links liblvbgpu: make the failure message the useful one
    if not LIB_PATH.exists():
        raise api.LvbGpuError(-2, f"{LIB_PATH} is not built (run `python -m lvb_amd.build`)")
    _lib = bind(C.CDLL(str(LIB_PATH)))
    return _lib


def bind(lib: C.CDLL) -> C.CDLL:
    """Declare include/lvbhost.h's signatures on a loaded library."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


TOK_ROW_MASK = 0x00FFFFFF
TOK_MERGE_SHIFT = 24
TOK_MERGE_MASK = 0x3F
TOK_FRESH = 1 << 30
TOK_PUSH = 1 << 31


class HostTree:
    """Topology + random stream (lvbhost_tree)."""

    def __init__(self, n: int | None = None, *, seed: int = 1, left=None, right=None, root: int = 0, handle=None,
                 lib=None):
        self.lib = lib or load_library()
        if handle is not None:  # adopt a tree the library made (lvbhost_reference_search)
            self.h = handle
        elif left is not None:
            l = np.ascontiguousarray(left, dtype=np.int32)
            r = np.ascontiguousarray(right, dtype=np.int32)
            n = (len(l) + 3) // 2
            self.h = self.lib.lvbhost_tree_from_arrays(n, l, r, int(root), seed)
        else:
            self.h = self.lib.lvbhost_tree_random(int(n), seed)
        if not self.h:
            raise ValueError("not a binary tree rooted at a leaf")
        self.n = int(self.lib.lvbhost_tree_n(self.h))
        self.nbranches = 2 * self.n - 3

    def close(self):
        if getattr(self, "h", None):
            self.lib.lvbhost_tree_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def root(self) -> int:
        return int(self.lib.lvbhost_tree_root(self.h))

    def arrays(self):
        p, l, r = (np.zeros(self.nbranches, dtype=np.int32) for _ in range(3))
        self.lib.lvbhost_tree_arrays(self.h, p, l, r)
        return p, l, r

    def reseed(self, seed: int) -> None:
        self.lib.lvbhost_tree_reseed(self.h, seed)

    def _edits(self, fn, *args, cap: int | None = None) -> np.ndarray:
        cap = cap or (2 * self.nbranches + 8)
        buf = np.zeros(cap, dtype=api.EDIT_DTYPE)
        k = fn(self.h, *args, buf.ctypes.data, cap)
        if k < 0:
            raise api.LvbGpuError(k, "proposal rejected")
        return buf[:k].copy()

    def propose(self, kind: int) -> np.ndarray:
        return self._edits(self.lib.lvbhost_propose, kind)

    def propose_batch(self, kind: int, B: int):
        """-> (edit_offsets[B+1], edits) ready for lvbgpu_batch_build (views into reused buffers:
        copy them if they must outlive the next call)."""
        cap = getattr(self, "_cap", 0)
        if cap < B * 8 + 64:
            cap = B * 8 + 64
        while True:
            if getattr(self, "_cap", 0) != cap or len(getattr(self, "_offs", ())) != B + 1:
                self._cap = cap
                self._buf = np.empty(cap, dtype=api.EDIT_DTYPE)
                self._offs = np.empty(B + 1, dtype=np.int32)
            state = self.lib.lvbhost_propose_batch(self.h, kind, B, self._offs, self._buf.ctypes.data, cap)
            if state >= 0:
                return self._offs, self._buf[:state]
            if cap > B * (2 * self.nbranches + 8):
                raise api.LvbGpuError(state, "propose_batch failed")
            cap *= 4  # note: the tree's stream has advanced; callers that need determinism reseed

    def nni_edits(self, u: int, swap_right: bool) -> np.ndarray:
        return self._edits(self.lib.lvbhost_nni_edits, int(u), int(bool(swap_right)))

    def spr_edits(self, src: int, dest: int) -> np.ndarray:
        return self._edits(self.lib.lvbhost_spr_edits, int(src), int(dest))

    def tbr_edits(self, src: int, dest: int, newroot_leaf: int) -> np.ndarray:
        return self._edits(self.lib.lvbhost_tbr_edits, int(src), int(dest), int(newroot_leaf))

    def reroot_edits(self, newroot: int) -> np.ndarray:
        return self._edits(self.lib.lvbhost_reroot_edits, int(newroot))

    def apply(self, edits, new_root: int = -1) -> None:
        e = api._edits_array(edits)
        rc = self.lib.lvbhost_tree_apply(self.h, e.ctypes.data, len(e), int(new_root))
        if rc != 0:
            raise api.LvbGpuError(rc, "edits do not give a tree")

    def program(self, mode: int = 0, edits=None, new_root: int = -1, dirty=None):
        """The token program the device would walk -> dict(toks, dsts, max_stack, dirty)."""
        cap = 2 * self.nbranches + 16
        toks = np.zeros(cap, dtype=np.uint32)
        dsts = np.zeros(cap, dtype=np.int32)
        ntok, ndst, mstack, nd = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        e = api._edits_array(edits) if edits is not None else np.zeros(0, dtype=api.EDIT_DTYPE)
        d = None if dirty is None else np.ascontiguousarray(dirty, dtype=np.uint8)
        rc = self.lib.lvbhost_program(self.h, mode, e.ctypes.data, len(e), int(new_root),
                                      None if d is None else d.ctypes.data, toks.ctypes.data, cap, C.byref(ntok),
                                      dsts.ctypes.data, cap, C.byref(ndst), C.byref(mstack), C.byref(nd))
        if rc != 0:
            raise api.LvbGpuError(rc, "program build failed")
        return {"toks": toks[: ntok.value].copy(), "dsts": dsts[: ndst.value].copy(), "max_stack": mstack.value,
                "dirty": nd.value}

    def best_trees(self) -> list["HostTree"]:
        """The distinct best topologies the last anneal run kept (the reference's treestack)."""
        out = []
        for i in range(int(self.lib.lvbhost_tree_best_kept(self.h))):
            l = np.zeros(self.nbranches, dtype=np.int32)
            r = np.zeros(self.nbranches, dtype=np.int32)
            root = C.c_int32()
            self.lib.lvbhost_tree_best_get(self.h, i, l, r, C.byref(root))
            out.append(HostTree(left=l, right=r, root=root.value, lib=self.lib))
        return out

    def ref_propose(self, rng: "RefRng", kind: int) -> np.ndarray:
        """mutate_nni/spr/tbr with the reference's own draws, as edits."""
        cap = 2 * self.nbranches + 8
        buf = np.zeros(cap, dtype=api.EDIT_DTYPE)
        k = C.c_int32()
        rc = self.lib.lvbhost_ref_propose(self.h, rng.h, int(kind), buf.ctypes.data, cap, C.byref(k))
        if rc != 0:
            raise api.LvbGpuError(rc, "ref_propose")
        return buf[: k.value].copy()

    def ref_draw_move(self, rng: "RefRng", kind: int) -> tuple[int, int, int, int]:
        """Parameters (kind, a, b, c) of one mutate_nni/spr/tbr draw: all of its stream consumption."""
        mv = np.zeros(1, dtype=api.MOVE_DTYPE)
        rc = self.lib.lvbhost_ref_draw_move(self.h, rng.h, int(kind), mv.ctypes.data)
        if rc != 0:
            raise api.LvbGpuError(rc, "ref_draw_move")
        return tuple(int(mv[0][k]) for k in ("kind", "a", "b", "c"))

    def move_edits(self, move) -> np.ndarray:
        mv = np.array([tuple(int(v) for v in move)], dtype=api.MOVE_DTYPE)
        cap = 2 * self.nbranches + 8
        buf = np.zeros(cap, dtype=api.EDIT_DTYPE)
        k = C.c_int32()
        rc = self.lib.lvbhost_move_edits(self.h, mv.ctypes.data, buf.ctypes.data, cap, C.byref(k))
        if rc != 0:
            raise api.LvbGpuError(rc, "move cannot be made on this tree")
        return buf[: k.value].copy()

    def ref_arbreroot(self, rng: "RefRng") -> tuple[np.ndarray, int]:
        cap = 2 * self.nbranches + 8
        buf = np.zeros(cap, dtype=api.EDIT_DTYPE)
        k, nr = C.c_int32(), C.c_int32()
        rc = self.lib.lvbhost_ref_arbreroot(self.h, rng.h, buf.ctypes.data, cap, C.byref(k), C.byref(nr))
        if rc != 0:
            raise api.LvbGpuError(rc, "ref_arbreroot")
        return buf[: k.value].copy(), nr.value

    def topology_hash(self) -> int:
        return int(self.lib.lvbhost_tree_topology_hash(self.h))

    def canonical(self) -> tuple:
        """Exact, rooting- and numbering-independent form of the topology (preorder from taxon 0)."""
        buf = np.zeros(2 * self.nbranches + 4, dtype=np.int32)
        k = self.lib.lvbhost_tree_canonical(self.h, buf, len(buf))
        if k < 0:
            raise api.LvbGpuError(k, "canonical")
        return tuple(int(x) for x in buf[:k])

    def best_count(self) -> int:
        return int(self.lib.lvbhost_tree_best_count(self.h))

    def upload(self, ctx: api.FitchContext) -> int:
        out = C.c_int64()
        ctx._chk(self.lib.lvbhost_tree_upload(ctx.h, self.h, C.byref(out)))
        return out.value


class RefRng:
    """The reference's uni()/randpint() stream (RandomNumberGenerator.c), seeded as rinit() does."""

    def __init__(self, seed: int, lib=None):
        self.lib = lib or load_library()
        h = C.c_void_p()
        rc = self.lib.lvbhost_refrng_new(C.byref(h), int(seed))
        if rc != 0:
            raise api.LvbGpuError(rc, f"seed {seed} is outside 0..900000000")
        self.h = h

    def uni(self) -> float:
        return float(self.lib.lvbhost_refrng_uni(self.h))

    def randpint(self, upper: int) -> int:
        return int(self.lib.lvbhost_refrng_randpint(self.h, int(upper)))

    def random_tree(self, n: int) -> "HostTree":
        """PullRandomTree: rooted at taxon 0."""
        l = np.zeros(2 * n - 3, dtype=np.int32)
        r = np.zeros(2 * n - 3, dtype=np.int32)
        rc = self.lib.lvbhost_ref_random_tree(self.h, int(n), l, r)
        if rc != 0:
            raise api.LvbGpuError(rc, "ref_random_tree")
        return HostTree(left=l, right=r, root=0, lib=self.lib)

    def close(self):
        if getattr(self, "h", None):
            self.lib.lvbhost_refrng_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def refsearch_defaults(lib=None) -> RefSearchParams:
    p = RefSearchParams()
    (lib or load_library()).lvbhost_refsearch_defaults(C.byref(p))
    return p


def reference_search(ctx_handle, params: RefSearchParams, lib=None) -> tuple[dict, "HostTree"]:
    """The reference's own trajectory on the device scorer -> (result dict, tree holding the
    final tree and, as best_trees(), the treestack in the reference's order)."""
    lib = lib or load_library()
    res = RefSearchResult()
    h = C.c_void_p()
    rc = lib.lvbhost_reference_search(ctx_handle, C.byref(params), C.byref(res), C.byref(h))
    if rc != 0:
        raise api.LvbGpuError(rc, "reference_search")
    return {k: getattr(res, k) for k, _ in RefSearchResult._fields_}, HostTree(handle=h, lib=lib)


def anneal_defaults(lib=None) -> AnnealParams:
    p = AnnealParams()
    (lib or load_library()).lvbhost_anneal_defaults(C.byref(p))
    return p


def anneal(ctx: api.FitchContext, tree: HostTree, params: AnnealParams):
    """Run the batched SA loop; -> (result dict, [(seconds, best_length), ...])."""
    lib = load_library()
    cap = max(int(params.log_cap), 0)
    secs = np.zeros(max(cap, 1), dtype=np.float64)
    best = np.zeros(max(cap, 1), dtype=np.int64)
    res = AnnealResult()
    ctx._chk(lib.lvbhost_anneal(ctx.h, tree.h, C.byref(params), C.byref(res), secs.ctypes.data, best.ctypes.data))
    out = {k: getattr(res, k) for k, _ in AnnealResult._fields_}
    log = [(float(secs[i]), int(best[i])) for i in range(res.n_log)]
    return out, log


def anneal_chains(ctx, trees: list[HostTree], params: list[AnnealParams], lib=None):
    """R chains stepped together on one GPU -> ([result dict per chain], [(seconds, best over all chains), ...]).
    ctx: an api.FitchContext (or, with `lib`, the raw handle of whatever that library's lvbgpu_* calls expect)."""
    lib = lib or load_library()
    R = len(trees)
    cap = max(int(params[0].log_cap), 0)
    secs = np.zeros(max(cap, 1), dtype=np.float64)
    best = np.zeros(max(cap, 1), dtype=np.int64)
    handles = (C.c_void_p * R)(*[t.h for t in trees])
    pars = (AnnealParams * R)(*params)
    res = (AnnealResult * R)()
    nlog = C.c_int32()
    rc = lib.lvbhost_anneal_chains(getattr(ctx, "h", ctx), R, handles, pars, res, secs.ctypes.data, best.ctypes.data,
                                   C.byref(nlog))
    if hasattr(ctx, "_chk"):
        ctx._chk(rc)
    elif rc != 0:
        raise api.LvbGpuError(rc, "lvbhost_anneal_chains")
    out = [{k: getattr(res[c], k) for k, _ in AnnealResult._fields_} for c in range(R)]
    return out, [(float(secs[i]), int(best[i])) for i in range(nlog.value)]


def starting_temperature(ctx: api.FitchContext, tree: HostTree, params: AnnealParams) -> float:
    t0 = C.c_double()
    ctx._chk(load_library().lvbhost_starting_temperature(ctx.h, tree.h, C.byref(params), C.byref(t0)))
    return t0.value


def prepare_alignment(rows: list[bytes], lib=None) -> tuple[list[bytes], int]:
    """matchange (reference DataOperations.c:309-360): drop constant columns, then
    MinimumTreeLength of what is left.  -> (rows after the cut, min_len_tree)."""
    lib = lib or load_library()
    n, m = len(rows), len(rows[0])
    arr = (C.c_char_p * n)(*rows)
    keep = np.zeros(m, dtype=np.uint8)
    kept = lib.lvbhost_variable_columns(n, m, arr, keep.ctypes.data)
    if kept < 0:
        raise api.LvbGpuError(int(kept), "variable_columns")
    if kept != m:
        idx = np.nonzero(keep)[0]
        rows = [np.frombuffer(r, dtype=np.uint8)[idx].tobytes() for r in rows]
        arr = (C.c_char_p * n)(*rows)
    if kept < 1:
        # the reference's wording (DataOperations.c:346-350, MIN_M = 1)
        raise ValueError(f"after constant columns are ignored, data MSA has\n{int(kept)} columns, which is less than "
                         "LVB's lower limit of\n1 columns.\n")
    return rows, int(lib.lvbhost_min_tree_length(n, int(kept), arr))


FORMATS = {"phylip": 0, "fasta": 1, "nexus": 2, "clustal": 3}  # the reference's -f (DataStructure.h:50-53)


def read_alignment(path, fmt: str | int = "phylip", lib=None) -> tuple[list[bytes], list[bytes]]:
    """Alignment file -> (names, rows): what the reference's reader hands to matchange."""
    import os
    lib = lib or load_library()
    code = FORMATS[fmt] if isinstance(fmt, str) else int(fmt)
    err = C.create_string_buffer(4096)
    h = lib.lvbhost_alignment_read(os.fsencode(path), code, err, 4096)
    if not h:
        raise ValueError(err.value.decode(errors="replace"))
    try:
        n = lib.lvbhost_alignment_n(h)
        return ([lib.lvbhost_alignment_name(h, i) for i in range(n)], [lib.lvbhost_alignment_row(h, i) for i in range(n)])
    finally:
        lib.lvbhost_alignment_free(h)


def read_phylip(path) -> tuple[list[bytes], list[bytes]]:
    return read_alignment(path, "phylip")


def newick(tree: "HostTree", names: list[bytes]) -> str:
    """One line, the reference's unrooted bracket form (ur_print)."""
    lib = tree.lib
    arr = (C.c_char_p * len(names))(*names)
    cap = sum(len(x) + 4 for x in names) * 2 + 64
    buf = C.create_string_buffer(cap)
    k = lib.lvbhost_tree_newick(tree.h, arr, buf, cap)
    if k < 0:
        raise api.LvbGpuError(int(k), "newick")
    return buf.value.decode()
