"""Python face of the C ABI in include/lvbgpu.h (ctypes; no torch types cross the boundary).

Mirrors the reference's call surface for the path: ``FitchContext.getplen(...)`` family instead of
``getplen(Dataptr, TREESTACK_TREE_NODES*, ...)`` (reference LVB.h:187).  Everything here calls
into lvb_amd/liblvbgpu.so; if that library (or a HIP device) is missing the call raises
``LvbGpuError`` - there is no CPU path in this package.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "liblvbgpu.so"

UNSET = -1


class LvbGpuError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"lvbgpu error {status}: {message}")
        self.status = status


class Edit(C.Structure):
    _fields_ = [("node", C.c_int32), ("left", C.c_int32), ("right", C.c_int32)]


class BatchStats(C.Structure):
    _fields_ = [("candidates", C.c_int64), ("combines", C.c_int64), ("rows_read", C.c_int64),
                ("dirty_nodes", C.c_int64), ("max_stack", C.c_int64), ("algorithmic_bytes", C.c_int64)]


EDIT_DTYPE = np.dtype([("node", np.int32), ("left", np.int32), ("right", np.int32)])
MOVE_DTYPE = np.dtype([("kind", np.int32), ("a", np.int32), ("b", np.int32), ("c", np.int32)])  # lvbgpu_move
DRAW_DTYPE = np.dtype([("chain", np.int32), ("count", np.int32), ("kind", np.int32), ("mix_a", np.uint32), ("mix_b", np.uint32),
                       ("_pad", np.uint32), ("seed", np.uint64)])  # lvbgpu_chain_draw (seed 8-byte aligned)
PICK_DTYPE = np.dtype([("chain", np.int32), ("b", np.int32)])  # lvbgpu_chain_pick
RULE_DTYPE = np.dtype([("cur_length", np.int64), ("temperature", np.float64), ("min_len_tree", np.float64),
                       ("accept_seed", np.uint64)])  # lvbgpu_chain_rule

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")

# every symbol include/lvbgpu.h declares: (restype, argtypes)
SIGNATURES = {
    "lvbgpu_strerror": (C.c_char_p, [C.c_int]),
    "lvbgpu_last_error": (C.c_char_p, [C.c_void_p]),
    "lvbgpu_device_count": (C.c_int, []),
    "lvbgpu_abi_version": (C.c_int, []),
    "lvbgpu_words_per_row": (C.c_long, [C.c_long]),
    "lvbgpu_encode_text": (C.c_int, [C.c_int, C.c_long, C.c_long, C.POINTER(C.c_char_p), _u64p]),
    "lvbgpu_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_long, C.c_long, _u64p, C.c_long]),
    "lvbgpu_create_from_text": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_long, C.c_long, C.POINTER(C.c_char_p)]),
    "lvbgpu_destroy": (None, [C.c_void_p]),
    "lvbgpu_n": (C.c_long, [C.c_void_p]),
    "lvbgpu_nwords": (C.c_long, [C.c_void_p]),
    "lvbgpu_set_chains": (C.c_int, [C.c_void_p, C.c_int32]),
    "lvbgpu_select_chain": (C.c_int, [C.c_void_p, C.c_int32]),
    "lvbgpu_chains": (C.c_int32, [C.c_void_p]),
    "lvbgpu_set_tree": (C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int32, C.POINTER(C.c_int64)]),
    "lvbgpu_current_length": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "lvbgpu_get_topology": (C.c_int, [C.c_void_p, _i32p, _i32p, _i32p, C.POINTER(C.c_int32)]),
    "lvbgpu_get_changes": (C.c_int, [C.c_void_p, _i64p]),
    "lvbgpu_get_sets": (C.c_int, [C.c_void_p, C.c_int32, _u64p]),
    "lvbgpu_score_batch": (C.c_int, [C.c_void_p, C.c_int32, _i32p, C.c_void_p, C.c_void_p, _i64p]),
    "lvbgpu_batch_build": (C.c_int, [C.c_void_p, C.c_int32, _i32p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "lvbgpu_batch_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lvbgpu_batch_lengths": (C.c_int, [C.c_void_p, C.c_void_p, _i64p]),
    "lvbgpu_batch_get_stats": (C.c_int, [C.c_void_p, C.POINTER(BatchStats)]),
    "lvbgpu_batch_free": (None, [C.c_void_p]),
    "lvbgpu_propose_score": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_uint64, _i64p]),
    "lvbgpu_score_moves": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, _i64p]),
    "lvbgpu_propose_score_mixed": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_int64, C.c_uint64,
                                            _i64p]),
    "lvbgpu_proposal_edits": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), _i32p]),
    "lvbgpu_chains_propose_score": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, _i64p]),
    "lvbgpu_chains_submit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "lvbgpu_chains_collect": (C.c_int, [C.c_void_p, C.c_int32, _i64p]),
    "lvbgpu_chains_commit": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "lvbgpu_chains_reroot": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "lvbgpu_parallel_for": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "lvbgpu_chains_score_edits": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p, C.c_void_p, _i64p]),
    "lvbgpu_chains_commit_edits": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p, C.c_void_p]),
    "lvbgpu_chains_picked_edits": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lvbgpu_chains_step_submit": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "lvbgpu_chains_step_collect": (C.c_int, [C.c_void_p, C.c_int32, _i64p, _i32p]),
    "lvbgpu_chains_ready": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lvbgpu_fork": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "lvbgpu_chains_step_edits": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lvbgpu_proposal_stats": (C.c_int, [C.c_void_p, C.POINTER(BatchStats)]),
    "lvbgpu_score_full_batch": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p, C.c_void_p, _i64p]),
    "lvbgpu_commit": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "lvbgpu_getplen_compat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.POINTER(C.c_int64)]),
    "lvbgpu_timer_start": (C.c_int, [C.c_void_p]),
    "lvbgpu_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "lvbgpu_synchronize": (C.c_int, [C.c_void_p]),
    "lvbgpu_walk_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "lvbgpu_walk_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "lvbgpu_probe_l2": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "lvbgpu_stream": (C.c_void_p, [C.c_void_p]),
    "lvbgpu_set_wait_limit": (C.c_int, [C.c_void_p, C.c_double]),
    "lvbgpu_debug_stall": (C.c_int, [C.c_void_p, C.c_int32]),
    "lvbgpu_debug_post_stamps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lvbgpu_debug_pairs": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "lvbgpu_debug_count": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]),
    "lvbgpu_comm_available": (C.c_int, []),
    "lvbgpu_comm_unique_id": (C.c_int, [C.c_void_p]),
    "lvbgpu_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "lvbgpu_comm_size": (C.c_int, [C.c_void_p]),
    "lvbgpu_allreduce_min": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "lvbgpu_allreduce_sum": (C.c_int, [C.c_void_p, _i64p, C.c_int32]),
    "lvbgpu_comm_destroy": (C.c_int, [C.c_void_p]),
}

_lib = None


def load_library() -> C.CDLL:
    """Load liblvbgpu.so or raise: the HIP library is the product, nothing stands in for it."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise LvbGpuError(-2, f"{LIB_PATH} is not built (run `python -m lvb_amd.build`); "
                              "lvb_amd has no CPU fallback")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def device_count() -> int:
    lib = load_library()
    c = lib.lvbgpu_device_count()
    if c < 0:
        raise LvbGpuError(c, (lib.lvbgpu_last_error(None) or b"").decode())
    return c


def words_per_row(m: int) -> int:
    return int(load_library().lvbgpu_words_per_row(m))


def _edits_array(edits) -> np.ndarray:
    arr = np.ascontiguousarray(edits)
    if arr.dtype != EDIT_DTYPE:
        arr = np.ascontiguousarray(np.asarray(edits, dtype=np.int32).reshape(-1, 3)).view(EDIT_DTYPE).reshape(-1)
    return arr


def encode_text(rows: list[bytes], device: int = 0) -> np.ndarray:
    """DNAToBinary on the device (reference DataOperations.c:164-249)."""
    lib = load_library()
    n, m = len(rows), len(rows[0])
    out = np.zeros((n, lib.lvbgpu_words_per_row(m)), dtype=np.uint64)
    arr = (C.c_char_p * n)(*rows)
    rc = lib.lvbgpu_encode_text(device, n, m, arr, out)
    if rc != 0:
        raise LvbGpuError(rc, (lib.lvbgpu_last_error(None) or b"").decode())
    return out


class Batch:
    """A resident batch of candidate programs (lvbgpu_batch)."""

    def __init__(self, ctx: "FitchContext", handle: C.c_void_p, B: int):
        self.ctx, self.h, self.B = ctx, handle, B

    def launch(self) -> None:
        self.ctx._chk(self.ctx.lib.lvbgpu_batch_launch(self.ctx.h, self.h))

    def lengths(self) -> np.ndarray:
        out = np.zeros(self.B, dtype=np.int64)
        self.ctx._chk(self.ctx.lib.lvbgpu_batch_lengths(self.ctx.h, self.h, out))
        return out

    def stats(self) -> dict:
        st = BatchStats()
        self.ctx._chk(self.ctx.lib.lvbgpu_batch_get_stats(self.h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in BatchStats._fields_}

    def free(self) -> None:
        if self.h:
            self.ctx.lib.lvbgpu_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FitchContext:
    """One alignment resident on one MI355X, plus (optionally) a resident current tree."""

    def __init__(self, leaf_matrix: np.ndarray | None = None, *, text_rows: list[bytes] | None = None,
                 device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        if leaf_matrix is not None:
            m = np.ascontiguousarray(leaf_matrix, dtype=np.uint64)
            n, nwords = m.shape
            rc = self.lib.lvbgpu_create(C.byref(h), device, n, nwords, m, nwords)
        elif text_rows is not None:
            n, mm = len(text_rows), len(text_rows[0])
            arr = (C.c_char_p * n)(*text_rows)
            rc = self.lib.lvbgpu_create_from_text(C.byref(h), device, n, mm, arr)
        else:
            raise ValueError("leaf_matrix or text_rows required")
        if rc != 0:
            raise LvbGpuError(rc, (self.lib.lvbgpu_last_error(None) or b"").decode()
                              or self.lib.lvbgpu_strerror(rc).decode())
        self.h = h
        self.n = int(self.lib.lvbgpu_n(h))
        self.nwords = int(self.lib.lvbgpu_nwords(h))
        self.nbranches = 2 * self.n - 3

    def fork(self) -> "FitchContext":
        """A second context on the same alignment (leaf rows copied on the device): own stream, one tree slot, no tree."""
        h = C.c_void_p()
        self._chk(self.lib.lvbgpu_fork(self.h, C.byref(h)))
        twin = object.__new__(FitchContext)
        twin.lib, twin.h, twin.n, twin.nwords, twin.nbranches = self.lib, h, self.n, self.nwords, self.nbranches
        return twin

    def last_error(self) -> str:
        return (self.lib.lvbgpu_last_error(self.h) or b"").decode()

    def debug_count(self, what: int) -> int:
        out = C.c_int64(0)
        self._chk(self.lib.lvbgpu_debug_count(self.h, int(what), C.byref(out)))
        return int(out.value)

    def paired_walks(self) -> int:
        """Scoring walks launched two candidates per wave so far (LVBGPU_PAIR)."""
        return self.debug_count(0)

    def commits_reusing_programs(self) -> int:
        """chains_commit_edits calls that walked the scored programs of the last chains_score_edits call."""
        return self.debug_count(1)

    def post_launches(self) -> tuple[int, int]:
        """(post launches so far - the chains' commits and re-roots, all pending ones in one launch -, those of them that
        carried the next batch's generator)."""
        return self.debug_count(2), self.debug_count(3)

    def last_pairs(self, slot: int = 0) -> np.ndarray:
        """who walked with whom in the last device-built batch of `slot`: [npairs, 2] candidate indices (0xFFFFFFFF: alone);
        empty when that batch was walked one candidate per wave"""
        cap = 1 << 16
        out = np.zeros((cap, 2), dtype=np.uint32)
        n = C.c_int32(0)
        self._chk(self.lib.lvbgpu_debug_pairs(self.h, int(slot), out.ctypes.data, cap, C.byref(n)))
        return out[: n.value].copy()

    def set_wait_limit(self, seconds: float) -> None:
        self._chk(self.lib.lvbgpu_set_wait_limit(self.h, float(seconds)))

    def _chk(self, rc: int) -> None:
        if rc != 0:
            msg = (self.lib.lvbgpu_last_error(self.h) or b"").decode() or self.lib.lvbgpu_strerror(rc).decode()
            raise LvbGpuError(rc, msg)

    def close(self) -> None:
        if getattr(self, "h", None):
            self.lib.lvbgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- several resident trees (chains)
    def set_chains(self, nchains: int) -> None:
        self._chk(self.lib.lvbgpu_set_chains(self.h, int(nchains)))

    def select_chain(self, chain: int) -> None:
        self._chk(self.lib.lvbgpu_select_chain(self.h, int(chain)))

    @staticmethod
    def _draws(draws) -> np.ndarray:
        d = np.zeros(len(draws), dtype=DRAW_DTYPE)
        for i, row in enumerate(draws):
            d[i]["chain"], d[i]["count"], d[i]["kind"], d[i]["seed"] = row[0], row[1], row[2], row[3]
            if len(row) > 4:
                d[i]["mix_a"], d[i]["mix_b"] = row[4], row[5]
        return d

    def chains_propose_score(self, draws) -> list[np.ndarray]:
        """draws: (chain, count, kind, seed[, mix_a, mix_b]) per chain -> the lengths, one array per draw."""
        d = self._draws(draws)
        out = np.zeros(int(d["count"].sum()), dtype=np.int64)
        self._chk(self.lib.lvbgpu_chains_propose_score(self.h, len(d), d.ctypes.data, out))
        return np.split(out, np.cumsum(d["count"])[:-1])

    def chains_submit(self, slot: int, draws) -> np.ndarray:
        """Asynchronous half: enqueue the batch in `slot` (0 or 1); -> the counts, for chains_collect."""
        d = self._draws(draws)
        self._chk(self.lib.lvbgpu_chains_submit(self.h, int(slot), len(d), d.ctypes.data))
        return d["count"].copy()

    def chains_collect(self, slot: int, counts) -> list[np.ndarray]:
        out = np.zeros(int(np.sum(counts)), dtype=np.int64)
        self._chk(self.lib.lvbgpu_chains_collect(self.h, int(slot), out))
        return np.split(out, np.cumsum(counts)[:-1])

    def chains_step(self, draws, rules, slot: int = 0):
        """One annealing step for the listed chains: draws as chains_propose_score, rules = (cur_length, temperature,
        min_len_tree, accept_seed) per draw -> (lengths per draw, picks per draw [-1: nothing accepted]); the accepted
        moves are committed."""
        d = self._draws(draws)
        r = np.zeros(len(d), dtype=RULE_DTYPE)
        for i, row in enumerate(rules):
            r[i]["cur_length"], r[i]["temperature"], r[i]["min_len_tree"], r[i]["accept_seed"] = row
        self._chk(self.lib.lvbgpu_chains_step_submit(self.h, int(slot), len(d), d.ctypes.data, r.ctypes.data))
        out = np.zeros(int(d["count"].sum()), dtype=np.int64)
        picks = np.zeros(len(d), dtype=np.int32)
        self._chk(self.lib.lvbgpu_chains_step_collect(self.h, int(slot), out, picks))
        return np.split(out, np.cumsum(d["count"])[:-1]), picks

    def chains_step_edits(self, i: int) -> np.ndarray:
        cap = 2 * self.nbranches + 8
        buf = np.zeros(cap, dtype=EDIT_DTYPE)
        ne = C.c_int32()
        self._chk(self.lib.lvbgpu_chains_step_edits(self.h, int(i), buf.ctypes.data, cap, C.byref(ne)))
        return buf[: ne.value].copy()

    def chains_reroot(self, reqs) -> None:
        """reqs: (chain, new_root_leaf) - re-root those chains in one commit walk."""
        p = np.array([tuple(int(v) for v in row) for row in reqs], dtype=PICK_DTYPE)
        self._chk(self.lib.lvbgpu_chains_reroot(self.h, len(p), p.ctypes.data))

    def chains_score_edits(self, chains, cands) -> np.ndarray:
        """Host-made candidates of several chains in one walk: cands[b] = rewrites of chain chains[b]'s resident tree."""
        off, ed, _ = self._pack(cands, None)
        out = np.zeros(len(cands), dtype=np.int64)
        ch = np.ascontiguousarray(chains, dtype=np.int32)
        self._chk(self.lib.lvbgpu_chains_score_edits(self.h, len(cands), ch, off, ed.ctypes.data, out))
        return out

    def chains_commit_edits(self, chains, cands) -> None:
        """Accept one host-made candidate per listed chain in one commit walk (asynchronous)."""
        off, ed, _ = self._pack(cands, None)
        ch = np.ascontiguousarray(chains, dtype=np.int32)
        self._chk(self.lib.lvbgpu_chains_commit_edits(self.h, len(cands), ch, off, ed.ctypes.data))

    def chains_commit(self, picks) -> None:
        """picks: (chain, b) - candidate b of that chain's draw in the last chains_propose_score call."""
        p = np.array([tuple(int(v) for v in row) for row in picks], dtype=PICK_DTYPE)
        self._chk(self.lib.lvbgpu_chains_commit(self.h, len(p), p.ctypes.data))

    # ---- resident tree
    def set_tree(self, left, right, root: int) -> int:
        l = np.ascontiguousarray(left, dtype=np.int32)
        r = np.ascontiguousarray(right, dtype=np.int32)
        out = C.c_int64()
        self._chk(self.lib.lvbgpu_set_tree(self.h, l, r, int(root), C.byref(out)))
        return out.value

    def current_length(self) -> int:
        out = C.c_int64()
        self._chk(self.lib.lvbgpu_current_length(self.h, C.byref(out)))
        return out.value

    def topology(self):
        nb = self.nbranches
        p, l, r = (np.zeros(nb, dtype=np.int32) for _ in range(3))
        root = C.c_int32()
        self._chk(self.lib.lvbgpu_get_topology(self.h, p, l, r, C.byref(root)))
        return p, l, r, root.value

    def changes(self) -> np.ndarray:
        out = np.zeros(self.nbranches, dtype=np.int64)
        self._chk(self.lib.lvbgpu_get_changes(self.h, out))
        return out

    def sets(self, node: int) -> np.ndarray:
        out = np.zeros(self.nwords, dtype=np.uint64)
        self._chk(self.lib.lvbgpu_get_sets(self.h, int(node), out))
        return out

    def all_sets(self) -> np.ndarray:
        return np.stack([self.sets(i) for i in range(self.nbranches)])

    # ---- batches of candidates (edits relative to the resident tree)
    @staticmethod
    def _pack(cands, roots):
        offs = np.zeros(len(cands) + 1, dtype=np.int32)
        flat = []
        for i, e in enumerate(cands):
            e = _edits_array(e)
            flat.append(e)
            offs[i + 1] = offs[i] + len(e)
        edits = np.concatenate(flat) if flat and offs[-1] > 0 else np.zeros(0, dtype=EDIT_DTYPE)
        rts = None if roots is None else np.ascontiguousarray(roots, dtype=np.int32)
        return offs, np.ascontiguousarray(edits), rts

    def score_batch(self, cands, roots=None) -> np.ndarray:
        offs, edits, rts = self._pack(cands, roots)
        out = np.zeros(len(cands), dtype=np.int64)
        self._chk(self.lib.lvbgpu_score_batch(self.h, len(cands), offs, edits.ctypes.data,
                                               None if rts is None else rts.ctypes.data, out))
        return out

    def build_batch(self, cands, roots=None) -> Batch:
        offs, edits, rts = self._pack(cands, roots)
        bh = C.c_void_p()
        self._chk(self.lib.lvbgpu_batch_build(self.h, len(cands), offs, edits.ctypes.data,
                                               None if rts is None else rts.ctypes.data, C.byref(bh)))
        return Batch(self, bh, len(cands))

    # ---- neighbourhoods drawn and scored on the device
    def propose_score(self, B: int, kind: int, seed: int) -> np.ndarray:
        out = np.zeros(B, dtype=np.int64)
        self._chk(self.lib.lvbgpu_propose_score(self.h, B, kind, seed, out))
        return out

    def propose_score_mixed(self, B: int, p_nni: float, p_spr: float, parity: int, seed: int) -> np.ndarray:
        out = np.zeros(B, dtype=np.int64)
        self._chk(self.lib.lvbgpu_propose_score_mixed(self.h, B, p_nni, p_spr, parity, seed, out))
        return out

    def score_moves(self, moves) -> np.ndarray:
        """Moves named by the caller (array of MOVE_DTYPE or rows of (kind, a, b, c)) -> lengths."""
        m = np.ascontiguousarray(moves if getattr(moves, "dtype", None) == MOVE_DTYPE
                                 else np.array([tuple(int(v) for v in row) for row in moves], dtype=MOVE_DTYPE))
        out = np.zeros(len(m), dtype=np.int64)
        self._chk(self.lib.lvbgpu_score_moves(self.h, len(m), m.ctypes.data, out))
        return out

    def proposal_edits(self, b: int):
        """Candidate b of the last propose_score batch -> (edits, [kind, a, b, c])."""
        cap = 2 * self.nbranches + 8
        buf = np.zeros(cap, dtype=EDIT_DTYPE)
        k = C.c_int32()
        info = np.zeros(4, dtype=np.int32)
        self._chk(self.lib.lvbgpu_proposal_edits(self.h, int(b), buf.ctypes.data, cap, C.byref(k), info))
        return buf[: k.value].copy(), info

    def proposal_stats(self) -> dict:
        """Counts of the last device-built batch (candidates, combines, rows_read, dirty_nodes, ...)."""
        st = BatchStats()
        self._chk(self.lib.lvbgpu_proposal_stats(self.h, C.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in BatchStats._fields_}

    def score_full_batch(self, lefts, rights, roots=None) -> np.ndarray:
        l = np.ascontiguousarray(lefts, dtype=np.int32)
        r = np.ascontiguousarray(rights, dtype=np.int32)
        B = l.shape[0]
        rts = None if roots is None else np.ascontiguousarray(roots, dtype=np.int32)
        out = np.zeros(B, dtype=np.int64)
        self._chk(self.lib.lvbgpu_score_full_batch(self.h, B, l.reshape(-1), r.reshape(-1),
                                                    None if rts is None else rts.ctypes.data, out))
        return out

    def commit(self, edits, root: int = -1) -> int:
        e = _edits_array(edits)
        out = C.c_int64()
        self._chk(self.lib.lvbgpu_commit(self.h, len(e), e.ctypes.data, int(root), C.byref(out)))
        return out.value

    # ---- strict compat on a reference-layout tree block
    def getplen_compat(self, tree_block, root: int) -> int:
        out = C.c_int64()
        ptr = tree_block if isinstance(tree_block, int) else C.cast(tree_block, C.c_void_p)
        self._chk(self.lib.lvbgpu_getplen_compat(self.h, ptr, int(root), C.byref(out)))
        return out.value

    # ---- timing / sync
    def timer_start(self) -> None:
        self._chk(self.lib.lvbgpu_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._chk(self.lib.lvbgpu_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def synchronize(self) -> None:
        self._chk(self.lib.lvbgpu_synchronize(self.h))

    def walk_timing(self, every: int | bool) -> None:
        """Time every `every`-th scoring walk with HIP events (True = all, 0/False = off)."""
        self._chk(self.lib.lvbgpu_walk_timing(self.h, int(every)))

    def walk_timing_read(self) -> tuple[float, int]:
        """(sum of the scoring walks' durations in ms, their number) since walk_timing(True)."""
        ms, k = C.c_double(), C.c_int64()
        self._chk(self.lib.lvbgpu_walk_timing_read(self.h, C.byref(ms), C.byref(k)))
        return ms.value, k.value

    def probe_l2(self, B: int, rows_per_wave: int = 24, reps: int = 20) -> float:
        """GB/s a pure-load kernel with the walk's geometry and access pattern reads on this device now."""
        out = C.c_double()
        self._chk(self.lib.lvbgpu_probe_l2(self.h, B, rows_per_wave, reps, C.byref(out)))
        return out.value

    # ---- multi-GPU
    def comm_init(self, nranks: int, rank: int, unique_id: bytes) -> None:
        buf = C.create_string_buffer(unique_id, 128)
        self._chk(self.lib.lvbgpu_comm_init(self.h, nranks, rank, buf))

    def comm_size(self) -> int:
        return int(self.lib.lvbgpu_comm_size(self.h))

    def allreduce_sum(self, values) -> np.ndarray:
        """Site-axis sharding: per-candidate partial lengths in, their sums over the ranks out."""
        v = np.ascontiguousarray(values, dtype=np.int64).copy()
        self._chk(self.lib.lvbgpu_allreduce_sum(self.h, v, len(v)))
        return v

    def allreduce_min(self, value: int) -> tuple[int, int]:
        v = C.c_int64(value)
        who = C.c_int32()
        self._chk(self.lib.lvbgpu_allreduce_min(self.h, C.byref(v), C.byref(who)))
        return v.value, who.value


def site_slice(m: int, rank: int, world: int, tile_sites: int = 2048) -> tuple[int, int]:
    """Columns [lo, hi) of an m-site alignment that rank `rank` of `world` scores when the SITE axis is sharded
    (lvbgpu_allreduce_sum): whole tiles of 2048 sites, dealt as evenly as they go."""
    tiles = (m + tile_sites - 1) // tile_sites
    lo = tiles * rank // world * tile_sites
    hi = tiles * (rank + 1) // world * tile_sites
    return min(lo, m), min(hi, m)


def comm_available() -> bool:
    """Can this process load RCCL?  Agree on it across ranks before the collective comm_init."""
    return load_library().lvbgpu_comm_available() == 0


def comm_unique_id() -> bytes:
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.lvbgpu_comm_unique_id(buf)
    if rc != 0:
        raise LvbGpuError(rc, (lib.lvbgpu_last_error(None) or b"").decode())
    return buf.raw


def edits_between(cur_left, cur_right, new_left, new_right) -> np.ndarray:
    """The child-pair rewrites that turn one topology into another (same node numbering)."""
    cl, cr = np.asarray(cur_left), np.asarray(cur_right)
    nl, nr = np.asarray(new_left), np.asarray(new_right)
    # a node whose two children merely swapped sides is unchanged (Fitch is symmetric)
    changed = np.nonzero((np.minimum(cl, cr) != np.minimum(nl, nr)) | (np.maximum(cl, cr) != np.maximum(nl, nr)))[0]
    out = np.zeros(len(changed), dtype=EDIT_DTYPE)
    out["node"] = changed
    out["left"] = nl[changed]
    out["right"] = nr[changed]
    return out
