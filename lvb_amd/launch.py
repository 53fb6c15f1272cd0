"""One process per GPU: the small amount of multi-process plumbing around the scoring library.

torch.distributed is used for rendezvous, barriers and the max-over-ranks of the timing only (backend
"nccl" = RCCL on the GPU box, "gloo" in CPU tests).  The data-path collective - the min-reduce of the
best tree length over independent restarts - is the library's own RCCL call (lvbgpu_allreduce_min);
the communicator id is created on rank 0 and shared through `share_bytes`.
"""
from __future__ import annotations

import os
import threading


def pin_to_share_of_cores() -> dict:
    """One process per GPU: keep this rank on ITS share of the cores the job may run on - the affinity mask cut into
    LOCAL_WORLD_SIZE contiguous pieces - and size the library's thread pool to it (LVBGPU_THREADS, unless set).  Call it
    before the scoring library is loaded (its pool reads the mask then) and before anything starts threads.  Eight ranks
    with a spinning submit / collect loop and sixteen pool threads each, all free to run anywhere, is how a scaling run
    loses on the host what the GPUs deliver.  -> what was done, for the result line."""
    lws = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1))
    lr = int(os.environ.get("LOCAL_RANK", "0") or 0) % lws
    info = {"local_world_size": lws, "pinned": False}
    try:
        cores = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = list(range(os.cpu_count() or 1))
    share = cores[len(cores) * lr // lws: len(cores) * (lr + 1) // lws] or cores
    if lws > 1 and share and len(share) < len(cores) and os.environ.get("LVBGPU_NO_PIN") is None:
        try:
            os.sched_setaffinity(0, share)
            info["pinned"] = True
        except (AttributeError, OSError):
            share = cores
    else:
        share = cores if lws == 1 else share
    threads = max(2, min(16, len(share)))
    os.environ.setdefault("LVBGPU_THREADS", str(threads))
    info.update(cores_allowed=len(cores), cores_per_rank=len(share), threads=int(os.environ["LVBGPU_THREADS"]))
    return info


def call_with_deadline(fn, seconds: float):
    """fn() on a helper thread, waited for at most `seconds` -> (done, result or exception).  For collective set-up calls
    that can hang when one rank never arrives (ncclCommInitRank has no deadline of its own): the caller falls back or gives
    up instead of hanging the whole job.  A call that did not come back keeps its (daemon) thread; the caller should end
    the process with os._exit once its work is reported, never start another program from it."""
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as exc:   # noqa: BLE001 - handed to the caller
            box["error"] = exc

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        return False, None
    if "error" in box:
        return True, box["error"]
    return True, box.get("value")


class Ranks:
    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = None
        self.device = self.local_rank  # the GPU this rank scores on
        # rehearsal of the N-rank flow on a box with ONE GPU: every rank shares device 0 and the
        # plumbing runs over gloo (RCCL refuses two ranks on one device).  Never set on a real node.
        if os.environ.get("LVBGPU_REHEARSE_ON_ONE_GPU"):
            backend = backend or "gloo"
            self.device = 0
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self.backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend)
            self.dist = dist

    def restart_seed(self, base: int) -> int:
        """Every rank is an independent restart: distinct, reproducible seeds."""
        return base * 1000 + self.rank + 1

    def share_bytes(self, payload: bytes | None, src: int = 0) -> bytes:
        if self.dist is None:
            return payload
        box = [payload if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def _tensor(self, value, dtype):
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        return torch.tensor([value], dtype=dtype, device=dev)

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return value
        import torch
        t = self._tensor(value, torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: int) -> int:
        if self.dist is None:
            return value
        import torch
        t = self._tensor(value, torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def all_values(self, value: float) -> list:
        """every rank's value, in rank order, on every rank (one sum of a vector that is zero but at one's own place)"""
        if self.dist is None:
            return [float(value)]
        import torch
        dev = "cuda" if self.backend == "nccl" else "cpu"
        t = torch.zeros(self.world, dtype=torch.float64, device=dev)
        t[self.rank] = float(value)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()
            if self.backend == "nccl":
                import torch
                torch.cuda.synchronize()

    def close(self) -> None:
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def _selftest() -> None:
    """Run under 2+ processes (gloo): prints one line per rank for tests/test_launch_gloo.py."""
    import json
    r = Ranks(backend="gloo")
    token = r.share_bytes(b"id-from-rank-0" if r.rank == 0 else None)
    slow = r.max_over_ranks(1.0 + r.rank)
    total = r.sum_over_ranks(10 + r.rank)
    r.barrier()
    print(json.dumps({"rank": r.rank, "world": r.world, "seed": r.restart_seed(3), "token": token.decode(),
                      "max": slow, "sum": total}), flush=True)
    r.close()


if __name__ == "__main__":
    _selftest()
