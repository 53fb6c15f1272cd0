"""Build the native libraries of lvb_amd in-tree (hipcc / g++ directly, no build system).

  lvb_amd/liblvbgpu.so         HIP kernels + C ABI of include/lvbgpu.h     (hipcc, gfx950)
  lvb_amd/liblvbgpu_compat.so  reference-signature getplen adapter (C++)   (g++, links liblvbgpu)
  lvb_amd/liblvbhost.so        host-side search mirror: proposals, SA loop (g++, links liblvbgpu)

``python -m lvb_amd.build`` builds everything whose sources are newer than the library.
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
INCLUDE = PKG.parent / "include"
ARCH = "gfx950"

GPU_SOURCES = ["fitch_kernels.hip", "propose_kernels.hip", "api_core.cpp", "api_batch.cpp", "api_propose.cpp",
               "api_compat.cpp", "api_comm.cpp", "program.cpp"]
GPU_HEADERS = ["kernels.hpp", "program.hpp", "pool.hpp", "ctx.hpp"]
COMPAT_SOURCES = ["getplen_adapter.cpp"]
HOST_SOURCES = ["host_api.cpp", "proposals.cpp", "anneal.cpp", "anneal_chains.cpp", "refsearch.cpp", "program.cpp"]
HOST_HEADERS = ["program.hpp", "proposals.hpp", "host_tree.hpp", "refrng.hpp"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the HIP path cannot be built (there is no CPU fallback)")


def _stale(target: Path, sources: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(s.exists() and s.stat().st_mtime > t for s in sources)


def _run(cmd: list[str]) -> None:
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build_gpu(force: bool = False) -> Path:
    out = PKG / "liblvbgpu.so"
    srcs = [CSRC / s for s in GPU_SOURCES]
    deps = srcs + [CSRC / h for h in GPU_HEADERS] + [INCLUDE / "lvbgpu.h"]
    if force or _stale(out, deps):
        _run([_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-result", "-x", "hip", *map(str, srcs), "-o", str(out), "-ldl", "-pthread"])
    return out


def build_compat(force: bool = False) -> Path | None:
    out = PKG / "liblvbgpu_compat.so"
    srcs = [CSRC / s for s in COMPAT_SOURCES]
    if not all(s.exists() for s in srcs):
        return None
    if force or _stale(out, srcs + [INCLUDE / "lvbgpu.h", PKG / "liblvbgpu.so"]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", f"-I{INCLUDE}", *map(str, srcs), "-o", str(out),
              f"-L{PKG}", "-llvbgpu", "-Wl,-rpath,$ORIGIN"])
    return out


def build_host(force: bool = False) -> Path | None:
    out = PKG / "liblvbhost.so"
    srcs = [CSRC / s for s in HOST_SOURCES]
    if not all(s.exists() for s in srcs):
        return None
    deps = srcs + [CSRC / h for h in HOST_HEADERS] + [INCLUDE / "lvbgpu.h", INCLUDE / "lvbhost.h"]
    if force or _stale(out, deps + [PKG / "liblvbgpu.so"]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-pthread", f"-I{INCLUDE}", *map(str, srcs), "-o",
              str(out), f"-L{PKG}", "-llvbgpu", "-Wl,-rpath,$ORIGIN"])
    return out


def build_all(force: bool = False) -> dict[str, Path | None]:
    return {"gpu": build_gpu(force), "compat": build_compat(force), "host": build_host(force)}


if __name__ == "__main__":
    libs = build_all(force="--force" in sys.argv)
    for k, v in libs.items():
        print(f"{k}: {v}")
