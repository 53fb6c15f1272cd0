"""Build the CPU test double: the host sources (lvb_amd/csrc/*.cpp that make liblvbhost.so) linked
against tests/cpu_double/lvbgpu_double.c + the oracle instead of liblvbgpu.so.  Test tier only."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
OUT = HERE / "_build" / "liblvbhost_double.so"


def build() -> Path:
    from lvb_amd import build as product_build
    srcs = [ROOT / "lvb_amd" / "csrc" / s for s in product_build.HOST_SOURCES]
    csrcs = [HERE / "lvbgpu_double.c", ROOT / "oracle" / "fitch_oracle.c"]
    deps = srcs + csrcs + list((ROOT / "lvb_amd" / "csrc").glob("*.hpp")) + list((ROOT / "include").glob("*.h"))
    if OUT.exists() and all(OUT.stat().st_mtime > d.stat().st_mtime for d in deps):
        return OUT
    OUT.parent.mkdir(exist_ok=True)
    objs = []
    for s in csrcs:
        o = OUT.parent / (s.stem + ".o")
        subprocess.run(["gcc", "-O2", "-fPIC", "-std=gnu11", "-c", str(s), "-o", str(o)], check=True)
        objs.append(str(o))
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", f"-I{ROOT / 'include'}",
                    *map(str, srcs), *objs, "-Wl,-Bsymbolic", "-Wl,--no-undefined", "-o", str(OUT)], check=True)
    return OUT


def load():
    """-> (library with lvbhost.h's signatures bound, new_ctx(text_rows) -> handle, free_ctx)."""
    from lvb_amd import host
    from oracle import binding
    lib = host.bind(C.CDLL(str(build())))
    lib.lvbgpu_double_new.restype = C.c_void_p
    lib.lvbgpu_double_new.argtypes = [C.c_long, C.c_long, np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")]
    lib.lvbgpu_double_free.argtypes = [C.c_void_p]

    def new_ctx(rows):
        enc = binding.encode_rows(rows)
        return C.c_void_p(lib.lvbgpu_double_new(enc.shape[0], enc.shape[1], np.ascontiguousarray(enc)))

    return lib, new_ctx, lib.lvbgpu_double_free
