"""Race detection for the host side (the reference has none): the thread pool and the
parallel program-build pattern of api_batch.cpp under ThreadSanitizer, on the CPU."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def test_parallel_program_build_is_race_free_and_equals_serial(tmp_path):
    exe = tmp_path / "pool_tsan"
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread",
                            str(ROOT / "tests" / "native" / "pool_tsan.cpp"), str(ROOT / "lvb_amd" / "csrc" / "program.cpp"),
                            str(ROOT / "lvb_amd" / "csrc" / "proposals.cpp"), "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in build.stderr.lower():
        pytest.skip("ThreadSanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env={"TSAN_OPTIONS": "halt_on_error=1 exitcode=66"})
    assert run.returncode == 0 and run.stdout.strip() == "ok", (run.returncode, run.stdout[-500:], run.stderr[-3000:])


def test_runs_of_acceptances_on_threads_are_race_free(tmp_path):
    """lvbhost_anneal_chains with runs of accepted moves (the hot chains' candidates drawn, consumed and followed on the
    context's host threads) under ThreadSanitizer, on the scorer's test double: no race between the chains' tasks, and
    every chain ends where it ends with one move per step."""
    from lvb_amd import build as product_build
    exe = tmp_path / "chains_tsan"
    srcs = [str(ROOT / "lvb_amd" / "csrc" / s) for s in product_build.HOST_SOURCES]
    objs = []
    for c in (ROOT / "tests" / "cpu_double" / "lvbgpu_double.c", ROOT / "oracle" / "fitch_oracle.c"):
        o = tmp_path / (c.stem + ".o")
        r = subprocess.run(["gcc", "-O1", "-g", "-std=gnu11", "-fsanitize=thread", "-c", str(c), "-o", str(o)], capture_output=True, text=True)
        if r.returncode != 0 and "tsan" in r.stderr.lower():
            pytest.skip("ThreadSanitizer runtime not installed")
        assert r.returncode == 0, r.stderr[-2000:]
        objs.append(str(o))
    build = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=thread", f"-I{ROOT / 'include'}",
                            str(ROOT / "tests" / "native" / "chains_tsan.cpp"), *srcs, *objs, "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in build.stderr.lower():
        pytest.skip("ThreadSanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900,
                         env={"TSAN_OPTIONS": "halt_on_error=1 exitcode=66"})
    assert run.returncode == 0 and run.stdout.strip() == "ok", (run.returncode, run.stdout[-500:], run.stderr[-3000:])
