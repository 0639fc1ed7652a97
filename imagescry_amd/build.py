"""Compile the HIP sources under `imagescry_amd/csrc/` into `imagescry_amd/libimagescry_hip.so` for gfx950.

Run as `python -m imagescry_amd.build`.  hipcc cross-compiles without a GPU.  The library is kept in-tree
(git-ignored) so that it travels with the working copy.
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "build"
LIB = PKG / "libimagescry_hip.so"
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
# `--ablation` builds a SECOND library, libimagescry_hip_ablation.so, with -DISC_ABLATION: the kernels' wrong-result
# bring-up variants behind ISC_DEBUG_MODE / ISC_FORCE_TILE / ISC_GEMM_DEBUG.  Select it with ISC_LIB=<path> (scripts/
# only); the production library contains none of them (tests/test_capi_symbols.py).
ABLATION_LIB = PKG / "libimagescry_hip_ablation.so"


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found (expected on PATH or at /opt/rocm/bin/hipcc)")
    return exe


def _stale(target: Path, deps: list[Path]) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True, ablation: bool = False, variant: str = "",
          defines: tuple[str, ...] = ()) -> Path:
    """`variant` + `defines`: an experiment build `libimagescry_hip_<variant>.so` with extra -D flags (scripts/ only,
    always an ablation build: the binding refuses it without ISC_ALLOW_ABLATION=1)."""
    hipcc = _hipcc()
    if variant:
        ablation = True
    obj_dir = CSRC / (f"build_{variant}" if variant else "build_ablation" if ablation else "build")
    lib = PKG / f"libimagescry_hip_{variant}.so" if variant else ABLATION_LIB if ablation else LIB
    flags = [*FLAGS, "-DISC_ABLATION", *(f"-D{d}" for d in defines)] if ablation else FLAGS
    obj_dir.mkdir(exist_ok=True)
    sources = sorted(CSRC.glob("*.hip"))
    headers = sorted(CSRC.glob("*.h")) + [PKG.parent / "include" / "imagescry_hip.h"]

    def compile_one(src: Path) -> Path:
        obj = obj_dir / (src.stem + ".o")
        if force or _stale(obj, [src, *headers]):
            cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, sources))
    if force or _stale(lib, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", str(lib), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    _variant = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")), "")
    _defines = tuple(a[2:] for a in sys.argv if a.startswith("-D"))
    print(build(force="--force" in sys.argv, ablation="--ablation" in sys.argv, variant=_variant, defines=_defines))
