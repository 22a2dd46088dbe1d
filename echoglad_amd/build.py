"""Build the gfx950 C-ABI library in-tree with hipcc (no torch headers, no cmake).

    python -m echoglad_amd.build [--force] [--verbose]

The shared object lands in echoglad_amd/lib/libechoglad_hip.so so it travels with
the repository snapshot to the GPU box."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIBNAME = "libechoglad_hip.so"
SOURCES = ["graph.hip", "gcn_layer.hip", "gcn_layer_ps.hip", "conn.hip", "classifier.hip", "train.hip", "bn_act_tiles.hip", "cls_train.hip", "coord.hip", "coord_mlp.hip", "heatmap.hip", "pack.hip", "pool.hip", "adam.hip"]
ARCH = "gfx950"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIP extension cannot be built")


def lib_path() -> Path:
    return LIBDIR / LIBNAME


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags=(), variant: str = "") -> Path:
    """variant != "": an experiment build (extra -D flags) written to lib/libechoglad_hip.<variant>.so;
    select it at run time with ECHOGLAD_LIB=<path>."""
    LIBDIR.mkdir(exist_ok=True)
    objdir = LIBDIR / ("obj" + ("_" + variant if variant else ""))
    objdir.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + [PKG.parent / "include" / "echoglad_hip.h"]
    cc = hipcc()
    # atomic optimizer off: it turns the single-lane queue claims into "atomic + wait + readfirstlane", which makes
    # every claim a synchronous memory round trip in front of the tile's loads
    flags = [f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
             "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", *extra_flags]
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    objs = []
    procs = []
    for src in SOURCES:
        s = CSRC / src
        if not s.exists():
            continue
        o = objdir / (s.stem + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cc, *flags, "-c", str(s), "-o", str(o)]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if verbose or p.returncode != 0:
            print(out, flush=True)
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    target = lib_path() if not variant else LIBDIR / f"libechoglad_hip.{variant}.so"
    if force or _stale(target, objs):
        cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(target), *map(str, objs)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return target


if __name__ == "__main__":
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    if "--" in sys.argv:                                   # everything after "--" goes to hipcc verbatim (experiments)
        extra += sys.argv[sys.argv.index("--") + 1:]
    var = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--variant=")]
    p = build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, extra_flags=extra,
              variant=var[0] if var else "")
    print(p)
