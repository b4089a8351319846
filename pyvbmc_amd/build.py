"""Build recipe for libvbmc_hip.so (gfx950 only, in-tree).

    python -m pyvbmc_amd.build           # rebuild if sources are newer
    python -m pyvbmc_amd.build --force

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels
to the GPU box with the working-tree snapshot.
"""
import os
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIB = HERE / "libvbmc_hip.so"
SOURCES = [
    "ctx.hip",
    "entropy.hip",
    "entropy_mfma.hip",
    "api_entropy.hip",
    "mixture.hip",
    "gp.hip",
    "api_gp.hip",
    "api_elbo.hip",
    "comm.hip",
]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",  # explicit fma() only: keep the arithmetic reproducible
    "-Wall",
    "-Wno-unused-function",
]


def _sources():
    return [CSRC / s for s in SOURCES if (CSRC / s).exists()]


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = _sources() + list(CSRC.glob("*.h")) + [HERE.parent / "include" / "vbmc_hip.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    bdir = CSRC / "_obj"
    bdir.mkdir(exist_ok=True)
    for src in _sources():
        obj = bdir / (src.stem + ".o")
        cmd = [HIPCC, *FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src.name} failed ---\n{out.decode()}\n")
        elif verbose and out.strip():
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB),
           "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
