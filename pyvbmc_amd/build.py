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
    "entropy_small.hip",
    "prep.hip",
    "api_entropy.hip",
    "mixture.hip",
    "gp.hip",
    "api_gp.hip",
    "api_elbo.hip",
    "api_batch.hip",
    "adam.hip",
    "adam_fused.hip",
    "api_acq.hip",
    "api_acq_is.hip",
    "sample.hip",
    "comm.hip",
    "host_randn.hip",
    "device_randn.hip",
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
    # a kernel that misses its __launch_bounds__ occupancy ("failed to meet occupancy target": one more register array in a
    # rider path once cost the headline entropy kernel its second wave per SIMD, silently) or a "#pragma unroll" the
    # compiler does not honour fails the build
    "-Werror=pass-failed",
]


WS_DPS = [2, 4, 6, 8, 10, 12, 16, 20, 24, 32]  # must match VBMC_WS_DPS in entropy_args.h
MFMA_DPS = [12, 16, 20, 24, 32]  # padded D of the entropy kernel's matrix-pipe form (entropy_mfma.hip), one object each
WS_EXTRA = os.environ.get("VBMC_WS_EXTRA_FLAGS", "").split()  # experiments on the entropy kernel only
ALL_EXTRA = os.environ.get("VBMC_EXTRA_FLAGS", "").split()    # experiments: flags for every translation unit


def _sources():
    return [CSRC / s for s in SOURCES if (CSRC / s).exists()]


SOURCE_FLAGS = {}  # per-source extra flags (file name -> list)


def _jobs(bdir):
    """(source, object, extra flags) for every translation unit; the wave-split entropy
    kernel is compiled once per padded D so the instantiations build in parallel."""
    jobs = [(src, bdir / (src.stem + ".o"), list(ALL_EXTRA) + SOURCE_FLAGS.get(src.name, [])) for src in _sources()]
    for dp in WS_DPS:
        jobs.append((CSRC / "entropy_ws.hip", bdir / f"entropy_ws_dp{dp}.o", [f"-DVBMC_DP={dp}"] + WS_EXTRA))
    for dp in MFMA_DPS:
        jobs.append((CSRC / "entropy_mfma.hip", bdir / f"entropy_mfma_dp{dp}.o", [f"-DVBMC_MFMA_DP={dp}"] + ALL_EXTRA))
    return jobs


def needs_build():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = _sources() + [CSRC / "entropy_ws.hip", CSRC / "entropy_mfma.hip"] + list(CSRC.glob("*.h")) + [HERE.parent / "include" / "vbmc_hip.h"]
    return any(p.stat().st_mtime > t for p in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    bdir = CSRC / "_obj"
    bdir.mkdir(exist_ok=True)
    import concurrent.futures as cf

    def run(job):
        src, obj, extra = job
        cmd = [HIPCC, *FLAGS, *extra, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        return obj, p

    jobs = _jobs(bdir)
    with cf.ThreadPoolExecutor(max_workers=max(2, (os.cpu_count() or 4))) as ex:
        results = list(ex.map(run, jobs))
    for (src, obj, _), (_, p) in zip(jobs, results):
        objs.append(obj)
        procs.append((obj, p))
    failed = False
    for src, p in procs:
        out = p.stdout
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src.name} failed ---\n{out.decode()}\n")
        elif verbose and out.strip():
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(LIB),
           "-ldl", "-Wl,-rpath,/opt/rocm/lib"]  # librccl is dlopen'ed lazily (csrc/comm.hip)
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
