"""Build libamc.so (HIP kernels + C-ABI, gfx950) in-tree with hipcc.

`python -m pycolmap_amd.build` or `pycolmap_amd.build.build_all()`.  hipcc cross-compiles for
gfx950 without a GPU, so this runs in the CPU-only container; the resulting .so files travel to
the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
OBJ = CSRC / "_obj"
LIB = PKG / "libamc.so"

HIP_SOURCES = ["amc_api.hip", "amc_comm.hip", "match_common.hip", "match_dot4.hip", "match_guided.hip", "match_mfma.hip", "tvg_e.hip", "tvg_fh.hip", "tvg_e_big.hip",
               "tvg_fh_big.hip", "pose.hip", "camera.hip"]
HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",  # host<->device FP64 bit parity for the verification kernels
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: libamc.so cannot be built (there is no CPU fallback)")


def _digest(deps: list[Path], extra: str = "") -> str:
    """Content hash of a target's inputs (sources, headers, flags).  Staleness is decided by content, not by
    modification times: a checkout, a copy to the GPU box or a clock skew can leave an .so that is newer than
    sources it was not built from."""
    import hashlib
    h = hashlib.sha256(extra.encode())
    for d in sorted(deps, key=str):
        if d.exists():
            h.update(str(d.name).encode())
            h.update(d.read_bytes())
    return h.hexdigest()


def _stale(target: Path, deps: list[Path], extra: str = "") -> bool:
    """True when `target` is missing or was built from other inputs (its .inputs stamp differs)."""
    stamp = target.with_name(target.name + ".inputs")
    return not target.exists() or not stamp.exists() or stamp.read_text() != _digest(deps, extra)


def _stamp(target: Path, deps: list[Path], extra: str = "") -> None:
    target.with_name(target.name + ".inputs").write_text(_digest(deps, extra))


def build_libamc(force: bool = False, verbose: bool = True) -> Path:
    hipcc = _hipcc()
    OBJ.mkdir(exist_ok=True)
    headers = list(CSRC.glob("*.h")) + list((ROOT / "include").glob("*.h"))
    objs, jobs = [], []
    for name in HIP_SOURCES:
        src = CSRC / name
        if not src.exists():
            continue
        obj = OBJ / (src.stem + ".o")
        flags = " ".join(HIPCC_FLAGS)
        deps = [src] + headers
        if src.stem.endswith("_big"):  # a second build of another source file (#include "tvg_e.hip"): that file is an input too
            deps.append(CSRC / (src.stem[:-4] + ".hip"))
        if force or _stale(obj, deps, flags):
            jobs.append((src, obj, deps, flags))
        objs.append(obj)

    def compile_one(job):
        src, obj, deps, flags = job
        cmd = [hipcc, *HIPCC_FLAGS, "-c", str(src), "-o", str(obj)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        _stamp(obj, deps, flags)

    if jobs:  # the translation units are independent: one hipcc per core, up to eight
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1, len(jobs)))) as pool:
            list(pool.map(compile_one, jobs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        _stamp(LIB, objs)
    return LIB


HOST_SOURCES = ["module.cc", "controller.cc", "database.cc"]


def build_host(force: bool = False, verbose: bool = True) -> Path:
    """The C++/pybind11 host layer (pycolmap API surface) -> pycolmap_amd/_pycolmap*.so, linked
    against libamc.so (C ABI) and the system SQLite."""
    import sysconfig

    import pybind11
    host = CSRC / "host"
    ext = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    out = PKG / f"_pycolmap{ext}"
    srcs = [host / n for n in HOST_SOURCES]
    deps = srcs + list(host.glob("*.h")) + list((ROOT / "include").glob("*.h")) + [LIB]
    if not force and not _stale(out, deps):
        return out
    sqlite_inc = next((p for p in ("/usr/include", "/opt/conda/include") if (Path(p) / "sqlite3.h").exists()), None)
    if sqlite_inc is None:
        raise RuntimeError("sqlite3.h not found")
    cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wall",
           f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}", f"-I{sqlite_inc}",
           *map(str, srcs), f"-L{PKG}", "-l:libamc.so", "-l:libsqlite3.so.0", "-Wl,-rpath,$ORIGIN",
           "-o", str(out)]
    if verbose:
        print("[build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    _stamp(out, deps)
    return out


def build_all(force: bool = False, verbose: bool = True) -> None:
    """The product: libamc.so and the host layer.  (The CPU oracle is test infrastructure and is built by
    __graft_entry__.build() / tests/oracle_lib.py, not from inside the package.)"""
    build_libamc(force=force, verbose=verbose)
    build_host(force=force, verbose=verbose)


def clean_variants(verbose: bool = True) -> list[str]:
    """Remove what the A/B tools leave under csrc/_obj/ beside the product's own objects: variant libraries
    (tools/variant_build_tvg.sh, tools/diag_build.sh: libamc_<name>.so), their objects (<file>_<name>.o), the previous
    revision's source copies and library (tools/ab_prev_lib.sh: prev_src/, libamc_prev.so; tools/ab_build.sh: *_prev.hip).
    They are git-ignored but ride to the GPU box with every `gpurun` push and confuse anyone grepping csrc/ - run this
    when an A/B is over (`python -m pycolmap_amd.build clean-variants`).  Returns the names it removed."""
    keep = {Path(n).stem + ".o" for n in HIP_SOURCES} | {Path(n).stem + ".o.inputs" for n in HIP_SOURCES}
    gone = []
    if not OBJ.exists():
        return gone
    for f in sorted(OBJ.iterdir()):
        if f.name in keep:
            continue
        if f.is_dir():
            shutil.rmtree(f)
        else:
            f.unlink()
        gone.append(f.name)
    if verbose:
        print(f"[build] removed {len(gone)} A/B artefacts from {OBJ.relative_to(ROOT)}: {' '.join(gone[:12])}{' ...' if len(gone) > 12 else ''}")
    return gone


if __name__ == "__main__":
    if "clean-variants" in sys.argv:
        clean_variants()
    else:
        build_all(force="--force" in sys.argv)
