"""In-tree build of the C-ABI HIP library (``csrc/libmfc.so``) for gfx950.

``python -m meanflow_audio_codec_amd._build`` or ``__graft_entry__.build()``.
hipcc cross-compiles without a GPU; the built ``.so`` is git-ignored but travels
with the working tree to the GPU box.
"""
from __future__ import annotations

import os
import pathlib
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = pathlib.Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libmfc.so"
ARCH = "gfx950"


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def sources() -> list[pathlib.Path]:
    return sorted(CSRC.glob("*.hip"))


def _stale(out: pathlib.Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True, jobs: int = 4) -> pathlib.Path:
    hipcc = _hipcc()
    headers = list(CSRC.glob("*.h")) + list(CSRC.glob("*.inc")) + list((CSRC.parents[1] / "include").glob("*.h"))
    objs = []
    todo = []
    for src in sources():
        obj = src.with_suffix(".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            todo.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", str(src), "-o", str(obj),
               "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
               "-mllvm", "-amdgpu-mfma-vgpr-form=1",   # MFMA results straight into VGPRs (no v_accvgpr_read traffic)
               # no SLP vectorisation: it turns adjacent scalar f32 ops into v_pk_{mul,add,fma}_f32, which on gfx950
               # issue slower than the two scalar instructions they replace (MI355X_MICROARCH.md, packed f32 VALU) and
               # cost registers; the VALU-bound ConvNeXt kernels run 2-4 % faster without (measured, -3.9 ms / step)
               "-fno-slp-vectorize"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(cc, todo))
    if force or todo or _stale(LIB, objs):
        # no rpath: inside a PyTorch-ROCm process the already-loaded
        # libamdhip64.so.7 (torch's) is reused, so streams are shared.
        rocm_lib = str(pathlib.Path(hipcc).resolve().parents[1] / "lib")
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-no-hip-rt", "-o", str(LIB)] \
            + [str(o) for o in objs] + ["-L" + rocm_lib, "-lamdhip64"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
