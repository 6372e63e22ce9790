"""Build libpinnfem_hip.so (gfx950) in-tree with hipcc.

    python -m pinn_fem_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so lands in pinn_fem_amd/lib/ (git-ignored; it
travels to the GPU box with the gpurun snapshot).  One translation unit per padded MLP width
keeps the compile parallel.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "lib", "libpinnfem_hip.so")
WIDTHS = (4, 8, 12, 16, 20, 24, 28, 32)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, "-I/opt/rocm/include", "-Wall", "-Wno-unused-function"]


def _sources():
    units = [("pf_api.o", "pf_api.hip", []), ("pf_mesh.o", "pf_mesh.hip", []), ("pf_comm.o", "pf_comm.hip", []), ("pf_pcg.o", "pf_pcg.hip", [])]
    # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950's register file is unified), which
    # removes the v_accvgpr_read copies in front of every tanh
    units += [(f"pf_net44_{w}.o", "pf_net44.hip", [f"-DPF_HP={w}", "-mllvm", "-amdgpu-mfma-vgpr-form=1"])
              for w in reversed(WIDTHS)]
    units += [(f"pf_net_{w}.o", "pf_net.hip", [f"-DPF_HP={w}"]) for w in WIDTHS]
    units += [(f"pf_net16_{w}.o", "pf_net16.hip", [f"-DPF_HP={w}", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]) for w in WIDTHS]
    return units


def _newest_input():
    paths = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    paths.append(os.path.join(ROOT, "include", "pinnfem_hip.h"))
    return max(os.path.getmtime(p) for p in paths)


def _compile(unit):
    obj, src, defs = unit
    out = os.path.join(OBJ, obj)
    if os.path.exists(out) and os.path.getmtime(out) >= _newest_input():
        return obj, 0, ""
    cmd = [HIPCC, *FLAGS, *defs, "-c", os.path.join(CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return obj, r.returncode, r.stdout + r.stderr


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    if (not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_input()):
        return LIB
    units = _sources()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for obj, rc, log in ex.map(_compile, units):
            if verbose and log.strip():
                print(f"[{obj}] {log.strip()}", file=sys.stderr)
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {obj}:\n{log}")
    objs = [os.path.join(OBJ, u[0]) for u in units]
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(f"built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
