"""Build libpinnfem_hip.so (gfx950) in-tree with hipcc.

    python -m pinn_fem_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so lands in pinn_fem_amd/lib/ (git-ignored; it
travels to the GPU box with the gpurun snapshot).  One translation unit per padded MLP width
keeps the compile parallel.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "lib", "libpinnfem_hip.so")
STAMP = os.path.join(HERE, "lib", "libpinnfem_hip.stamp")   # hash of everything the library was built from
WIDTHS = (4, 8, 12, 16, 20, 24, 28, 32)
NR_BUCKETS = (2, 4, 6, 8, 10, 12, 15)      # pf_net32.hip: registers per lane (widths <= 2*nr)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, "-I/opt/rocm/include", "-Wall", "-Wno-unused-function"]


def _sources():
    # experiment build: nodes per lane of a node task inside the fused forward launch (pf_net32.h: PF_GU_M)
    gu = ["-DPF_GU_M=" + str(int(os.environ["PINNFEM_GU_M"]))] if os.environ.get("PINNFEM_GU_M") else []
    units = [("pf_api.o", "pf_api.hip", gu), ("pf_mesh.o", "pf_mesh.hip", ["-fno-slp-vectorize"] +
                                                     (["-DPF_RESIDUAL_NODES=" + str(int(os.environ["PINNFEM_RESIDUAL_NODES"]))]
                                                      if os.environ.get("PINNFEM_RESIDUAL_NODES") else [])),
             ("pf_comm.o", "pf_comm.hip", []), ("pf_pcg.o", "pf_pcg.hip", [])]
    # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950's register file is unified), which
    # removes the v_accvgpr_read copies in front of every tanh
    units += [(f"pf_net44_{w}.o", "pf_net44.hip", [f"-DPF_HP={w}", "-mllvm", "-amdgpu-mfma-vgpr-form=1"])
              for w in reversed(WIDTHS)]
    # -fno-slp-vectorize: left on, the SLP vectoriser pairs the kernels' independent fma / mul chains into v_pk_fma_f32 /
    # v_pk_mul_f32, which issue SLOWER than the two scalar instructions they replace on gfx950 (6.6 against 2 x 1.4 cycles
    # per SIMD, tools/valu_rate2.hip) and need register-pair shuffling on top (100 v_mov per 64 elements in the forward
    # kernel, 31 without); registers per lane drop from 108 to 95
    n32 = ["-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1"] + gu
    if os.environ.get("PINNFEM_N32_DBG", "0") == "1":       # timing-experiment build: the PF_N32_DBG knobs are live
        n32.append("-DPF_N32_DBG_ENABLE=1")
    if os.environ.get("PINNFEM_N32_NOPIPE", "0") == "1":    # experiment build: hidden layers as [all MFMAs][all tanh stages]
        n32.append("-DPF_N32_PIPE=0")
    units += [(f"pf_net32_{r}.o", "pf_net32.hip", [f"-DPF_NR={r}"] + n32) for r in reversed(NR_BUCKETS)]
    units += [(f"pf_net32b_{r}.o", "pf_net32.hip", [f"-DPF_NR={r}", "-DPF_PREC=1"] + n32) for r in reversed(NR_BUCKETS)]
    # the fused two-net kernels (E net of bucket r, A net of any bucket): translation units of their own
    fused = ["-DPF_N32_PART=1"]
    if os.environ.get("PINNFEM_BW_NOPARK", "0") == "1":     # experiment build: block barrier between the fused backward's phases
        fused.append("-DPF_BW_PARK=0")
    if os.environ.get("PINNFEM_BW2_NOPAIR", "0") == "1":    # experiment build: the fused backward recomputes tile by tile
        fused += ["-DPF_BW_PAIR=0", "-DPF_BW_MAX_THREADS=512"]   # (~190 registers: room for a co-resident memory-bound kernel)
    units += [(f"pf_net32f_{r}.o", "pf_net32.hip", [f"-DPF_NR={r}"] + fused + n32) for r in reversed(NR_BUCKETS)]
    units += [(f"pf_net32fb_{r}.o", "pf_net32.hip", [f"-DPF_NR={r}", "-DPF_PREC=1"] + fused + n32)
              for r in reversed(NR_BUCKETS)]
    units += [(f"pf_net_{w}.o", "pf_net.hip", [f"-DPF_HP={w}"]) for w in WIDTHS]
    return units


def _input_paths():
    paths = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    paths.append(os.path.join(ROOT, "include", "pinnfem_hip.h"))
    return paths


def inputs_hash() -> str:
    """Content hash of every kernel source, the header and the compile recipe.  File times do not survive a
    snapshot copy to another machine; contents do."""
    h = hashlib.sha256()
    for p in _input_paths():
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(repr((FLAGS[:4], [(u[0], u[1], u[2]) for u in _sources()])).encode())
    return h.hexdigest()


def up_to_date() -> bool:
    try:
        with open(STAMP) as f:
            return os.path.exists(LIB) and f.read().strip() == inputs_hash()
    except OSError:
        return False


def _newest_input(src):
    """Newest of one unit's own source and the headers (every unit includes them)."""
    paths = [os.path.join(CSRC, src)] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    paths.append(os.path.join(ROOT, "include", "pinnfem_hip.h"))
    paths.append(os.path.abspath(__file__))      # the compile recipe
    return max(os.path.getmtime(p) for p in paths)


def _obj_path(unit):
    """Object file of a unit; the unit's compile flags are part of the NAME, so an object built with another define set
    (e.g. the PINNFEM_N32_DBG=1 timing build) is never mistaken for this one on its mtime alone."""
    obj, _, defs = unit
    tag = hashlib.sha256(repr((FLAGS[:4], defs)).encode()).hexdigest()[:10]
    stem, ext = os.path.splitext(obj)
    return os.path.join(OBJ, f"{stem}.{tag}{ext}")


def _compile(unit):
    obj, src, defs = unit
    out = _obj_path(unit)
    if os.path.exists(out) and os.path.getmtime(out) >= _newest_input(src):
        return obj, 0, ""
    cmd = [HIPCC, *FLAGS, *defs, "-c", os.path.join(CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return obj, r.returncode, r.stdout + r.stderr


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and up_to_date():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    # one builder at a time (several ranks of one job may find the library stale together)
    with open(os.path.join(os.path.dirname(LIB), ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and up_to_date():
            return LIB
        return _build_locked(force, verbose)


def _build_locked(force: bool, verbose: bool) -> str:
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    units = _sources()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for obj, rc, log in ex.map(_compile, units):
            if verbose and log.strip():
                print(f"[{obj}] {log.strip()}", file=sys.stderr)
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {obj}:\n{log}")
    objs = [_obj_path(u) for u in units]
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    with open(STAMP, "w") as f:
        f.write(inputs_hash())
    if verbose:
        print(f"built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
