#!/usr/bin/env python3
"""Run the reference's example JSONs end to end through the drop-in CLI (generic.py) on the GPU and
print wall time / iterations / final displacements next to the reference's CPU numbers stored in the
golden files (tests/golden/run_*.json, measured with the reference itself in the build container)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    names = sys.argv[1:] or ["example2", "example2-P", "example3", "example3-P", "example4", "example4-P",
                             "example7", "example7-P"]
    tmp = tempfile.mkdtemp(prefix="pf_examples_")
    rows = []
    for ex in names:
        src = os.path.join(GOLD, "inputs", ex + ".json")
        dst = os.path.join(tmp, ex + ".json")
        shutil.copy(src, dst)
        env = dict(os.environ, PINNFEM_QUIET="1")
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "generic.py"), dst], env=env,
                           capture_output=True, text=True)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            rows.append({"example": ex, "error": r.stdout[-500:] + r.stderr[-500:]})
            continue
        with open(os.path.join(tmp, ex + ".res.json")) as f:
            out = json.load(f)
        with open(os.path.join(GOLD, f"run_{ex}.json")) as f:
            ref = json.load(f)
        # solver-only time: the log has timestamps, but simplest is a second in-process timing
        rows.append({"example": ex, "wall_s_process": round(wall, 2), "converged": out["converged"],
                     "last_increment_iterations": out["iterations"],
                     "reference_last_increment_iterations": ref["result"]["iterations"],
                     "reference_solver_wall_s_cpu": round(ref["wall_s_reference_cpu"], 1),
                     "ux": [round(v, 6) for v in out["displacements"][0::2]],
                     "reference_ux": [round(v, 6) for v in ref["result"]["displacements"][0::2]]})
    print(json.dumps(rows, indent=1))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
