"""
Build ``libtapqir_hip.so`` in-tree for gfx950 (MI355X):  ``python -m tapqir_amd.build``.

Plain ``hipcc`` on the sources in ``tapqir_amd/csrc`` -- no torch extension machinery, the
library has a C ABI (include/tapqir_hip.h) and no torch types.
"""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libtapqir_hip.so")
SOURCES = ["tq_ksmogn.hip", "tq_xtalk.hip", "tq_cosmos.hip", "tq_glimpse.hip", "tq_aux.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "tapqir_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for src in SOURCES:  # one hipcc per translation unit, side by side
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        jobs.append((cmd, obj, subprocess.Popen(cmd)))
    for cmd, _, proc in jobs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [obj for _, obj, _ in jobs]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
