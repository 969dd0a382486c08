"""Builds piqp_amd/lib/libpiqp_amd.so with hipcc for gfx950 (in-tree; the .so travels with gpurun)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib")
LIB = os.path.join(OUT, "libpiqp_amd.so")


def sources():
    return sorted(os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith((".hip", ".cpp")))


def headers():
    hs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".hpp")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "piqp_amd.h"))
    return hs


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in sources() + headers())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in sources():
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(f) for f in [s] + headers()):
            objs.append(o)
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-ffp-contract=on",
               "-Wall", "-Wno-unused-function", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + s)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
