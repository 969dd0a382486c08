"""Builds piqp_amd/lib/libpiqp_amd.so with hipcc for gfx950 (in-tree; the .so travels with gpurun)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib")
LIB = os.path.join(OUT, "libpiqp_amd.so")


# Files whose floating-point expressions are evaluated as written -- every product and every sum rounded on its own, like the reference built for plain
# x86-64 (no FMA instructions) and like the CPU oracle's restatement (oracle/Makefile: -ffp-contract=off): the interior-point loop, the KKTSystem shell, the
# sparse mat-vecs, the equilibration and the reference-order sparse engine.  With the sparse_ldlt engine of sparse_exact.hip a whole solve is then the same
# sequence of IEEE operations as the oracle's (tests/test_exact_gpu.py).  The dense / multifrontal / multistage / batched kernels keep contraction: their
# sums are re-associated for the matrix cores anyway and they are held to the 1e-10 residual bar, not to bits.
NO_CONTRACT = {"kkt_system.hip", "device_ipm.hip", "sparse_ops.hip", "sparse_exact.hip", "ruiz_kernels.hip", "solver.cpp"}


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
    if any(os.path.getmtime(f) > t for f in sources() + headers()):
        return True
    # (objects without the record of their command line -- built by an older build.py -- are rebuilt once)
    return any(not os.path.exists(os.path.join(OUT, os.path.basename(s) + ".o.cmd")) for s in sources())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in sources():
        o = os.path.join(OUT, os.path.basename(s) + ".o")
        contract = "off" if os.path.basename(s) in NO_CONTRACT else "on"
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-ffp-contract=" + contract,
               "-Wall", "-Wno-unused-function", "-c", s, "-o", o]
        # an object is reused only if it is newer than its sources AND was built with this very command line (the flags are part of the contract: an object
        # of one of the NO_CONTRACT files built with contraction on would silently break the bitwise parity of the reference-order path)
        stamp = o + ".cmd"
        same_cmd = os.path.exists(stamp) and open(stamp).read() == " ".join(cmd)
        if not force and same_cmd and os.path.exists(o) and os.path.getmtime(o) > max(os.path.getmtime(f) for f in [s] + headers()):
            objs.append(o)
            continue
        if verbose:
            print(" ".join(cmd), flush=True)
        if os.path.exists(stamp):
            os.remove(stamp)
        procs.append((s, subprocess.Popen(cmd), stamp, " ".join(cmd)))
        objs.append(o)
    for s, p, stamp, line in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + s)
        with open(stamp, "w") as f:
            f.write(line)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
