#!/usr/bin/env python3
"""Worker of tests/test_exact_gpu.py: the bitwise factor / solve check of the reference-order engine on a few fixtures with whatever PIQP_AMD_DEBUG schedule tokens the
parent put into the environment (the library reads them once per process).

  python tests/workers/exact_variant.py fixture [fixture ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch  # noqa: F401
    import piqp_amd as hip
    from oracle import pyorc as orc
    from test_exact_gpu import _check
    for name in sys.argv[1:]:
        _check(hip, orc, name, hip.SPARSE_LDLT_EXACT)
    print("ok", os.environ.get("PIQP_AMD_DEBUG", ""), len(sys.argv) - 1)


if __name__ == "__main__":
    main()
