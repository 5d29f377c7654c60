#!/usr/bin/env python
"""Build the stand-alone GPU tools under tools/bin (they link against the in-tree libezclip_hip.so)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from easynlp_amd.csrc import build as B  # noqa: E402


def main():
    lib = B.build(verbose=True)
    os.makedirs(os.path.join(HERE, "bin"), exist_ok=True)
    srcs = [("gemm_bench.hip", [])]
    for src, extra in srcs:
        out = os.path.join(HERE, "bin", os.path.basename(src).replace(".hip", ""))
        cmd = [B.hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17"] + extra + [os.path.join(HERE, src), "-o", out,
               "-L" + os.path.dirname(lib), "-lezclip_hip", "-Wl,-rpath,$ORIGIN/../../easynlp_amd/csrc"]
        subprocess.check_call(cmd)
        print("built", out)


if __name__ == "__main__":
    main()
