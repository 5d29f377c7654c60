#!/usr/bin/env python
"""Build experiment variants of libezclip_hip.so under tools/bin/var_<name>/ (same objects, ONE source recompiled with
extra -D flags: gemm8p.hip unless the name says otherwise).  Run a C tool against one with LD_LIBRARY_PATH=tools/bin/var_<name>, a Python
tool / test with EZCLIP_LIB=tools/bin/var_<name>/libezclip_hip.so.

    python tools/build_variants.py name1:-DFOO=1 name2:"-DBAR -DBAZ=2" lnpacked@rowops.hip:-DEZ_LNBWD_PACKED ...
"""
import os
import shlex
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from easynlp_amd.csrc import build as B  # noqa: E402


def main():
    B.build(verbose=True)
    csrc = B.HERE
    for spec in sys.argv[1:]:
        name, _, defs = spec.partition(":")
        name, _, which = name.partition("@")
        which = (which or "gemm8p.hip").split("+")          # name@a.hip+b.hip: several sources recompiled with the flags
        assert all(w in B.SOURCES for w in which), which
        d = os.path.join(HERE, "bin", "var_" + name)
        os.makedirs(d, exist_ok=True)
        objs = []
        for src in B.SOURCES:
            o = os.path.join(csrc, "build", src.replace(".hip", ".o"))
            if src in which:
                o = os.path.join(d, src.replace(".hip", ".o"))
                subprocess.check_call([B.hipcc()] + B.FLAGS + B.FILE_FLAGS.get(src, []) + shlex.split(defs) + ["-c", os.path.join(csrc, src), "-o", o])
            objs.append(o)
        subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(d, "libezclip_hip.so")] + objs)
        print("built", d)


if __name__ == "__main__":
    main()
