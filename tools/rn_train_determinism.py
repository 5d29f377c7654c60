#!/usr/bin/env python
"""ModifiedResNet training tower: is the backward pass deterministic?  One forward, then the backward pass THREE times on the same saved
state with EZCLIP_RN_DEBUG=1 (csrc/resnet.hip prints the sum of squares of every intermediate to stderr); this script runs itself in a
child process, groups the lines by pass and prints the first stage whose numbers differ between passes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from easynlp_amd import lib as L
    from easynlp_amd.appzoo.clip.rn_tower import RnEngine
    from oracle import resnet_oracle as RO
    layers, width, e, res, B = (1, 1, 1, 1), int(sys.argv[2]), 128, 64, 4
    sd = RO.make_state_dict(layers, width, e, res, 17)
    g = torch.Generator().manual_seed(6)
    px, probe = torch.randn(B, 3, res, res, generator=g), torch.randn(B, e, generator=g)
    dev = torch.device("cuda", 0)
    eng = RnEngine(layers, width, e, res, L.DTYPE_F32)
    tensors = {n: sd[n].to(dev).contiguous() for n in eng.names}
    eng.sync_train(tensors)
    out = eng.encode_image_train(px.to(dev))
    torch.cuda.synchronize()
    for rep in range(3):
        grads = {n: torch.zeros(eng.shapes[n], dtype=torch.float32, device=dev) for n in eng.names if not eng.is_statistic(n)}
        sys.stderr.write("[rn-dbg] PASS %d\n" % rep)
        sys.stderr.flush()
        eng.backward(out, probe.to(dev), grads)
        torch.cuda.synchronize()
    sys.exit(0)

for width in (64, 48, 32):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(width)], capture_output=True, text=True,
                       env=dict(os.environ, EZCLIP_RN_DEBUG="1"), timeout=600)
    passes, cur = [], None
    for ln in r.stderr.splitlines():
        if ln.startswith("[rn-dbg] PASS"):
            cur = []
            passes.append(cur)
        elif ln.startswith("[rn-dbg]") and cur is not None:
            cur.append(ln)
    print("== width %d: rc %d, %d passes of %s lines" % (width, r.returncode, len(passes), [len(p) for p in passes]))
    if r.returncode != 0:
        print(r.stderr[-1500:])
        continue
    first = None
    for i in range(min(len(p) for p in passes)):
        if len({p[i] for p in passes}) > 1:
            first = i
            break
    if first is None:
        print("   every stage identical over the passes")
    else:
        for j in range(max(0, first - 3), min(first + 4, len(passes[0]))):
            for k, p in enumerate(passes):
                print("   pass %d: %s" % (k, p[j]))
