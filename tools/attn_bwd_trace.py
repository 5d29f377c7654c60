#!/usr/bin/env python
"""Where a workgroup of the fused attention backward spends its life: per-wave shader-clock timestamps of the phase boundaries
(variant build -DEZ_ATTN_BWD_TRACE: tools/build_variants.py "attntrace@attention_short_bwd.hip:-DEZ_ATTN_BWD_TRACE"), ViT-B/16 shape.
    EZCLIP_LIB=tools/bin/var_attntrace/libezclip_hip.so python tools/attn_bwd_trace.py [B] [L] [H]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easynlp_amd import lib as L  # noqa: E402

lib = L.load()
B, Lq, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), (int(sys.argv[2]) if len(sys.argv) > 2 else 197), (int(sys.argv[3]) if len(sys.argv) > 3 else 12)
D = H * 64
qkv = (torch.randn(B * Lq, 3 * D, device="cuda") * 0.5).bfloat16()
dctx = torch.randn(B * Lq, D, device="cuda").bfloat16()
ctx, lse = L.op_attention(qkv, B, Lq, H, key_bias=None, want_lse=True)
L.check(lib.ezclip_debug_set(11, 0))
base = qkv.data_ptr()
dqkv = torch.zeros_like(qkv)
db = torch.zeros(3 * D, device="cuda")
scratch = torch.empty(B * 3 * D, dtype=torch.float32, device="cuda")


def run():
    dbase, bb = dqkv.data_ptr(), db.data_ptr()
    L.check(lib.ezclip_op_attention_bwd_bias(base, base + D * 2, base + 2 * D * 2, 3 * D, ctx.data_ptr(), dctx.data_ptr(), D, None, lse.data_ptr(),
                                             dbase, dbase + D * 2, dbase + 2 * D * 2, bb, bb + 4 * D, bb + 8 * D, scratch.data_ptr(), B, Lq, H,
                                             L.DTYPE_BF16, None, L.stream_ptr()))


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
print("one launch (with the bias-gradient sum behind it): %.4f ms" % e0.elapsed_time(e1))
nt = (Lq + 31) // 32
WORDS, WAVES, WGS = 10, 9, 16384
raw = np.zeros(WGS * WAVES * WORDS, dtype=np.uint64)
fn = lib.ezclip_dbg_attn_trace
fn.argtypes = [C.c_void_p, C.c_size_t]
assert fn(raw.ctypes.data, raw.nbytes) == 0
t = raw.reshape(WGS, WAVES, WORDS)[:min(WGS, B * H), :nt].astype(np.int64)
n = t.shape[0]
ent, landed_own, landed, pub, enda, endb, ext = t[..., 0], t[..., 9], t[..., 1], t[..., 2], t[..., 3], t[..., 4], t[..., 5]
hw = t[..., 6]
rt0, rt1 = t[..., 7], t[..., 8]
# s_memtime counters are per XCC (not synchronised across XCCs): only differences inside one wave / one CU are meaningful.  Its unit is
# calibrated per wave against the 100 MHz real-time counter over the wave's own life.
life_rt = (rt1 - rt0).astype(np.float64)
ok = life_rt > 50
clk = float(np.median((ext - ent)[ok].astype(np.float64) / (life_rt[ok] / 100.0)))
print("kernel: %d workgroups x %d waves, first entry -> last exit %.1f us real time, %.2f s_memtime ticks per us" % (n, nt, (rt1.max() - rt0.min()) / 100.0, clk))
us = lambda cyc: cyc / clk


def row(name, d):
    d = us(d.reshape(-1).astype(np.float64))
    print("  %-44s mean %7.2f  median %7.2f  p10 %7.2f  p90 %7.2f us" % (name, d.mean(), np.median(d), np.quantile(d, 0.1), np.quantile(d, 0.9)))


print("per wave (all waves of all workgroups):")
row("entry -> own loads landed (vmcnt 0)", landed_own - ent)
row("own loads landed -> barrier passed", landed - landed_own)
row("D / lse -> second barrier passed", pub - landed)
row("pass A (incl. dq stores issued, k / v bias shares)", enda - pub)
row("pass B (incl. dk / dv stores issued, q bias share)", endb - enda)
row("bias-gradient combine -> exit", ext - endb)
row("whole life of a wave", ext - ent)
wg_life = ext.max(axis=1) - ent.min(axis=1)
row("whole life of a workgroup", wg_life)
for w in range(nt):
    print("   wave %d: pass A %.2f us, pass B %.2f us, life %.2f us (means)" % (w, us((enda - pub)[:, w].mean()), us((endb - enda)[:, w].mean()), us((ext - ent)[:, w].mean())))
# per CU: how many workgroups, idle gaps between one workgroup's exit and the next one's entry
xcc = (hw[:, 0] >> 32) & 0xf
cu = (hw[:, 0] >> 8) & 0xf
sh = (hw[:, 0] >> 12) & 1
se = (hw[:, 0] >> 13) & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
ids = np.unique(key)
gaps, busy, per = [], [], []
for k in ids:
    m = np.where(key == k)[0]
    s, e = ent[m].min(axis=1), ext[m].max(axis=1)
    o = np.argsort(s)
    s, e = s[o], e[o]
    per.append(len(m))
    gaps.extend(us((s[1:] - e[:-1]).astype(np.float64)))
    busy.append(float((e - s).sum()) / float(e.max() - s.min()))
gaps = np.array(gaps)
print("CUs seen: %d; workgroups per CU %d..%d; per-CU busy fraction (sum of workgroup lives / span) mean %.3f min %.3f" % (len(ids), min(per), max(per), np.mean(busy), np.min(busy)))
print("gap between a workgroup's exit and the next entry on the same CU: mean %.2f median %.2f p90 %.2f us; negative (overlapping) %.1f %%" % (gaps.mean(), np.median(gaps), np.quantile(gaps, 0.9), 100.0 * (gaps < 0).mean()))
print("XCC ids seen:", sorted(set(int(x) for x in xcc)))
