#!/usr/bin/env python
"""CPU emulation, lane by lane, of rn_wgrad3x3_c64_kernel's index arithmetic (csrc/resnet_train.hip): the strip -> LDS layout with zero pad
pixels and the slot swizzle, the per-lane addresses of the LDS transpose read (16 lanes read a 4 x 16 block, lane c receives column c), the
tap offsets, the split of the result over waves and the partial sums -- against a direct evaluation of the 3 x 3 weight gradient.  Written
before the kernel's first GPU run (round 5); integer-valued inputs, so a correct emulation prints a maximum error of exactly 0."""
import numpy as np
rng = np.random.default_rng(0)
def slot(row): return ((row >> 1) & 1) | (((row >> 3) & 1) << 1)
def chunk_off(row, ch): return row * 128 + (((ch >> 1) ^ slot(row)) << 5) + ((ch & 1) << 4)
def strip_rows(H, W):          # wgrad64_strip_rows of resnet_train.hip
    Wp = W + 2; best = None; best_cost = 0.0
    for R in range(1, min(H, 16) + 1):
        lr = (R * Wp + 31) // 32 * 32
        b = (lr + 2 * Wp + 2 + lr) * 128
        if b > 78 * 1024: break
        cost = ((H + R - 1) // R) * (lr + 0.5 * (R + 2) * Wp)
        if best is None or cost <= best_cost: best, best_cost = (R, lr, b), cost
    return best
def run(B, H, W, G):
    x = rng.integers(-3, 4, size=(B, H, W, 64)).astype(np.float64)
    dz = rng.integers(-3, 4, size=(B, H, W, 64)).astype(np.float64)
    R, lz_r, _ = strip_rows(H, W)
    Wp = W + 2; nx = lz_r + 2 * Wp + 2
    spi = (H + R - 1) // R; strips = B * spi
    G = min(G, strips)
    total = np.zeros((64, 576))
    for g in range(G):
        lds = np.zeros((nx + lz_r) * 64)   # elements (2 bytes each): index = byte/2
        acc = np.zeros((9, 4, 4, 16, 16))  # tap, wave, ob, o16, c16
        for s in range(strips * g // G, strips * (g + 1) // G):
            b, y0 = divmod(s, spi); y0 *= R
            rc = W * 8
            for q in range((R + 2) * rc):
                ry, rem = divmod(q, rc); xx, ch = rem >> 3, rem & 7; y = y0 - 1 + ry
                v = x[b, y, xx, ch*8:ch*8+8] if 0 <= y < H else np.zeros(8)
                o = chunk_off(ry * Wp + xx + 2, ch) // 2
                lds[o:o+8] = v
            for q in range(R * rc):
                rz, rem = divmod(q, rc); xx, ch = rem >> 3, rem & 7; y = y0 + rz
                v = dz[b, y, xx, ch*8:ch*8+8] if y < H else np.zeros(8)
                o = (nx * 128 + chunk_off(rz * Wp + xx + 1, ch)) // 2
                lds[o:o+8] = v
            def frag(img_px0, first_row, cb):
                # returns F[i=16][k=32]: value for column i of block cb, pixel first_row + k -- through the lane addresses + transpose semantic
                F = np.zeros((16, 32))
                for q4 in range(4):
                    for half in range(2):
                        addr = {}
                        for t in range(16):
                            row = first_row + 8 * q4 + (t >> 2) + 4 * half
                            addr[t] = (img_px0 * 128 + row * 128 + ((cb ^ slot(row)) << 5) + (t & 3) * 8) // 2
                        for i in range(16):
                            for r in range(4):
                                src = addr[r * 4 + i // 4] + (i % 4)
                                F[i, 8 * q4 + 4 * half + r] = lds[src]
                return F
            for k in range(0, lz_r, 32):
                af = [frag(nx, k, ob) for ob in range(4)]
                for tap in range(9):
                    off = (tap // 3) * Wp + tap % 3
                    for wave in range(4):
                        bf = frag(0, k + off, wave)
                        for ob in range(4):
                            acc[tap, wave, ob] += af[ob] @ bf.T     # [o16][c16]
        for tap in range(9):
            for wave in range(4):
                for ob in range(4):
                    total[ob*16:ob*16+16, tap*64 + wave*16: tap*64 + wave*16 + 16] += acc[tap, wave, ob]
    # reference
    ref = np.zeros((64, 576))
    xp = np.zeros((B, H + 2, W + 2, 64)); xp[:, 1:-1, 1:-1] = x
    for tap in range(9):
        ky, kx = divmod(tap, 3)
        sh = xp[:, ky:ky+H, kx:kx+W]      # pixel shifted by (ky-1, kx-1)
        ref[:, tap*64:(tap+1)*64] = np.einsum('bhwo,bhwc->oc', dz, sh)
    print(B, H, W, "R", R, "lz_r", lz_r, "G", G, "max err", np.abs(total - ref).max())
if __name__ == "__main__":
    run(2, 5, 6, 3); run(1, 7, 12, 1); run(2, 3, 40, 5); run(1, 9, 6, 1); run(2, 10, 5, 1)
