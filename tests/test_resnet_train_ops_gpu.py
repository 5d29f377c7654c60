"""ModifiedResNet training path, operator level: every row-matrix kernel of csrc/resnet_train.hip against the step of
oracle/resnet_oracle.py it implements (``train_step_grads_by_steps`` and the data-layout functions), fp32 and bf16, and the device's own
implicit 3x3 convolution / TN product on the packed input-gradient weights and the explicit im2col -- the conventions the tower's
training orchestration will rely on."""
import ctypes as C

import pytest
import torch

from easynlp_amd import lib as L
from oracle import resnet_oracle as RO

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0) if torch.cuda.is_available() else None
DT = {"fp32": (torch.float32, L.DTYPE_F32), "bf16": (torch.bfloat16, L.DTYPE_BF16)}


def _scratch(rows, cp):
    lib = L.load()
    return torch.empty(int(lib.ezclip_op_rn_bn_scratch_bytes(rows, cp)) // 4 + 16, dtype=torch.float32, device=DEV)


def _nhwc(x, cp, dt):
    return RO.to_nhwc(x, cp).to(dt).to(DEV).contiguous()


def _unfold_cols(x_rows, B, H, W, cp):
    """The 3 x 3 column matrix [B * H * W, 9 * cp] (K index (ky * 3 + kx) * cp + c, zero padding 1) of NHWC rows, built by torch's own
    ``F.unfold`` on the CPU in float64 -- NOT by the library's im2col kernel: the reference the weight-gradient kernels are read against."""
    x = x_rows.detach().cpu().double().reshape(B, H, W, cp).permute(0, 3, 1, 2)                     # NCHW
    cols = torch.nn.functional.unfold(x, (3, 3), padding=1)                                         # [B, cp * 9, H * W], index c * 9 + tap
    return cols.reshape(B, cp, 9, H * W).permute(0, 3, 2, 1).reshape(B * H * W, 9 * cp)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,C,H,residual,relu", [(3, 24, 6, False, True), (2, 64, 8, True, True), (5, 300, 4, True, False), (2, 40, 40, False, True)])
def test_batchnorm_train_forward_and_backward(dtype, B, C, H, residual, relu):
    lib = L.load()
    tdt, edt = DT[dtype]
    cp = (C + 63) // 64 * 64
    g = torch.Generator().manual_seed(B * 1000 + C)
    z = (0.7 * torch.randn(B, C, H, H, generator=g) + 0.3).to(tdt).float()          # (values the device will see: rounded once)
    res = torch.randn(B, C, H, H, generator=g).to(tdt).float() if residual else None
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    rm0, rv0 = 0.2 * torch.randn(C, generator=g), 0.5 + torch.rand(C, generator=g)
    dy = torch.randn(B, C, H, H, generator=g).to(tdt).float()
    rows = B * H * H
    # ---- oracle
    y_lin, (xh, rstd_o) = RO._bn_train_fwd(z.double(), gamma.double(), beta.double())
    y_o = y_lin + (res.double() if residual else 0)
    if relu:
        y_o = torch.relu(y_o)
    mean_o = z.double().mean(dim=(0, 2, 3))
    var_o = z.double().var(dim=(0, 2, 3), unbiased=False)
    # ---- device
    pad = lambda v: torch.cat([v, torch.zeros(cp - C)]).to(DEV)
    zd = _nhwc(z, cp, tdt)
    resd = _nhwc(res, cp, tdt) if residual else None
    gd, bd, rmd, rvd = pad(gamma), pad(beta), pad(rm0), pad(rv0)
    yd = torch.full((rows, cp), 7.0, dtype=tdt, device=DEV)
    mean_d, rstd_d = torch.empty(cp, device=DEV), torch.empty(cp, device=DEV)
    sc = _scratch(rows, cp)
    L.check(lib.ezclip_op_rn_bn_train_fwd(L.ptr(zd), rows, C, cp, L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd), 0.1, 1e-5,
                                          L.ptr(resd) if residual else None, 1 if relu else 0, L.ptr(yd), L.ptr(mean_d), L.ptr(rstd_d),
                                          L.ptr(sc), edt, L.stream_ptr()))
    torch.cuda.synchronize()
    assert float((mean_d[:C].cpu().double() - mean_o).abs().max()) < 2e-6
    assert float((rstd_d[:C].cpu().double() - rstd_o).abs().max()) < 2e-5 * float(rstd_o.max())
    assert float(mean_d[C:].abs().max() if cp > C else 0) == 0.0
    n = rows
    want_rm = 0.9 * rm0.double() + 0.1 * mean_o
    want_rv = 0.9 * rv0.double() + 0.1 * var_o * n / (n - 1)
    assert float((rmd[:C].cpu().double() - want_rm).abs().max()) < 1e-6 and float((rvd[:C].cpu().double() - want_rv).abs().max()) < 1e-5
    got = RO.from_nhwc(yd.float().cpu(), B, C, H, H)
    tol = 2e-5 if dtype == "fp32" else 2e-2
    assert float((got.double() - y_o).abs().max()) < tol * max(1.0, float(y_o.abs().max()))
    if cp > C:
        assert float(yd[:, C:].float().abs().max()) == 0.0                      # padded channels stay exactly zero
    # ---- backward (mask from the DEVICE's y: what the tower will hold)
    y_dev_nchw = got.double()
    gm = dy.double() * (y_dev_nchw > 0) if relu else dy.double()
    dz_o, dgam_o, dbet_o = RO._bn_train_bwd(gm, xh, rstd_o, gamma.double())
    dyd = _nhwc(dy, cp, tdt)
    dzd = torch.full((rows, cp), 5.0, dtype=tdt, device=DEV)
    dresd = torch.full((rows, cp), 5.0, dtype=tdt, device=DEV) if residual else None
    dgam = torch.zeros(cp, device=DEV) + 3.0
    dbet = torch.zeros(cp, device=DEV) + 3.0
    L.check(lib.ezclip_op_rn_bn_train_bwd(L.ptr(dyd), L.ptr(yd) if relu else None, L.ptr(zd), rows, C, cp, L.ptr(gd), L.ptr(mean_d),
                                          L.ptr(rstd_d), L.ptr(dzd), L.ptr(dresd) if residual else None, L.ptr(dgam), L.ptr(dbet), 0,
                                          L.ptr(sc), edt, L.stream_ptr()))
    torch.cuda.synchronize()
    rel = 3e-5 if dtype == "fp32" else 2e-2
    assert float((dgam[:C].cpu().double() - dgam_o).abs().max()) < rel * max(1.0, float(dgam_o.abs().max()))
    assert float((dbet[:C].cpu().double() - dbet_o).abs().max()) < rel * max(1.0, float(dbet_o.abs().max()))
    got_dz = RO.from_nhwc(dzd.float().cpu(), B, C, H, H).double()
    assert float((got_dz - dz_o).abs().max()) < rel * max(1.0, float(dz_o.abs().max()))
    if cp > C:
        assert float(dzd[:, C:].float().abs().max()) == 0.0
    if residual:
        assert float((RO.from_nhwc(dresd.float().cpu(), B, C, H, H).double() - gm).abs().max()) == 0.0
    # accumulate = 1 adds onto what is there
    L.check(lib.ezclip_op_rn_bn_train_bwd(L.ptr(dyd), L.ptr(yd) if relu else None, L.ptr(zd), rows, C, cp, L.ptr(gd), L.ptr(mean_d),
                                          L.ptr(rstd_d), L.ptr(dzd), None, L.ptr(dgam), L.ptr(dbet), 1, L.ptr(sc), edt, L.stream_ptr()))
    torch.cuda.synchronize()
    assert float((dbet[:C].cpu().double() - 2 * dbet_o).abs().max()) < 2 * rel * max(1.0, float(dbet_o.abs().max()))
    # bit-reproducible (fixed-order reductions)
    dz2 = torch.empty_like(dzd)
    L.check(lib.ezclip_op_rn_bn_train_bwd(L.ptr(dyd), L.ptr(yd) if relu else None, L.ptr(zd), rows, C, cp, L.ptr(gd), L.ptr(mean_d),
                                          L.ptr(rstd_d), L.ptr(dz2), None, L.ptr(dgam), L.ptr(dbet), 0, L.ptr(sc), edt, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(dz2, dzd)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_avgpool_backward_and_im2col(dtype):
    lib = L.load()
    tdt, edt = DT[dtype]
    B, C, H, cp = 3, 40, 6, 64
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(B, C, H // 2, H // 2, generator=g).to(tdt).float()
    dyd = _nhwc(dy, cp, tdt)
    dxd = torch.empty(B * H * H, cp, dtype=tdt, device=DEV)
    L.check(lib.ezclip_op_rn_avgpool2_bwd(L.ptr(dyd), B, H, H, cp, L.ptr(dxd), edt, L.stream_ptr()))
    want = RO._avgpool_bwd(dy, 2)
    assert float((RO.from_nhwc(dxd.float().cpu(), B, C, H, H) - want).abs().max()) < (1e-7 if dtype == "fp32" else 4e-3)
    x = torch.randn(B, C, H, H, generator=g).to(tdt).float()
    xd = _nhwc(x, cp, tdt)
    cold = torch.full((B * H * H, 9 * cp), 3.0, dtype=tdt, device=DEV)
    L.check(lib.ezclip_op_rn_im2col3x3(L.ptr(xd), B, H, H, cp, L.ptr(cold), edt, L.stream_ptr()))
    assert torch.equal(cold.float().cpu(), RO.im2col3x3_nhwc(RO.to_nhwc(x, cp), B, H, H))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,cp,opad,H,W", [(2, 64, 64, 6, 6), (3, 128, 64, 7, 7), (5, 64, 192, 12, 12), (2, 192, 128, 9, 4),
                                           (37, 64, 64, 28, 28)])
def test_weight_gradient_product_gathers_the_neighbourhoods_itself(dtype, B, cp, opad, H, W):
    """ezclip_op_gemm_tn_conv3x3 (the generic weight-gradient kernel reading the 3 x 3 neighbourhoods of x itself) == ezclip_op_gemm_tn on
    the explicit ezclip_op_rn_im2col3x3 matrix, BIT FOR BIT: the same tile values meet the same MFMA sequence.  Shapes: one and several
    column tiles (9 * cp = 576 ... 1728: a partial last tile), row counts that are not multiples of the 64 / 32-row step, a width of 4 (the
    smallest accepted: one wrap per step), enough rows (37 * 784) that the contraction is split across workgroups, `accumulate`."""
    lib = L.load()
    tdt, edt = DT[dtype]
    rows = B * H * W
    g = torch.Generator().manual_seed(B * 100 + cp + H)
    xd = torch.randn(rows, cp, generator=g).to(tdt).to(DEV)
    dzd = torch.randn(rows, opad, generator=g).to(tdt).to(DEV)
    cold = torch.empty(rows, 9 * cp, dtype=tdt, device=DEV)
    L.check(lib.ezclip_op_rn_im2col3x3(L.ptr(xd), B, H, W, cp, L.ptr(cold), edt, L.stream_ptr()))
    want = torch.full((opad, 9 * cp), 7.0, dtype=torch.float32, device=DEV)
    L.check(lib.ezclip_op_gemm_tn(L.ptr(dzd), opad, L.ptr(cold), 9 * cp, L.ptr(want), 9 * cp, rows, opad, 9 * cp, 0, edt, L.stream_ptr()))
    got = torch.full((opad, 9 * cp), -3.0, dtype=torch.float32, device=DEV)
    L.check(lib.ezclip_op_gemm_tn_conv3x3(L.ptr(dzd), opad, L.ptr(xd), B, H, W, cp, L.ptr(got), 9 * cp, opad, 0, edt, L.stream_ptr()))
    torch.cuda.synchronize()
    ref = (dzd.cpu().double().t() @ _unfold_cols(xd, B, H, W, cp)).to(DEV)          # the column matrix of torch's own unfold (float64, CPU)
    assert float((want.double() - ref).abs().max()) < (1e-3 if dtype == "fp32" else 1e-2) * max(1.0, float(ref.abs().max()))
    assert float((got.double() - ref).abs().max()) < (1e-3 if dtype == "fp32" else 1e-2) * max(1.0, float(ref.abs().max()))
    if rows <= 256:                  # one workgroup per tile: a fixed summation order
        assert torch.equal(got, want)
    else:                            # the contraction is split: float atomics in an order that differs between any two launches
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    L.check(lib.ezclip_op_gemm_tn_conv3x3(L.ptr(dzd), opad, L.ptr(xd), B, H, W, cp, L.ptr(got), 9 * cp, opad, 1, edt, L.stream_ptr()))
    torch.cuda.synchronize()
    assert float((got - 2 * want).abs().max()) <= 4e-5 * max(1.0, float(want.abs().max()))
    # the guard: a width below 4 (the pixel walk allows one wrap per step) is refused, loudly
    assert lib.ezclip_op_gemm_tn_conv3x3(L.ptr(dzd), opad, L.ptr(xd), B, H, 3, cp, L.ptr(got), 9 * cp, opad, 0, edt, L.stream_ptr()) != 0
    assert "W >= 4" in L.last_error()


@pytest.mark.parametrize("B,H,W,cp,opad,parts", [(2, 8, 8, 64, 64, 512), (3, 7, 12, 64, 64, 512), (4, 5, 40, 64, 64, 2), (6, 56, 56, 64, 64, 512),
                                                 (3, 112, 112, 64, 64, 100), (2, 9, 140, 64, 64, 512), (5, 28, 28, 128, 128, 512),
                                                 (2, 10, 6, 128, 64, 7), (3, 6, 9, 64, 128, 512)])
def test_weight_gradient_at_64_channel_blocks_reads_its_operands_once(B, H, W, cp, opad, parts):
    """ezclip_op_rn_wgrad3x3_c64 (per 64-output x 64-channel block the whole 64 x 576 result in registers, strips of image rows through LDS
    with zero pad pixels, the LDS transpose read) against the explicit column matrix + float64: strips of 1 ... 16 rows (the LDS budget and
    the work per image decide), a last strip that runs past the image, non-square images, fewer partial buffers than strips (`parts`),
    128 channels on either side (two / four sub-problems writing their quarters of the result), `accumulate`, bit-reproducibility."""
    lib = L.load()
    tdt, edt = DT["bf16"]
    rows, K = B * H * W, 9 * cp
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + W)
    xd = torch.randn(rows, cp, generator=g).to(tdt).to(DEV)
    dzd = torch.randn(rows, opad, generator=g).to(tdt).to(DEV)
    cols = _unfold_cols(xd, B, H, W, cp)                                        # torch's unfold, float64, CPU
    ref = (dzd.cpu().double().t() @ cols).to(DEV)
    cold = torch.empty(rows, K, dtype=tdt, device=DEV)                          # ... and the library's im2col kernel IS that matrix
    L.check(lib.ezclip_op_rn_im2col3x3(L.ptr(xd), B, H, W, cp, L.ptr(cold), edt, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(cold.cpu().double(), cols)
    del cols
    sub = (cp // 64) * (opad // 64)
    scratch = torch.empty(parts * sub * 64 * 576, dtype=torch.float32, device=DEV)
    ldo = K + 64

    def run(dst, accumulate=0, nbytes=None):
        return lib.ezclip_op_rn_wgrad3x3_c64(L.ptr(xd), L.ptr(dzd), B, H, W, cp, opad, L.ptr(scratch), scratch.numel() * 4 if nbytes is None else nbytes,
                                             L.ptr(dst), ldo, accumulate, L.stream_ptr())

    got = torch.full((opad, ldo), -3.0, dtype=torch.float32, device=DEV)
    L.check(run(got))
    torch.cuda.synchronize()
    scale = max(1.0, float(ref.abs().max()))
    assert float((got[:, :K].double() - ref).abs().max()) < 2e-3 * scale, float((got[:, :K].double() - ref).abs().max()) / scale
    assert float((got[:, K:] + 3.0).abs().max()) == 0.0                       # columns beyond 9 cp are not touched
    again = torch.zeros(opad, ldo, dtype=torch.float32, device=DEV)
    scratch.fill_(float("nan"))
    L.check(run(again))
    torch.cuda.synchronize()
    assert torch.equal(again[:, :K], got[:, :K])
    L.check(run(again, accumulate=1))
    torch.cuda.synchronize()
    assert float((again[:, :K] - 2 * got[:, :K]).abs().max()) <= 1e-5 * scale
    # not this kernel's shape: refused, loudly (no scratch for even one partial per sub-problem)
    assert run(again, nbytes=1000) != 0
    assert "not a shape of this kernel" in L.last_error()


@pytest.mark.parametrize("M,N,K,lda,ldb,parts", [(4096, 64, 64, 64, 64, 512), (5000, 256, 64, 256, 64, 512), (4100, 64, 256, 64, 256, 3),
                                                 (6001, 128, 512, 128, 512, 512), (4097, 512, 128, 640, 192, 512), (7000, 128, 192, 128, 192, 512),
                                                 (50000, 64, 128, 64, 128, 512)])
def test_weight_gradient_of_a_1x1_convolution_with_a_small_result(M, N, K, lda, ldb, parts):
    """ezclip_op_rn_tn_skinny against float64 and ezclip_op_gemm_tn: column blocks of 64 / 128 / 256 (K = 512 -> two of 256, K = 192 -> three of 64),
    row counts that are not multiples of the strip (zero-filled tail), leading dimensions wider than the matrix, fewer partial buffers
    than strips, `accumulate`, bit-reproducibility, refusal of other shapes."""
    lib = L.load()
    tdt, edt = DT["bf16"]
    g = torch.Generator().manual_seed(M + N + K)
    ad = torch.randn(M, lda, generator=g).to(tdt).to(DEV)
    bd = torch.randn(M, ldb, generator=g).to(tdt).to(DEV)
    ref = ad[:, :N].double().t() @ bd[:, :K].double()
    kb = 256 if K % 256 == 0 else 128 if K % 128 == 0 else 64
    sub = (N // 64) * (K // kb)
    scratch = torch.empty(parts * sub * 64 * kb, dtype=torch.float32, device=DEV)
    ldc = K + 4

    def run(dst, accumulate=0, nbytes=None, m=M):
        return lib.ezclip_op_rn_tn_skinny(L.ptr(ad), lda, L.ptr(bd), ldb, L.ptr(dst), ldc, m, N, K, accumulate, L.ptr(scratch),
                                          scratch.numel() * 4 if nbytes is None else nbytes, L.stream_ptr())

    got = torch.full((N, ldc), -3.0, dtype=torch.float32, device=DEV)
    L.check(run(got))
    torch.cuda.synchronize()
    scale = max(1.0, float(ref.abs().max()))
    assert float((got[:, :K].double() - ref).abs().max()) < 2e-3 * scale, float((got[:, :K].double() - ref).abs().max()) / scale
    assert float((got[:, K:] + 3.0).abs().max()) == 0.0
    want = torch.zeros(N, ldc, dtype=torch.float32, device=DEV)
    L.check(lib.ezclip_op_gemm_tn(L.ptr(ad), lda, L.ptr(bd), ldb, L.ptr(want), ldc, M, N, K, 0, edt, L.stream_ptr()))
    torch.cuda.synchronize()
    assert float((got[:, :K] - want[:, :K]).abs().max()) < 1e-3 * scale
    again = torch.zeros(N, ldc, dtype=torch.float32, device=DEV)
    scratch.fill_(float("nan"))
    L.check(run(again))
    torch.cuda.synchronize()
    assert torch.equal(again[:, :K], got[:, :K])
    L.check(run(again, accumulate=1))
    torch.cuda.synchronize()
    assert float((again[:, :K] - 2 * got[:, :K]).abs().max()) <= 1e-5 * scale
    assert run(again, nbytes=1000) != 0 and "not a shape of this kernel" in L.last_error()
    assert run(again, m=1000) != 0 and "not a shape of this kernel" in L.last_error()          # few rows: the generic kernel's case


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("B,I,O,H", [(2, 24, 40, 6), (1, 64, 128, 10)])
def test_convolution_gradients_through_the_device_products(dtype, B, I, O, H):
    """dx = the tower's implicit 3x3 convolution of dz on the packed input-gradient weights; dw = gemm_tn(dz, im2col(x)) unpacked --
    against torch's convolution backward.  Also the 1x1 case (k = 1: the packed input-gradient weight is the transpose)."""
    lib = L.load()
    tdt, edt = DT[dtype]
    cp, opad = (I + 63) // 64 * 64, (O + 63) // 64 * 64
    g = torch.Generator().manual_seed(B * 10 + I)
    x = torch.randn(B, I, H, H, generator=g).to(tdt).float()
    w = (torch.randn(O, I, 3, 3, generator=g) * (I * 9) ** -0.5)
    dz = torch.randn(B, O, H, H, generator=g).to(tdt).float()
    xr, wr = x.clone().requires_grad_(True), w.to(tdt).float().clone().requires_grad_(True)
    (torch.nn.functional.conv2d(xr, wr, padding=1) * dz).sum().backward()
    rows = B * H * H
    zero = torch.zeros(256, dtype=torch.uint8, device=DEV)
    wd = torch.empty(cp, 9 * opad, dtype=tdt, device=DEV)
    L.check(lib.ezclip_op_rn_pack_conv_dgrad(L.ptr(w.to(DEV).contiguous()), O, I, 3, opad, cp, L.ptr(wd), edt, L.stream_ptr()))
    assert torch.equal(wd.float().cpu(), RO.pack_conv3x3_dgrad(w, cp, opad).to(tdt).float())
    dzd = _nhwc(dz, opad, tdt)
    dxd = torch.empty(rows, cp, dtype=tdt, device=DEV)
    L.check(lib.ezclip_op_conv3x3_nhwc(L.ptr(dzd), B, H, H, opad, L.ptr(wd), cp, L.ptr(dxd), L.ptr(zero), edt, L.stream_ptr()))
    rel = 1e-4 if dtype == "fp32" else 2e-2
    got_dx = RO.from_nhwc(dxd.float().cpu(), B, I, H, H)
    assert float((got_dx - xr.grad).abs().max()) < rel * max(1.0, float(xr.grad.abs().max()))
    if cp > I:
        assert float(dxd[:, I:].float().abs().max()) == 0.0
    # weight gradient
    xd = _nhwc(x, cp, tdt)
    cold = torch.empty(rows, 9 * cp, dtype=tdt, device=DEV)
    L.check(lib.ezclip_op_rn_im2col3x3(L.ptr(xd), B, H, H, cp, L.ptr(cold), edt, L.stream_ptr()))
    dwp = torch.zeros(opad, 9 * cp, dtype=torch.float32, device=DEV)
    L.check(lib.ezclip_op_gemm_tn(L.ptr(dzd), opad, L.ptr(cold), 9 * cp, L.ptr(dwp), 9 * cp, rows, opad, 9 * cp, 0, edt, L.stream_ptr()))
    dw = torch.full((O, I, 3, 3), 2.0, dtype=torch.float32, device=DEV)
    L.check(lib.ezclip_op_rn_unpack_wgrad(L.ptr(dwp), 9 * cp, O, I, 3, cp, 0, L.ptr(dw), L.stream_ptr()))
    torch.cuda.synchronize()
    assert float((dw.cpu() - wr.grad).abs().max()) < rel * max(1.0, float(wr.grad.abs().max()))
    L.check(lib.ezclip_op_rn_unpack_wgrad(L.ptr(dwp), 9 * cp, O, I, 3, cp, 1, L.ptr(dw), L.stream_ptr()))
    torch.cuda.synchronize()
    assert float((dw.cpu() - 2 * wr.grad).abs().max()) < 2 * rel * max(1.0, float(wr.grad.abs().max()))
    # 1 x 1: transpose pack, plain products
    w1 = torch.randn(O, I, 1, 1, generator=g) * I ** -0.5
    wd1 = torch.empty(cp, opad, dtype=tdt, device=DEV)
    L.check(lib.ezclip_op_rn_pack_conv_dgrad(L.ptr(w1.to(DEV).contiguous()), O, I, 1, opad, cp, L.ptr(wd1), edt, L.stream_ptr()))
    want1 = torch.zeros(cp, opad)
    want1[:I, :O] = w1[:, :, 0, 0].t()
    assert torch.equal(wd1.float().cpu(), want1.to(tdt).float())
