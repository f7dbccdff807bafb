"""GPU parity of every HIP kernel (called through the C ABI via mimic_amd.ops) against the plain
PyTorch fp32 emulation of the same op (tests/torch_backend.py, evaluated on the CPU).

Tolerance: fp32 everywhere.  The MFMA path is an exact k-ordered fp32 fma chain; the CPU reference sums
in a different order, so results agree to ~1e-6 relative per accumulated term:
    |hip - ref| <= 2e-4 * max|ref| + 2e-4 * |ref|   (tensors),   1e-4 relative (double statistics).
"""
import math
import os
import zlib

import numpy as np
import pytest
import torch

import torch_backend as TB
from mimic_amd import ops
from mimic_amd.ops import Bn, Geom, Mask

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _log(msg):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/op_parity.log", "a") as f:
        f.write(msg + "\n")


def check(name, got, ref, rtol=2e-4, atol_rel=2e-4):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs()
    bound = atol_rel * scale + rtol * ref.abs()
    worst = (err / bound.clamp_min(1e-30)).max().item() if err.numel() else 0.0
    _log(f"{name}: max_abs_err={err.max().item():.3e} scale={scale:.3e} worst/bound={worst:.3f}")
    assert torch.isfinite(got).all(), name
    assert worst <= 1.0, f"{name}: max err {err.max().item():.3e} vs scale {scale:.3e} (ratio {worst:.2f})"


def to_dev(x):
    if isinstance(x, torch.Tensor):
        return x.to(DEV)
    if isinstance(x, Bn):
        return Bn(to_dev(x.gamma), to_dev(x.beta), x.mode, None if x.sums is None else to_dev(x.sums), x.count,
                  None if x.rmean is None else to_dev(x.rmean), None if x.rvar is None else to_dev(x.rvar), x.eps)
    if isinstance(x, Mask):
        return Mask(to_dev(x.mask), x.kind, x.rows_per_sample)
    return x


def make_bn(c, rows, mode, gen, x=None):
    gamma = 1 + 0.3 * torch.randn(c, generator=gen)
    beta = 0.2 * torch.randn(c, generator=gen)
    if mode == 1:
        if x is None:
            x = torch.randn(rows, c, generator=gen)
        x2 = x.reshape(-1, c).double()
        sums = torch.stack([x2.sum(0), (x2 * x2).sum(0)])
        return Bn(gamma, beta, 1, sums=sums, count=x2.shape[0])
    return Bn(gamma, beta, 2, rmean=0.1 * torch.randn(c, generator=gen), rvar=0.5 + torch.rand(c, generator=gen))


# (name, Geom) -- every geometry family the four networks use, plus ragged / tiny-channel cases
GEOMS = [
    ("enc_k4s2p1_64to128", Geom(3, 8, 8, 16, 16, 64, 128, 4, 4, 2, 2, 1, 1, False)),
    ("enc_k4s2p1_192to256_b5", Geom(5, 4, 4, 8, 8, 192, 256, 4, 4, 2, 2, 1, 1, False)),
    ("enc_k4s2p0_4to1", Geom(6, 1, 1, 4, 4, 320, 320, 4, 4, 2, 2, 0, 0, False)),
    ("enc_k4s4p1_16to4", Geom(2, 4, 4, 16, 16, 64, 80, 4, 4, 4, 4, 1, 1, False)),
    ("enc_1x1_128", Geom(2, 16, 16, 16, 16, 128, 128, 1, 1, 1, 1, 0, 0, False)),
    ("stem_k3s2_cin1", Geom(2, 16, 16, 32, 32, 1, 64, 3, 3, 2, 2, 1, 1, False)),
    ("stem_k3s2_cin1_ragged_rows", Geom(3, 5, 16, 10, 32, 1, 64, 3, 3, 2, 2, 1, 1, False)),   # 240 pixels: partial MFMA tile
    ("stem_k3s2_cin1_c32", Geom(2, 8, 8, 16, 16, 1, 32, 3, 3, 2, 2, 1, 1, False)),              # C != 64: streaming kernels
    ("linear_320to128", Geom(7, 1, 1, 1, 1, 320, 128, 1, 1, 1, 1, 0, 0, False)),
    ("tiny_c4_k4s2p1", Geom(4, 16, 16, 32, 32, 4, 8, 4, 4, 2, 2, 1, 1, False)),
    ("tiny_c20_k4s2p0", Geom(4, 1, 1, 4, 4, 16, 20, 4, 4, 2, 2, 0, 0, False)),
    ("text_conv1d_k4s2p1", Geom(3, 1, 64, 1, 128, 128, 128, 1, 4, 1, 2, 0, 1, False)),
    ("text_conv1d_to1", Geom(5, 1, 1, 1, 2, 512, 640, 1, 4, 1, 2, 0, 1, False)),
    ("text_vocab_k1", Geom(2, 1, 128, 1, 128, 128, 3517, 1, 1, 1, 1, 0, 0, False)),
    ("dec_k4s2p1_128to64", Geom(3, 8, 8, 16, 16, 128, 64, 4, 4, 2, 2, 1, 1, True)),
    ("dec_k4_from1x1", Geom(5, 1, 1, 4, 4, 320, 256, 4, 4, 4, 4, 0, 0, True)),
    ("dec_1x1", Geom(2, 8, 8, 8, 8, 192, 192, 1, 1, 1, 1, 0, 0, True)),
    ("dec_head_k3s2p1op1_cout1", Geom(2, 16, 16, 32, 32, 64, 1, 3, 3, 2, 2, 1, 1, True)),
    ("dec_head_k3s2p1op1_ragged_rows", Geom(3, 5, 16, 10, 32, 64, 1, 3, 3, 2, 2, 1, 1, True)),
    ("tiny_dec_c8_k4s2p1", Geom(4, 4, 4, 8, 8, 8, 4, 4, 4, 2, 2, 1, 1, True)),
    ("text_convT1d_k4s2p1", Geom(3, 1, 16, 1, 32, 640, 512, 1, 4, 1, 2, 0, 1, True)),
    ("text_convT1d_from1", Geom(4, 1, 1, 1, 4, 640, 640, 1, 4, 1, 4, 0, 0, True)),
]


@pytest.mark.parametrize("name,g", GEOMS, ids=[n for n, _ in GEOMS])
def test_conv_family(name, g: Geom):
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)   # (str hashes change per process)
    x = torch.randn(g.in_shape, generator=gen)
    wp = torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)
    bias = 0.1 * torch.randn(g.Cout, generator=gen)
    rows_in = x.numel() // g.Cin
    rows_out = math.prod(g.out_shape[:3])
    rps_out = rows_out // g.N
    # ---- forward, plain
    y_ref = TB.conv_fwd(x, wp, g)
    y = ops.conv_fwd(x.to(DEV), wp.to(DEV), g)
    check(f"{name}/fwd", y, y_ref)
    # ---- forward with every fused option (train-mode BN, bias, channel mask, statistics)
    for mode in (1, 2):
        bn = make_bn(g.Cin, rows_in, mode, gen, x if mode == 1 else None)
        cmask = Mask((torch.rand(g.N, g.Cout, generator=gen) < 0.5).float() * 2, 1, rps_out)
        st_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
        y_ref = TB.conv_fwd(x, wp, g, bn_in=bn, bias=bias, mask=cmask, out_stats=st_ref)
        st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
        y = ops.conv_fwd(x.to(DEV), wp.to(DEV), g, bn_in=to_dev(bn), bias=bias.to(DEV), mask=to_dev(cmask), out_stats=st)
        check(f"{name}/fwd_fused_bn{mode}", y, y_ref)
        check(f"{name}/fwd_fused_bn{mode}/stats", st, st_ref, rtol=1e-4, atol_rel=1e-4)
    # elementwise mask
    emask = Mask((torch.rand(g.out_shape, generator=gen) < 0.5).float() * 2, 2, rps_out)
    y_ref = TB.conv_fwd(x, wp, g, bias=bias, mask=emask)
    y = ops.conv_fwd(x.to(DEV), wp.to(DEV), g, bias=bias.to(DEV), mask=to_dev(emask))
    check(f"{name}/fwd_emask", y, y_ref)
    # ---- dgrad
    dy = torch.randn(g.out_shape, generator=gen)
    dx_ref = TB.conv_dgrad(dy, wp, g)
    dx = ops.conv_dgrad(dy.to(DEV), wp.to(DEV), g)
    check(f"{name}/dgrad", dx, dx_ref)
    for mode in (1, 2):
        bn = make_bn(g.Cin, rows_in, mode, gen, x if mode == 1 else None)
        s_ref = torch.zeros(2, g.Cin, dtype=torch.float64)
        dx_ref = TB.conv_dgrad(dy, wp, g, relu_bn=bn, xin=x, bwd_sums=s_ref)
        s = torch.zeros(2, g.Cin, dtype=torch.float64, device=DEV)
        dx = ops.conv_dgrad(dy.to(DEV), wp.to(DEV), g, relu_bn=to_dev(bn), xin=x.to(DEV), bwd_sums=s)
        check(f"{name}/dgrad_relubn{mode}", dx, dx_ref)
        check(f"{name}/dgrad_relubn{mode}/sums", s, s_ref, rtol=2e-4, atol_rel=2e-4)
    # ---- wgrad
    dw_ref = TB.conv_wgrad(x, dy, g)
    dw = ops.conv_wgrad(x.to(DEV), dy.to(DEV), g)
    check(f"{name}/wgrad", dw, dw_ref)
    bn = make_bn(g.Cin, rows_in, 1, gen, x)
    dw_ref = TB.conv_wgrad(x, dy, g, bn_in=bn)
    dw = ops.conv_wgrad(x.to(DEV), dy.to(DEV), g, bn_in=to_dev(bn))
    check(f"{name}/wgrad_bn", dw, dw_ref)


@pytest.mark.parametrize("name,g", [
    ("narrow_many_rows_C", Geom(16, 32, 32, 64, 64, 16, 8, 4, 4, 2, 2, 1, 1, False)),      # dgrad -> Cn=16: 256x64 tile, phases
    ("narrow_many_rows_T", Geom(16, 32, 32, 64, 64, 12, 8, 4, 4, 2, 2, 1, 1, True)),       # fwd -> Cn=8: 256x64 tile, phases
    ("narrow_1x1", Geom(5, 64, 64, 64, 64, 64, 64, 1, 1, 1, 1, 0, 0, False)),              # 20480 rows, 256x64 tile
    ("wide_many_rows", Geom(6, 32, 32, 64, 64, 32, 192, 4, 4, 2, 2, 1, 1, False)),         # 128x128 + 64-wide remainder rule
], ids=["narrow_C", "narrow_T", "narrow_1x1", "wide"])
def test_conv_many_rows_tile_configs(name, g):
    """Shapes with >= 16384 output rows: the persistent 256x64 / 128x128 8-wave tiles, in both weight
    orientations and with every fused epilogue (the small-batch cases above only reach the 64x64 tile)."""
    test_conv_family(name, g)


PLAN_GEOMS = [
    ("enc_192to256", Geom(5, 4, 4, 8, 8, 192, 256, 4, 4, 2, 2, 1, 1, False)),
    ("dec_128to64", Geom(3, 8, 8, 16, 16, 128, 64, 4, 4, 2, 2, 1, 1, True)),
    ("linear_320to128", Geom(7, 1, 1, 1, 1, 320, 128, 1, 1, 1, 1, 0, 0, False)),
    ("text_convT1d", Geom(3, 1, 16, 1, 32, 640, 512, 1, 4, 1, 2, 0, 1, True)),
    ("ragged_c20", Geom(9, 3, 5, 6, 10, 36, 20, 4, 4, 2, 2, 1, 1, False)),
    ("many_rows", Geom(3, 32, 32, 64, 64, 32, 192, 4, 4, 2, 2, 1, 1, False)),
]


@pytest.mark.parametrize("name,g", PLAN_GEOMS, ids=[n for n, _ in PLAN_GEOMS])
def test_conv_every_launch_plan(name, g):
    """mopoe_conv_plan: every tile x split the autotuner may pick computes the same result (fused epilogues on)."""
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(g.in_shape, generator=gen)
    wp = torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)
    bias = 0.1 * torch.randn(g.Cout, generator=gen)
    dy = torch.randn(g.out_shape, generator=gen)
    rows_in, rows_out = x.numel() // g.Cin, math.prod(g.out_shape[:3])
    bn = make_bn(g.Cin, rows_in, 1, gen, x)
    cmask = Mask((torch.rand(g.N, g.Cout, generator=gen) < 0.5).float() * 2, 1, rows_out // g.N)
    st_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
    y_ref = TB.conv_fwd(x, wp, g, bn_in=bn, bias=bias, mask=cmask, out_stats=st_ref)
    s_ref = torch.zeros(2, g.Cin, dtype=torch.float64)
    dx_ref = TB.conv_dgrad(dy, wp, g, relu_bn=bn, xin=x, bwd_sums=s_ref)
    dw_ref = TB.conv_wgrad(x, dy, g, bn_in=bn)
    xd, wd, dyd, bnd = x.to(DEV), wp.to(DEV), dy.to(DEV), to_dev(bn)
    # residual mix in conv2's epilogue (mopoe_conv_fwd_mix): vector-path shapes only
    mixable = ops.conv_mix_supported(xd, g)
    assert mixable == (g.Cin % 4 == 0 and g.Cout % 4 == 0)
    if mixable:
        sres = torch.randn(g.out_shape, generator=gen)
        bns = make_bn(g.Cout, rows_out, 1, gen, sres)
        stm_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
        ym_ref = TB.conv_fwd(x, wp, g, bn_in=bn, bias=bias, mask=cmask, out_stats=stm_ref, mix=(sres, bns))
        sresd, bnsd = sres.to(DEV), to_dev(bns)
    for tile in range(12):
        for split in (1, 2, 5, 16):
            with ops.force_plan(tile, split):
                st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
                y = ops.conv_fwd(xd, wd, g, bn_in=bnd, bias=bias.to(DEV), mask=to_dev(cmask), out_stats=st)
                if mixable:
                    stm = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
                    ym = ops.conv_fwd(xd, wd, g, bn_in=bnd, bias=bias.to(DEV), mask=to_dev(cmask), out_stats=stm,
                                      mix=(sresd, bnsd))
                    check(f"plan/{name}/t{tile}s{split}/fwd_mix", ym, ym_ref)
                    check(f"plan/{name}/t{tile}s{split}/fwd_mix_stats", stm, stm_ref, rtol=1e-4, atol_rel=1e-4)
                s = torch.zeros(2, g.Cin, dtype=torch.float64, device=DEV)
                dx = ops.conv_dgrad(dyd, wd, g, relu_bn=bnd, xin=xd, bwd_sums=s)
            tag = f"plan/{name}/t{tile}s{split}"
            check(f"{tag}/fwd", y, y_ref)
            check(f"{tag}/fwd_stats", st, st_ref, rtol=1e-4, atol_rel=1e-4)
            check(f"{tag}/dgrad", dx, dx_ref)
            check(f"{tag}/dgrad_sums", s, s_ref, rtol=2e-4, atol_rel=2e-4)
    dwp_ref = TB.conv_wgrad(x, dy, g)
    for tile in (0, 2, 5, 6):      # 5 / 6: the 128 / 64 tiles on LDS-DMA (vector path: channel counts % 4 == 0)
        if tile >= 5 and (g.Cin % 4 or g.Cout % 4):
            continue
        if tile == 5 and min(g.Cin, g.Cout) <= 64:     # the 128 tile on a narrow layer: refused (the tuner never offers it)
            with ops.force_plan(5, 1), pytest.raises(ops.MopoeHipError):
                ops.conv_wgrad(xd, dyd, g)
            continue
        for split in (1, 3, 64):
            with ops.force_plan(tile, split):
                dw = ops.conv_wgrad(xd, dyd, g, bn_in=bnd)
                dwp = ops.conv_wgrad(xd, dyd, g)
            check(f"plan/{name}/wgrad_t{tile}s{split}", dw, dw_ref, rtol=5e-4, atol_rel=5e-4)
            check(f"plan/{name}/wgrad_plain_t{tile}s{split}", dwp, dwp_ref, rtol=5e-4, atol_rel=5e-4)
    for tile in (7, 8):           # fp32 wgrad tiles 5 / 6 with the products on the bf16 matrix pipe: plain operand only
        if g.Cin % 4 or g.Cout % 4:
            continue
        if tile == 7 and min(g.Cin, g.Cout) <= 64:
            with ops.force_plan(7, 1), pytest.raises(ops.MopoeHipError):
                ops.conv_wgrad(xd, dyd, g)
            continue
        for split in (1, 3, 64):
            with ops.force_plan(tile, split):
                dwp = ops.conv_wgrad(xd, dyd, g)
            check(f"plan/{name}/wgrad_plain_t{tile}s{split}", dwp, dwp_ref, rtol=5e-4, atol_rel=5e-4)
        with ops.force_plan(tile, 1), pytest.raises(ops.MopoeHipError):
            ops.conv_wgrad(xd, dyd, g, bn_in=bnd)
    with pytest.raises(ops.MopoeHipError):
        with ops.force_plan(20, 1):
            ops.conv_fwd(xd, wd, g)


GLDS_GEOMS32 = [PLAN_GEOMS[i] for i in (0, 1, 2, 3, 5)] + [
    ("enc_64to128_b9_ragged", Geom(9, 8, 8, 16, 16, 64, 128, 4, 4, 2, 2, 1, 1, False)),
    ("enc_1x1_128", Geom(2, 16, 16, 16, 16, 128, 128, 1, 1, 1, 1, 0, 0, False)),
    ("enc_k4s4p1_16to4", Geom(2, 4, 4, 16, 16, 256, 320, 4, 4, 4, 4, 1, 1, False)),
    ("odd_grid_T_96to32", Geom(2, 5, 6, 10, 12, 96, 32, 4, 4, 2, 2, 1, 1, True)),
]


@pytest.mark.parametrize("name,g", GLDS_GEOMS32, ids=[n for n, _ in GLDS_GEOMS32])
def test_conv_lds_dma_tiles(name, g):
    """the fp32 LDS-DMA tiles (csrc/conv_gemm_glds.inc: 12 = 128x128 / 2 buffers, 13 = 128x64 / 3, 14 = 64x64 / 4,
    15 = 256x128 / 2) with and without a split reduction: forward with a plain operand (projection shortcut: bias +
    statistics; element mask), forward with BN -> ReLU on load and the residual mix (tiles 12, 14, 15), every input gradient"""
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(g.in_shape, generator=gen)
    wp = torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)
    bias = 0.1 * torch.randn(g.Cout, generator=gen)
    dy = torch.randn(g.out_shape, generator=gen)
    rows_in, rows_out = x.numel() // g.Cin, math.prod(g.out_shape[:3])
    bn = make_bn(g.Cin, rows_in, 1, gen, x)
    cmask = Mask((torch.rand(g.N, g.Cout, generator=gen) < 0.5).float() * 2, 1, rows_out // g.N)
    emask = Mask((torch.rand(g.out_shape, generator=gen) < 0.5).float() * 2, 2, rows_out // g.N)
    sres = torch.randn(g.out_shape, generator=gen)
    bns = make_bn(g.Cout, rows_out, 1, gen, sres)
    xd, wd, dyd, bnd, bd = x.to(DEV), wp.to(DEV), dy.to(DEV), to_dev(bn), bias.to(DEV)
    st_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
    y_short = TB.conv_fwd(x, wp, g, bias=bias, out_stats=st_ref)
    y_emask = TB.conv_fwd(x, wp, g, bias=bias, mask=emask)
    stx_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
    y_x = TB.conv_fwd(x, wp, g, bn_in=bn, bias=bias, mask=cmask, out_stats=stx_ref)
    stm_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
    y_mix = TB.conv_fwd(x, wp, g, bn_in=bn, bias=bias, mask=cmask, out_stats=stm_ref, mix=(sres, bns))
    stp_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
    y_mixp = TB.conv_fwd(x, wp, g, bias=bias, mask=cmask, out_stats=stp_ref, mix=(sres, bns))     # residual mix, plain operand
    s_ref = torch.zeros(2, g.Cin, dtype=torch.float64)
    dx_ref = TB.conv_dgrad(dy, wp, g, relu_bn=bn, xin=x, bwd_sums=s_ref)
    dx_plain = TB.conv_dgrad(dy, wp, g)
    for tile in (12, 13, 14, 15, 16, 17, 18, 19):     # 16..19: tiles 12..15 with the fp32 products on the bf16 matrix pipe
        for split in (1, 3):
            tag = f"glds/{name}/t{tile}s{split}"
            with ops.force_plan(tile, split):
                if g.Cin % 32 == 0:
                    st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
                    check(f"{tag}/fwd_shortcut", ops.conv_fwd(xd, wd, g, bias=bd, out_stats=st), y_short)
                    check(f"{tag}/fwd_shortcut_stats", st, st_ref, rtol=1e-4, atol_rel=1e-4)
                    check(f"{tag}/fwd_emask", ops.conv_fwd(xd, wd, g, bias=bd, mask=to_dev(emask)), y_emask)
                    st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
                    check(f"{tag}/fwd_mix_plain", ops.conv_fwd(xd, wd, g, bias=bd, mask=to_dev(cmask), out_stats=st,
                                                               mix=(sres.to(DEV), to_dev(bns))), y_mixp)
                    check(f"{tag}/fwd_mix_plain_stats", st, stp_ref, rtol=1e-4, atol_rel=1e-4)
                    if tile not in (13, 16, 17, 18, 19):
                        st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
                        check(f"{tag}/fwd_bn", ops.conv_fwd(xd, wd, g, bn_in=bnd, bias=bd, mask=to_dev(cmask), out_stats=st), y_x)
                        check(f"{tag}/fwd_bn_stats", st, stx_ref, rtol=1e-4, atol_rel=1e-4)
                        st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
                        check(f"{tag}/fwd_mix", ops.conv_fwd(xd, wd, g, bn_in=bnd, bias=bd, mask=to_dev(cmask), out_stats=st,
                                                             mix=(sres.to(DEV), to_dev(bns))), y_mix)
                        check(f"{tag}/fwd_mix_stats", st, stm_ref, rtol=1e-4, atol_rel=1e-4)
                    else:
                        with pytest.raises(ops.MopoeHipError):
                            ops.conv_fwd(xd, wd, g, bn_in=bnd)
                if g.Cout % 32 == 0:
                    s = torch.zeros(2, g.Cin, dtype=torch.float64, device=DEV)
                    check(f"{tag}/dgrad_relubn", ops.conv_dgrad(dyd, wd, g, relu_bn=bnd, xin=xd, bwd_sums=s), dx_ref)
                    check(f"{tag}/dgrad_sums", s, s_ref, rtol=2e-4, atol_rel=2e-4)
                    check(f"{tag}/dgrad", ops.conv_dgrad(dyd, wd, g), dx_plain)


def _err64(y, ref64):
    d = y.double() - ref64
    return float(d.norm() / ref64.norm()), float(d.abs().max() / ref64.abs().max())


@pytest.mark.parametrize("name,g", [
    ("rb1_C64to128", Geom(4, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False)),
    ("dec_T128to64", Geom(4, 16, 16, 32, 32, 128, 64, 4, 4, 2, 2, 1, 1, True)),
    ("text_T512to512_k1x4", Geom(16, 1, 32, 1, 64, 512, 512, 1, 4, 1, 2, 0, 1, True)),
    ("fc_1x1_320", Geom(64, 1, 1, 1, 1, 320, 320, 1, 1, 1, 1, 0, 0, False)),
], ids=lambda v: v if isinstance(v, str) else "")
def test_f32_products_on_the_bf16_pipe(name, g):
    """plan tiles 16..19 / wgrad tiles 7, 8 (csrc/conv_gemm_glds.inc, EMU): every fp32 operand value split exactly into three
    bf16 parts, six of the nine partial products accumulated in fp32 by the bf16 MFMA.  The claim tested: the result is an
    fp32 result -- its error against an fp64 product of the SAME fp32 operands is no larger than that of the fp32-MFMA tiles
    (v_mfma_f32_32x32x2_f32, tiles 12..15 / 5, 6), and a few 1e-7 relative.  Inputs with a wide dynamic range
    (random signs, magnitudes over 2^+-12, a block of exact zeros) so that no part of the split is trivially empty."""
    gen = torch.Generator().manual_seed(29)
    def wide(shape):
        t = torch.randn(shape, generator=gen) * torch.exp2(torch.randint(-12, 13, shape, generator=gen).float())
        t.view(-1)[: t.numel() // 7] = 0.0
        return t
    for dist in ("normal", "wide"):
        if dist == "normal":
            x, dy = torch.randn(g.in_shape, generator=gen), torch.randn(g.out_shape, generator=gen)
            wp = torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)
        else:
            x, dy, wp = wide(g.in_shape), wide(g.out_shape), wide((g.taps, g.Cin, g.Cout))
        xd, wd, dyd = x.to(DEV), wp.to(DEV), dy.to(DEV)
        y64 = TB.conv_fwd(xd.double(), wd.double(), g)
        dx64 = TB.conv_dgrad(dyd.double(), wd.double(), g)
        dw64 = TB.conv_wgrad(xd.double(), dyd.double(), g)
        err = {}
        for tile in (12, 13, 14, 15, 16, 17, 18, 19):
            with ops.force_plan(tile, 1):
                err[("fwd", tile)] = _err64(ops.conv_fwd(xd, wd, g), y64)
                err[("dgrad", tile)] = _err64(ops.conv_dgrad(dyd, wd, g), dx64)
        wtiles = (6, 8) if min(g.Cin, g.Cout) <= 64 else (5, 6, 7, 8)
        for tile in wtiles:
            with ops.force_plan(tile, 1):
                err[("wgrad", tile)] = _err64(ops.conv_wgrad(xd, dyd, g), dw64)
        for (op, tile), (l2, mx) in sorted(err.items()):
            _log(f"emu/{name}/{dist}/{op}/t{tile}: relL2 vs fp64 {l2:.3e}  max {mx:.3e}")
        for op, pairs in (("fwd", ((16, 12), (17, 13), (18, 14), (19, 15))), ("dgrad", ((16, 12), (17, 13), (18, 14), (19, 15))),
                          ("wgrad", tuple((a, b) for a, b in ((7, 5), (8, 6)) if a in wtiles))):
            for emu, native in pairs:
                l2e, mxe = err[(op, emu)]
                l2n, mxn = err[(op, native)]
                assert l2e <= 1.05 * l2n + 1e-9, (name, dist, op, emu, l2e, l2n)
                assert l2e < 2e-6 and mxe < 1e-5, (name, dist, op, emu, l2e, mxe)


PARITY_GEOMS32 = [
    ("enc_64to128_b3", Geom(3, 16, 16, 32, 32, 64, 128, 4, 4, 2, 2, 1, 1, False)),      # conv (rb1's shape): one G tile, two S tiles
    ("enc_128to192_b5", Geom(5, 8, 8, 16, 16, 128, 192, 4, 4, 2, 2, 1, 1, False)),      # two G tiles, three S tiles
    ("enc_64to64_b2", Geom(2, 8, 16, 16, 32, 64, 64, 4, 4, 2, 2, 1, 1, False)),         # non-square map
    ("enc_72to136_b2", Geom(2, 8, 8, 16, 16, 72, 136, 4, 4, 2, 2, 1, 1, False)),        # partial channel tiles on both sides
    ("enc_64to256_b2", Geom(2, 8, 8, 16, 16, 64, 256, 4, 4, 2, 2, 1, 1, False)),        # two S tiles of 128
    ("dec_128to64_b5", Geom(5, 8, 8, 16, 16, 128, 64, 4, 4, 2, 2, 1, 1, True)),         # transposed: G = dy (64), S = x (128)
    ("dec_64to64_b3", Geom(3, 16, 16, 32, 32, 64, 64, 4, 4, 2, 2, 1, 1, True)),         # transposed
    ("dec_192to128_b2", Geom(2, 8, 8, 16, 16, 192, 128, 4, 4, 2, 2, 1, 1, True)),       # transposed, two G tiles
]


@pytest.mark.parametrize("name,g", PARITY_GEOMS32, ids=[n for n, _ in PARITY_GEOMS32])
def test_wgrad_four_taps_per_block_f32(name, g):
    """fp32 wgrad tiles 9 / 10 (csrc/conv_gemm_glds_parity.inc): one parity class of a k4 s2 p1 kernel -- four taps -- per block, the
    9 x 9 big-grid pixels of an 8 x 8 tile of small-grid pixels staged once for all of them, the fp32 products on the bf16
    matrix pipe: every tap of every class against the emulation AND against fp64 (no further from it than the fp32-MFMA
    tile 2), with and without a split of the pixel tiles; refused with BN on load, for other kernel shapes and for maps that
    are not whole tiles"""
    gen = torch.Generator().manual_seed(41 + len(name))
    x = torch.randn(g.in_shape, generator=gen)
    dy = torch.randn(g.out_shape, generator=gen)
    ref = TB.conv_wgrad(x, dy, g)
    xd, dyd = x.to(DEV), dy.to(DEV)
    ref64 = TB.conv_wgrad(xd.double(), dyd.double(), g)
    with ops.force_plan(2, 1):
        e_native = _err64(ops.conv_wgrad(xd, dyd, g), ref64)
    csm = g.Cin if g.transposed else g.Cout
    for tile in (9, 10):          # 10: S tile of 128 channels (a wave owns all four taps)
        if tile == 10 and csm % 128:
            with ops.force_plan(10, 1), pytest.raises(ops.MopoeHipError):
                ops.conv_wgrad(xd, dyd, g)
            continue
        for split in (1, 2, 5):
            with ops.force_plan(tile, split):
                dw = ops.conv_wgrad(xd, dyd, g)
            check(f"{name}/wgrad_t{tile}s{split}", dw, ref, rtol=3e-4, atol_rel=3e-4)
            e = _err64(dw, ref64)
            _log(f"parity32/{name}/t{tile}s{split}: relL2 vs fp64 {e[0]:.3e} (tile 2: {e_native[0]:.3e})")
            assert e[0] <= 1.05 * e_native[0] + 1e-9, (name, tile, split, e, e_native)
    with ops.force_plan(9, 1):
        bn = make_bn(g.Cin, x.numel() // g.Cin, 1, gen, x)
        with pytest.raises(ops.MopoeHipError):
            ops.conv_wgrad(xd, dyd, g, bn_in=to_dev(bn))
        for bad in (Geom(2, 6, 5, 12, 10, 64, 64, 4, 4, 2, 2, 1, 1, False),           # not whole 8 x 8 tiles
                    Geom(2, 1, 16, 1, 32, 64, 64, 1, 4, 1, 2, 0, 1, True),            # 1-D
                    Geom(2, 4, 4, 16, 16, 64, 64, 4, 4, 4, 4, 1, 1, False),           # stride 4
                    Geom(2, 8, 8, 16, 16, 66, 64, 4, 4, 2, 2, 1, 1, False)):          # channel count % 4 != 0 (scalar path)
            with pytest.raises(ops.MopoeHipError):
                ops.conv_wgrad(torch.zeros(bad.in_shape, device=DEV), torch.zeros(bad.out_shape, device=DEV), bad)


def test_conv_large_rows_splitk_and_big_tiles():
    """shapes of the C2 config's heaviest layers at reduced batch: exercises the 128x128 tiles and the
    split pixel reduction of wgrad."""
    gen = torch.Generator().manual_seed(5)
    for name, g in (("rb1", Geom(8, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False)),
                    ("g3", Geom(8, 16, 16, 32, 32, 128, 64, 4, 4, 2, 2, 1, 1, True))):
        x = torch.randn(g.in_shape, generator=gen)
        wp = torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)
        dy = torch.randn(g.out_shape, generator=gen)
        check(f"big/{name}/fwd", ops.conv_fwd(x.to(DEV), wp.to(DEV), g), TB.conv_fwd(x, wp, g))
        check(f"big/{name}/dgrad", ops.conv_dgrad(dy.to(DEV), wp.to(DEV), g), TB.conv_dgrad(dy, wp, g))
        check(f"big/{name}/wgrad", ops.conv_wgrad(x.to(DEV), dy.to(DEV), g), TB.conv_wgrad(x, dy, g), rtol=5e-4,
              atol_rel=5e-4)


@pytest.mark.parametrize("rows,c", [(4096, 128), (1000, 192), (37, 20), (5, 640), (3000, 1), (64, 3517 // 7 * 4)])
def test_block_glue(rows, c):
    gen = torch.Generator().manual_seed(rows + c)
    s, m, g = (torch.randn(rows, c, generator=gen) for _ in range(3))
    n = 1 if rows < 8 else 4
    rps = rows // n
    rows = n * rps
    s, m, g = s[:rows].contiguous(), m[:rows].contiguous(), g[:rows].contiguous()
    for mode in (1, 2):
        bn = make_bn(c, rows, mode, gen, s if mode == 1 else None)
        st_ref = torch.zeros(2, c, dtype=torch.float64)
        out_ref = TB.block_out_fwd(s, m, bn, out_stats=st_ref)
        st = torch.zeros(2, c, dtype=torch.float64, device=DEV)
        out = ops.block_out_fwd(s.to(DEV), m.to(DEV), to_dev(bn), out_stats=st)
        check(f"block_out_fwd[{rows}x{c}]bn{mode}", out, out_ref)
        check(f"bn_relu_apply[{rows}x{c}]bn{mode}", ops.bn_relu_apply(s.to(DEV), to_dev(bn)), TB.bn_relu_apply(s, bn))
        check(f"block_out_fwd[{rows}x{c}]bn{mode}/stats", st, st_ref, rtol=1e-4, atol_rel=1e-4)
        sums_ref = TB.bn_bwd_reduce(g, s, bn)
        sums = ops.bn_bwd_reduce(g.to(DEV), s.to(DEV), to_dev(bn))
        check(f"bn_bwd_reduce[{rows}x{c}]bn{mode}", sums, sums_ref, rtol=2e-4, atol_rel=2e-4)
        for mask in (None, Mask((torch.rand(n, c, generator=gen) < 0.5).float() * 2, 1, rps),
                     Mask((torch.rand(rows, c, generator=gen) < 0.5).float() * 2, 2, rps)):
            tag = f"[{rows}x{c}]bn{mode}mask{0 if mask is None else mask.kind}"
            ref = TB.block_out_bwd(g, s, bn, sums_ref, mask, want_colsum_dm=True, want_colsum_ds=True)
            got = ops.block_out_bwd(g.to(DEV), s.to(DEV), to_dev(bn), sums_ref.to(DEV), to_dev(mask),
                                    want_colsum_dm=True, want_colsum_ds=True)
            for nm, a, b in zip(("dm", "ds", "dgamma", "dbeta", "colsum_dm"), got, ref):
                check(f"block_out_bwd{tag}/{nm}", a, b)
            # colsum of ds is analytically ~0 in train mode: compare on the scale of ds's column L1 norms
            scale = ref[1].abs().sum(0).max().item()
            assert (got[5].cpu() - ref[5]).abs().max().item() <= 2e-5 * scale + 1e-6
            ref = TB.bn_bwd_apply(g, s, bn, sums_ref, mask=mask, add=m, want_colsum=True)
            got = ops.bn_bwd_apply(g.to(DEV), s.to(DEV), to_dev(bn), sums_ref.to(DEV), mask=to_dev(mask),
                                   add=m.to(DEV), want_colsum=True)
            for nm, a, b in zip(("dx", "dgamma", "dbeta"), got, ref):
                check(f"bn_bwd_apply{tag}/{nm}", a, b)
            scale = ref[0].abs().sum(0).max().item()
            assert (got[3].cpu() - ref[3]).abs().max().item() <= 2e-5 * scale + 1e-6
    check(f"colsum[{rows}x{c}]", ops.colsum(s.to(DEV)), TB.colsum(s), rtol=1e-4, atol_rel=1e-4 * math.sqrt(rows))


def test_bn_running_update():
    gen = torch.Generator().manual_seed(3)
    entries_ref, entries = [], []
    for c, rows in ((64, 1000), (20, 7), (640, 64)):
        x = torch.randn(rows, c, generator=gen).double() * 2 + 1
        sums = torch.stack([x.sum(0), (x * x).sum(0)])
        rm, rv = torch.randn(c, generator=gen), torch.rand(c, generator=gen) + 0.5
        entries_ref.append((sums, rm.clone(), rv.clone(), rows))
        entries.append((sums.to(DEV), rm.to(DEV), rv.to(DEV), rows))
    TB.bn_running_update(entries_ref)
    keep = ops.bn_running_update(entries)
    torch.cuda.synchronize()
    for (_, rm_r, rv_r, _), (_, rm, rv, _) in zip(entries_ref, entries):
        check("running_mean", rm, rm_r, 1e-5, 1e-5)
        check("running_var", rv, rv_r, 1e-5, 1e-5)


def test_adam_step_against_fused_adam():
    """csrc/adam.hip against PyTorch's fused capturable Adam (the kernel it replaces) and the CPU restatement: many
    tensors (> one launch's 64 records), sizes around the 4096-element chunk and the 4-element vector, a tensor without
    gradient, a misaligned view, the bf16 copy, a learning-rate change on the device"""
    gen = torch.Generator().manual_seed(11)
    sizes = [1, 3, 4, 5, 4095, 4096, 4097, 8192 + 2, 100_000, 1_638_400] + [17 + 13 * i for i in range(70)]
    big = torch.randn(sum(sizes) + 8, generator=gen)
    host, off = [], 0
    for i, n in enumerate(sizes):
        o = off + (1 if i == 7 else 0)                 # tensor 7 starts 4 bytes off a 16-byte boundary
        host.append(big[o:o + n].clone())
        off += n
    p_hip = [h.to(DEV) for h in host]
    stash = torch.zeros(sizes[7] + 4, device=DEV)
    stash[1:1 + sizes[7]] = p_hip[7]
    p_hip[7] = stash[1:1 + sizes[7]]
    assert p_hip[7].data_ptr() % 16 == 4
    p_ref = [torch.nn.Parameter(h.to(DEV)) for h in host]
    p_cpu = [h.clone() for h in host]
    m, v = [torch.zeros_like(p) for p in p_hip], [torch.zeros_like(p) for p in p_hip]
    m_c, v_c = [torch.zeros_like(p) for p in p_cpu], [torch.zeros_like(p) for p in p_cpu]
    lowp = [torch.zeros(p.numel(), dtype=torch.bfloat16, device=DEV) if i in (5, 9, 12) else None for i, p in enumerate(p_hip)]
    lr = torch.tensor(2e-3, device=DEV)
    lr_ref = torch.tensor(2e-3, device=DEV)
    step, coef = torch.zeros((), device=DEV), torch.zeros(2, device=DEV)
    step_c = torch.zeros(())
    opt = torch.optim.Adam(p_ref, lr=lr_ref, betas=(0.9, 0.999), fused=True, capturable=True)
    for it in range(4):
        grads = [torch.randn(n, generator=gen) * (0.01 if i % 3 else 3.0) for i, n in enumerate(sizes)]
        grads[2] = None
        for p, g in zip(p_ref, grads):
            p.grad = None if g is None else g.to(DEV)
        opt.step()
        ops.adam_step(p_hip, [None if g is None else g.to(DEV) for g in grads], m, v, step, lr, 0.9, 0.999, 1e-8, coef, lowp=lowp)
        TB.adam_step(p_cpu, grads, m_c, v_c, step_c, 2e-3 if it < 2 else 5e-4, 0.9, 0.999, 1e-8, None)
        if it == 1:
            lr.fill_(5e-4)
            lr_ref.fill_(5e-4)
    torch.cuda.synchronize()
    assert float(step) == 4
    for i, (a, b, c) in enumerate(zip(p_hip, p_ref, p_cpu)):
        check(f"adam[{i}] vs fused", a, b.detach().cpu(), 2e-6, 2e-7)
        check(f"adam[{i}] vs restatement", a, c, 2e-6, 2e-7)
        if lowp[i] is not None:
            assert torch.equal(lowp[i].cpu(), a.cpu().to(torch.bfloat16)), i
    assert torch.equal(p_hip[2].cpu(), host[2]) and float(m[2].abs().sum()) == 0.0
    check("exp_avg", m[9], opt.state[p_ref[9]]["exp_avg"].cpu(), 2e-6, 1e-6)
    # (PyTorch's GPU kernel forms 1 - beta2 in float: 0.00100005; the reference's CPU Adam and this kernel use the double)
    check("exp_avg_sq", v[9], opt.state[p_ref[9]]["exp_avg_sq"].cpu(), 1e-4, 1e-6)
    check("exp_avg vs restatement", m[9], m_c[9], 2e-6, 2e-7)
    check("exp_avg_sq vs restatement", v[9], v_c[9], 2e-6, 2e-7)


@pytest.mark.parametrize("present", [(1, 1, 1), (1, 0, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 0)])
@pytest.mark.parametrize("b,d", [(64, 128), (7, 8), (65, 64)])
def test_latent(present, b, d):
    from mimic_amd.mmvae import kl_weights, mixture_row_starts
    gen = torch.Generator().manual_seed(b * 100 + d + sum(present))
    mu = [torch.randn(b, d, generator=gen) if p else None for p in present]
    lv = [0.5 * torch.randn(b, d, generator=gen) if p else None for p in present]
    eps = torch.randn(b, d, generator=gen)
    k = len(TB._active_subsets(mu))
    rs, w, norm = mixture_row_starts(b, k), kl_weights(k), float(b + 3)
    ref = TB.latent_fwd(mu, lv, eps, rs, w, norm)
    got = ops.latent_fwd([to_dev(t) for t in mu], [to_dev(t) for t in lv], eps.to(DEV), rs, w, norm)
    for nm, a, r in zip(("mus", "lvs", "jm", "jl", "z", "klds", "jd"), got, ref):
        check(f"latent_fwd{present}[{b}x{d}]/{nm}", a, r, 1e-4, 1e-5)
    # second call must see a clean workspace
    got2 = ops.latent_fwd([to_dev(t) for t in mu], [to_dev(t) for t in lv], eps.to(DEV), rs, w, norm)
    check("latent_fwd/rerun/klds", got2[5], ref[5], 1e-4, 1e-5)
    gs = [torch.randn(t.shape, generator=gen) for t in ref]
    for combo in ("all", "train"):  # 'train': only z and joint_divergence carry gradient
        g_use = gs if combo == "all" else [None, None, None, None, gs[4], None, gs[6]]
        dmu_r, dlv_r = TB.latent_bwd(mu, lv, eps, rs, w, norm, *g_use)
        dmu, dlv = ops.latent_bwd([to_dev(t) for t in mu], [to_dev(t) for t in lv], eps.to(DEV), rs, w, norm,
                                  *[to_dev(t) for t in g_use])
        for s in range(3):
            if present[s]:
                check(f"latent_bwd{present}[{b}x{d}]{combo}/dmu{s}", dmu[s], dmu_r[s], 2e-4, 2e-5)
                check(f"latent_bwd{present}[{b}x{d}]{combo}/dlv{s}", dlv[s], dlv_r[s], 2e-4, 2e-5)


def test_likelihoods_and_embedding():
    gen = torch.Generator().manual_seed(11)
    for n in (64 * 128 * 128, 4099, 17):
        xh, x = torch.randn(n, generator=gen), torch.rand(n, generator=gen)
        ref = TB.laplace_nll_fwd(xh, x, 0.75, 64.0)
        got = ops.laplace_nll_fwd(xh.to(DEV), x.to(DEV), 0.75, 64.0)
        check(f"laplace_fwd[{n}]", got, ref, 2e-6, 2e-6)
        got = ops.laplace_nll_fwd(xh.to(DEV), x.to(DEV), 0.75, 64.0)  # workspace left clean
        check(f"laplace_fwd[{n}]/rerun", got, ref, 2e-6, 2e-6)
        g = torch.tensor([0.33])
        check(f"laplace_bwd[{n}]", ops.laplace_nll_bwd(xh.to(DEV), x.to(DEV), g.to(DEV), 0.75, 64.0),
              TB.laplace_nll_bwd(xh, x, g, 0.75, 64.0), 1e-6, 1e-6)
    for rows, v in ((8 * 128, 3517), (33, 50), (5, 1024), (3, 7000)):
        x = 3 * torch.randn(rows, v, generator=gen)
        y_ref = TB.logsoftmax_fwd(x)
        y = ops.logsoftmax_fwd(x.to(DEV))
        check(f"logsoftmax_fwd[{rows}x{v}]", y, y_ref, 1e-5, 1e-6)
        dy = torch.randn(rows, v, generator=gen)
        check(f"logsoftmax_bwd[{rows}x{v}]", ops.logsoftmax_bwd(dy.to(DEV), y), TB.logsoftmax_bwd(dy, y_ref), 1e-4, 1e-5)
        ids = torch.randint(0, v, (rows,), generator=gen).float()
        check(f"token_nll_fwd[{rows}x{v}]", ops.token_nll_fwd(y, ids.to(DEV), 8.0), TB.token_nll_fwd(y_ref, ids, 8.0),
              2e-6, 2e-6)
        g = torch.tensor([0.7])
        check(f"token_nll_bwd[{rows}x{v}]", ops.token_nll_bwd(ids.to(DEV), g.to(DEV), (rows, v), 8.0),
              TB.token_nll_bwd(ids, g, (rows, v), 8.0), 1e-6, 1e-6)
    for v, d, shape in ((3517, 128, (8, 128)), (50, 4, (4, 128))):
        table = torch.randn(v, d, generator=gen)
        ids = torch.randint(0, v, shape, generator=gen).float()
        ids[:, :3] = 0
        check(f"embedding_fwd[{v}x{d}]", ops.embedding_fwd(ids.to(DEV), table.to(DEV)), TB.embedding_fwd(ids, table), 0, 0)
        gout = torch.randn(*shape, d, generator=gen)
        check(f"embedding_bwd[{v}x{d}]", ops.embedding_bwd(ids.to(DEV), gout.to(DEV), v, 0),
              TB.embedding_bwd(ids, gout, v, 0), 1e-4, 1e-5)


@pytest.mark.parametrize("rows,tb,per_row", [(12, 4, 64 * 64), (6, 6, 4099), (10, 5, 7), (384, 64, 128 * 128)])
def test_laplace_logprob_rows(rows, tb, per_row):
    gen = torch.Generator().manual_seed(rows + per_row)
    xh = torch.rand(rows, per_row, generator=gen)
    x = torch.rand(tb, per_row, generator=gen)
    ref = TB.laplace_logprob_rows(xh.double(), x.double(), 0.75)
    got = ops.laplace_logprob_rows(xh.to(DEV), x.to(DEV), 0.75)
    check("laplace_logprob_rows", got, ref, rtol=2e-6, atol_rel=2e-6)
    # unaligned views take the scalar path
    got = ops.laplace_logprob_rows(torch.cat([torch.zeros(1), xh.flatten()]).to(DEV)[1:].view(rows, per_row), x.to(DEV), 0.75)
    check("laplace_logprob_rows/unaligned", got, ref, rtol=2e-6, atol_rel=2e-6)


@pytest.mark.parametrize("b,L,F,onehot", [(4, 1024, 71, True), (3, 37, 5, False), (1, 1, 1, False), (16, 1024, 71, False)])
def test_dense_nll_and_rows(b, L, F, onehot):
    """text_encoding='char' likelihood (MimicText.py:37-40): dense [B, L, F] targets, one-hot or not"""
    gen = torch.Generator().manual_seed(b * L + F)
    logp = torch.log_softmax(torch.randn(b, L, F, generator=gen), dim=-1)
    if onehot:
        tgt = torch.nn.functional.one_hot(torch.randint(0, F, (b, L), generator=gen), F).float()
    else:
        tgt = torch.rand(b, L, F, generator=gen)
    ref = TB.dense_nll_fwd(logp.double(), tgt.double(), float(b))
    got = ops.dense_nll_fwd(logp.to(DEV), tgt.to(DEV), float(b))
    check("dense_nll_fwd", got.view(-1), ref.view(-1), rtol=2e-6, atol_rel=2e-6)
    g = torch.tensor(0.37)
    check("dense_nll_bwd", ops.dense_nll_bwd(tgt.to(DEV), g.to(DEV), float(b)), TB.dense_nll_bwd(tgt, g, float(b)), rtol=1e-6, atol_rel=1e-7)
    k = 3
    lp_rep = torch.log_softmax(torch.randn(k * b, L, F, generator=gen), dim=-1)
    check("dense_logprob_rows", ops.dense_logprob_rows(lp_rep.to(DEV), tgt.to(DEV)), TB.dense_logprob_rows(lp_rep.double(), tgt.double()),
          rtol=2e-6, atol_rel=2e-6)


@pytest.mark.parametrize("b,L,V", [(4, 128, 3520), (3, 7, 52), (2, 5, 9), (1, 3, 5000)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_token_softmax_grad(b, L, V, out_dtype):
    """mopoe_token_softmax_grad == token_nll_bwd followed by logsoftmax_bwd (what it replaces), and the emulation"""
    gen = torch.Generator().manual_seed(b * L + V)
    logp = torch.log_softmax(torch.randn(b, L, V, generator=gen), dim=-1)
    ids = torch.randint(0, V, (b, L), generator=gen).float()
    g = torch.tensor([0.73])
    ref = TB.token_softmax_grad(logp.double(), ids, g.double(), float(b)).float()
    got = ops.token_softmax_grad(logp.to(DEV), ids.to(DEV), g.to(DEV), float(b), out_dtype=out_dtype)
    assert got.dtype == out_dtype
    two_step = ops.logsoftmax_bwd(ops.token_nll_bwd(ids.to(DEV), g.to(DEV), (b, L, V), float(b)), logp.to(DEV))
    if out_dtype == torch.float32:
        check("token_softmax_grad", got, ref, rtol=2e-6, atol_rel=2e-6)
        check("token_softmax_grad/vs_two_kernels", got, two_step, rtol=2e-6, atol_rel=2e-6)
    else:
        check("token_softmax_grad/bf16", got.float(), ref.to(torch.bfloat16).float(), rtol=1e-2, atol_rel=1e-4)


@pytest.mark.parametrize("n,rps,drop", [(3, 64, True), (2, 1024, False), (5, 32, True), (1, 4096, True)])
def test_block_front_streaming_kernels_f32(n, rps, drop):
    """csrc/pointwise.hip, fp32 family: the front of a residual block (bn1 -> relu -> conv1 1x1 at 64 channels -> Dropout2d ->
    bn2 -> relu) without ever writing d1 -- statistics pass, a2 pass, fused backward -- against the emulation and against the
    HIP ops they replace (conv_fwd + bn_relu_apply / bn_bwd_apply + conv_dgrad + conv_wgrad)"""
    c, rows = 64, n * rps
    gen = torch.Generator().manual_seed(n * 1000 + rps)
    x = torch.randn(n, rps, 1, c, generator=gen)
    w1 = torch.randn(1, c, c, generator=gen) / 8
    bias = 0.1 * torch.randn(c, generator=gen)
    bn1 = make_bn(c, rows, 1, gen, x)
    mask = Mask((torch.rand(n, c, generator=gen) < 0.5).float() * 2, 1, rps) if drop else None
    g1 = Geom(n, rps, 1, rps, 1, c, c, 1, 1, 1, 1, 0, 0, False)
    xd, wd, bd, bn1d, md = x.to(DEV), w1.to(DEV), bias.to(DEV), to_dev(bn1), to_dev(mask)
    assert ops.block_front_supported(xd, g1, md)
    st_ref, st = torch.zeros(2, c, dtype=torch.float64), torch.zeros(2, c, dtype=torch.float64, device=DEV)
    TB.block_front_stats(x, w1, bias, bn1, mask, st_ref)
    ops.block_front_stats(xd, wd, bd, bn1d, md, st)
    check("front32/stats", st, st_ref, rtol=1e-4, atol_rel=1e-4)
    st_old = torch.zeros(2, c, dtype=torch.float64, device=DEV)
    d1_old = ops.conv_fwd(xd, wd, g1, bn_in=bn1d, bias=bd, mask=md, out_stats=st_old)
    check("front32/stats_vs_conv_fwd", st, st_old, rtol=1e-4, atol_rel=1e-4)
    bn2 = Bn(torch.rand(c, generator=gen) + 0.5, 0.1 * torch.randn(c, generator=gen), 1, st_ref.clone(), rows)
    bn2d = to_dev(bn2)
    a2_ref = TB.block_front_apply(x, w1, bias, bn1, bn2, mask)
    a2 = ops.block_front_apply(xd, wd, bd, bn1d, bn2d, md)
    check("front32/a2", a2, a2_ref)
    check("front32/a2_vs_bn_relu_apply", a2, ops.bn_relu_apply(d1_old, bn2d))
    dh2 = torch.randn(rows, c, generator=gen).view(x.shape) * (a2_ref > 0)
    mean2, rstd2, _, _ = TB.bn_coef(bn2)
    d1_ref = TB._front_d1(x, w1, bias, bn1, mask)[1]
    sums2 = torch.stack([dh2.reshape(rows, c).double().sum(0), (dh2 * ((d1_ref - mean2) * rstd2)).reshape(rows, c).double().sum(0)])
    ref = dict(s1=torch.zeros(2, c, dtype=torch.float64), dw=torch.zeros(1, c, c), db=torch.zeros(c), dg=torch.zeros(c), dbt=torch.zeros(c))
    dh1_ref = TB.block_front_bwd(x, dh2, w1, bias, bn1, bn2, mask, sums2, ref["s1"], ref["dw"], ref["db"], ref["dg"], ref["dbt"])
    s1 = torch.zeros(2, c, dtype=torch.float64, device=DEV)
    dw, db, dg, dbt = (torch.zeros(1, c, c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(c, device=DEV))
    dh1 = ops.block_front_bwd(xd, dh2.to(DEV), wd, bd, bn1d, bn2d, md, sums2.to(DEV), s1, dw, db, dg, dbt)
    check("front32/dh1", dh1, dh1_ref)
    check("front32/sums1", s1, ref["s1"], rtol=5e-4, atol_rel=5e-4)
    check("front32/dw1", dw, ref["dw"], rtol=5e-4, atol_rel=5e-4)
    # (the column sums of a BatchNorm backward are analytically zero: the bound is fp32 rounding of an element times sqrt(rows))
    noise = 2.0 ** -22 * float(dh2.abs().max()) * float((bn2.gamma * rstd2).abs().max()) * rows ** 0.5 * 8 + 1e-5
    assert (db.cpu() - ref["db"]).abs().max().item() <= noise, ((db.cpu() - ref["db"]).abs().max().item(), noise)
    check("front32/dgamma2", dg, ref["dg"], rtol=1e-6, atol_rel=1e-6)
    check("front32/dbeta2", dbt, ref["dbt"], rtol=1e-6, atol_rel=1e-6)
    dc1_old, _, _, _ = ops.bn_bwd_apply(dh2.to(DEV), d1_old, bn2d, sums2.to(DEV), mask=md, want_colsum=True)
    s1_old = torch.zeros(2, c, dtype=torch.float64, device=DEV)
    dh1_old = ops.conv_dgrad(dc1_old, wd, g1, relu_bn=bn1d, xin=xd, bwd_sums=s1_old)
    check("front32/dh1_vs_old_ops", dh1, dh1_old)
    check("front32/sums1_vs_old_ops", s1, s1_old, rtol=5e-4, atol_rel=5e-4)
    check("front32/dw1_vs_old_ops", dw, ops.conv_wgrad(xd, dc1_old, g1, bn_in=bn1d), rtol=5e-4, atol_rel=5e-4)


@pytest.mark.parametrize("b,L,V", [(4, 128, 3520), (3, 7, 64), (2, 5, 8), (1, 3, 5000), (2, 4, 10240)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vocabulary_head_from_logits(b, L, V, dtype):
    """the factored vocabulary head (round 4): log-sum-exp per row, token NLL and the logits' gradient straight from the
    stored logits (fp32 / bf16, padded columns at -1e30) -- against double-precision torch on the SAME stored values, and
    against the dense kernels they replace (logsoftmax_fwd -> token_nll_fwd / token_softmax_grad)"""
    gen = torch.Generator().manual_seed(b * L + V)
    logits = (3.0 * torch.randn(b, L, V, generator=gen)).to(dtype)
    logits[..., V - 3:] = -1e30                       # the padded head's pad columns
    logits[0, 0, : V - 3] += 40.0                     # a row far from the others' range (the online max has to move)
    ids = torch.randint(0, V - 3, (b, L), generator=gen).float()
    g = torch.tensor([0.73])
    x64 = logits.double()
    lse_ref = torch.logsumexp(x64, dim=-1)
    lse = ops.lse_rows(logits.to(DEV))
    check("lse_rows", lse, lse_ref.float(), rtol=2e-6, atol_rel=2e-6)
    nll_ref = TB.token_nll_logits_fwd(x64, lse_ref, ids, float(b))
    nll = ops.token_nll_logits_fwd(logits.to(DEV), lse, ids.to(DEV), float(b))
    check("token_nll_logits_fwd", nll, nll_ref, rtol=5e-6, atol_rel=0)
    grad_ref = TB.token_softmax_grad_logits(x64, lse_ref, ids, g.double(), float(b))
    grad = ops.token_softmax_grad_logits(logits.to(DEV), lse, ids.to(DEV), g.to(DEV), float(b))
    assert grad.dtype == dtype and grad.data_ptr() != logits.data_ptr()
    if dtype == torch.float32:
        check("token_softmax_grad_logits", grad, grad_ref.float(), rtol=1e-5, atol_rel=2e-6)
        if V > 8192:      # (the dense kernels keep a row in registers: V <= 8192; the factored ones walk any V)
            return
        # the dense path on the same logits
        logp = ops.logsoftmax_fwd(logits.to(DEV))
        check("vs logsoftmax+token_nll", nll, ops.token_nll_fwd(logp, ids.to(DEV), float(b)), rtol=5e-6, atol_rel=0)
        check("vs token_softmax_grad", grad, ops.token_softmax_grad(logp, ids.to(DEV), g.to(DEV), float(b)), rtol=1e-5, atol_rel=2e-6)
    else:
        check("token_softmax_grad_logits/bf16", grad.float(), grad_ref.to(torch.bfloat16).float(), rtol=1e-2, atol_rel=1e-4)
    # in place: the gradient may overwrite the logits
    buf = logits.to(DEV).clone()
    same = ops.token_softmax_grad_logits(buf, lse, ids.to(DEV), g.to(DEV), float(b), inplace=True)
    assert same.data_ptr() == buf.data_ptr() and torch.equal(same, grad)


@pytest.mark.parametrize("rows,tb,L,V", [(12, 4, 128, 50), (6, 3, 300, 3517), (5, 5, 1, 9)])
def test_token_logprob_rows(rows, tb, L, V):
    gen = torch.Generator().manual_seed(rows + V)
    logp = torch.log_softmax(torch.randn(rows, L, V, generator=gen), dim=-1)
    ids = torch.randint(0, V, (tb, L), generator=gen).float()
    ref = TB.token_logprob_rows(logp.double(), ids)
    got = ops.token_logprob_rows(logp.to(DEV), ids.to(DEV))
    check("token_logprob_rows", got, ref, rtol=2e-6, atol_rel=2e-6)
