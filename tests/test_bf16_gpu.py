"""bf16 storage family (BASELINE configs #3 and #5) on the GPU, through the C ABI:

  * every bf16 kernel against the plain-PyTorch emulation of the same op evaluated on the CPU from the SAME bf16 inputs
    with the kernels' rounding points (tests/torch_backend.py: operands after BN+ReLU and every stored result rounded to
    bfloat16, sums in fp32).  Tolerance: one bf16 rounding step (2^-7 relative) on bf16 results -- the two sides sum in
    different orders, so a value that lands next to a rounding boundary may round the other way -- plus the fp32
    accumulation noise of the fp32 tests; fp32 results (weight gradients, statistics, fp32 outputs) as in the fp32 tests;
  * the whole model against the CPU oracle run in its bf16 mode (oracle/mopoe_ref.py: the reference arithmetic with the
    same rounding points), and against the oracle's plain fp32 arithmetic at SURVEY 8c's bf16 tolerance (rtol 2e-2);
  * BASELINE config #3 (128 px, class_dim 128, B = 256) and #5 (256 px, class_dim 256, B = 32) at full size.
"""
import math
import os
import zlib

import numpy as np
import pytest
import torch

import mopoe_ref as R
import torch_backend as TB
from model_util import build_exp
from golden_util import load, cfg_from, weights_fingerprint
from g7_util import check_bf16_case, g7_inputs
from mimic_amd import ops, run_epochs as RE
from mimic_amd.ops import Bn, Geom, Mask
from test_hip_ops_gpu import check, make_bn, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
ULP = 2.0 ** -7      # one bf16 rounding step, relative


def _log(msg):
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bf16_parity.log", "a") as f:
        f.write(msg + "\n")


def check16(name, got, ref, atol_rel=2e-4):
    """bf16 result: within one rounding step of the reference (+ fp32 accumulation noise on the tensor's scale)"""
    assert got.dtype == ref.dtype, (name, got.dtype, ref.dtype)
    if got.dtype == BF:
        check(name, got.float(), ref.float(), rtol=1.01 * ULP, atol_rel=atol_rel)
    else:
        check(name, got, ref, rtol=2e-4, atol_rel=2e-4)


# (name, Geom): the geometry families of the four networks at channel counts the bf16 family accepts (K % 32 == 0)
GEOMS16 = [
    ("enc_k4s2p1_64to128", Geom(3, 8, 8, 16, 16, 64, 128, 4, 4, 2, 2, 1, 1, False)),
    ("enc_k4s2p1_192to256_b5", Geom(5, 4, 4, 8, 8, 192, 256, 4, 4, 2, 2, 1, 1, False)),
    ("enc_k4s2p0_4to1", Geom(6, 1, 1, 4, 4, 320, 320, 4, 4, 2, 2, 0, 0, False)),
    ("enc_k4s4p1_16to4", Geom(2, 4, 4, 16, 16, 256, 320, 4, 4, 4, 4, 1, 1, False)),
    ("enc_1x1_128", Geom(2, 16, 16, 16, 16, 128, 128, 1, 1, 1, 1, 0, 0, False)),
    ("linear_320to128", Geom(7, 1, 1, 1, 1, 320, 128, 1, 1, 1, 1, 0, 0, False)),
    ("linear_128to320_b300", Geom(300, 1, 1, 1, 1, 128, 320, 1, 1, 1, 1, 0, 0, False)),
    ("text_conv1d_k4s2p1", Geom(3, 1, 64, 1, 128, 128, 128, 1, 4, 1, 2, 0, 1, False)),
    ("text_conv1d_to1", Geom(5, 1, 1, 1, 2, 512, 640, 1, 4, 1, 2, 0, 1, False)),
    ("dec_k4s2p1_128to64", Geom(3, 8, 8, 16, 16, 128, 64, 4, 4, 2, 2, 1, 1, True)),
    ("dec_k4_from1x1", Geom(5, 1, 1, 4, 4, 320, 256, 4, 4, 4, 4, 0, 0, True)),
    ("dec_1x1", Geom(2, 8, 8, 8, 8, 192, 192, 1, 1, 1, 1, 0, 0, True)),
    ("text_convT1d_k4s2p1", Geom(3, 1, 16, 1, 32, 640, 512, 1, 4, 1, 2, 0, 1, True)),
    ("text_convT1d_from1", Geom(4, 1, 1, 1, 4, 640, 640, 1, 4, 1, 4, 0, 0, True)),
    ("odd_grid_k4s2p1_b3", Geom(3, 6, 5, 12, 10, 64, 96, 4, 4, 2, 2, 1, 1, False)),     # non-power-of-two grids
    ("odd_grid_T_k4s2p1", Geom(2, 5, 6, 10, 12, 96, 32, 4, 4, 2, 2, 1, 1, True)),
]


def _conv_case(name, g: Geom, plan=None):
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)   # (str hashes change per process)
    x = torch.randn(g.in_shape, generator=gen).to(BF)
    wp = (torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)).to(BF)
    bias = 0.1 * torch.randn(g.Cout, generator=gen)
    rows_in = x.numel() // g.Cin
    rows_out = math.prod(g.out_shape[:3])
    rps_out = rows_out // g.N
    xd, wd = x.to(DEV), wp.to(DEV)
    # forward, plain (bf16 and fp32 results)
    check16(f"{name}/fwd", ops.conv_fwd(xd, wd, g), TB.conv_fwd(x, wp, g))
    check16(f"{name}/fwd_f32out", ops.conv_fwd(xd, wd, g, out_dtype=torch.float32), TB.conv_fwd(x, wp, g, out_dtype=torch.float32))
    for mode in (1, 2):
        bn = make_bn(g.Cin, rows_in, mode, gen, x.float() if mode == 1 else None)
        cmask = Mask((torch.rand(g.N, g.Cout, generator=gen) < 0.5).float() * 2, 1, rps_out)
        st_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
        y_ref = TB.conv_fwd(x, wp, g, bn_in=bn, bias=bias, mask=cmask, out_stats=st_ref)
        st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
        y = ops.conv_fwd(xd, wd, g, bn_in=to_dev(bn), bias=bias.to(DEV), mask=to_dev(cmask), out_stats=st)
        # (BN+ReLU'd operand elements are rounded to bf16 from an fma here and a multiply-add in the emulation: about one
        # element in 4e4 rounds the other way, which moves an output by |w| times one operand step -- visible on outputs
        # that are themselves near zero, hence the wider absolute term)
        check16(f"{name}/fwd_fused_bn{mode}", y, y_ref, atol_rel=1.5e-3)
        # statistics are sums over the STORED values: a result that rounds the other way moves them by one step of one element
        check(f"{name}/fwd_fused_bn{mode}/stats", st, st_ref, rtol=2e-3, atol_rel=2e-3)
    # residual mix in the epilogue (mopoe_conv_fwd_mix_bf16): y = 2 bn_s(s) + 0.3 mask (conv + bias)
    for mode in (1, 2):
        bn = make_bn(g.Cin, rows_in, mode, gen, x.float() if mode == 1 else None)
        sres = torch.randn(g.out_shape, generator=gen).to(BF)
        bns = make_bn(g.Cout, rows_out, mode, gen, sres.float() if mode == 1 else None)
        cmask = Mask((torch.rand(g.N, g.Cout, generator=gen) < 0.5).float() * 2, 1, rps_out)
        st_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
        y_ref = TB.conv_fwd(x, wp, g, bn_in=bn, bias=bias, mask=cmask, out_stats=st_ref, mix=(sres, bns))
        st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
        y = ops.conv_fwd(xd, wd, g, bn_in=to_dev(bn), bias=bias.to(DEV), mask=to_dev(cmask), out_stats=st,
                         mix=(sres.to(DEV), to_dev(bns)))
        check16(f"{name}/fwd_mix_bn{mode}", y, y_ref, atol_rel=1.5e-3)
        check(f"{name}/fwd_mix_bn{mode}/stats", st, st_ref, rtol=2e-3, atol_rel=2e-3)
    emask = Mask((torch.rand(g.out_shape, generator=gen) < 0.5).float() * 2, 2, rps_out)
    check16(f"{name}/fwd_emask", ops.conv_fwd(xd, wd, g, bias=bias.to(DEV), mask=to_dev(emask)),
            TB.conv_fwd(x, wp, g, bias=bias, mask=emask))
    # input gradient
    dy = torch.randn(g.out_shape, generator=gen).to(BF)
    dyd = dy.to(DEV)
    check16(f"{name}/dgrad", ops.conv_dgrad(dyd, wd, g), TB.conv_dgrad(dy, wp, g))
    check16(f"{name}/dgrad_f32out", ops.conv_dgrad(dyd, wd, g, out_dtype=torch.float32), TB.conv_dgrad(dy, wp, g, out_dtype=torch.float32))
    for mode in (1, 2):
        bn = make_bn(g.Cin, rows_in, mode, gen, x.float() if mode == 1 else None)
        s_ref = torch.zeros(2, g.Cin, dtype=torch.float64)
        dx_ref = TB.conv_dgrad(dy, wp, g, relu_bn=bn, xin=x, bwd_sums=s_ref)
        s = torch.zeros(2, g.Cin, dtype=torch.float64, device=DEV)
        dx = ops.conv_dgrad(dyd, wd, g, relu_bn=to_dev(bn), xin=xd, bwd_sums=s)
        check16(f"{name}/dgrad_relubn{mode}", dx, dx_ref)
        check(f"{name}/dgrad_relubn{mode}/sums", s, s_ref, rtol=2e-3, atol_rel=2e-3)
    # weight gradient (fp32 result)
    check(f"{name}/wgrad", ops.conv_wgrad(xd, dyd, g), TB.conv_wgrad(x, dy, g), rtol=3e-4, atol_rel=3e-4)
    bn = make_bn(g.Cin, rows_in, 1, gen, x.float())
    # (the BN+ReLU'd operand is rounded to bf16 on both sides, from an fma here and a multiply-add there: a few
    # operand elements round the other way, which moves a weight gradient more than fp32 summation order does)
    check(f"{name}/wgrad_bn", ops.conv_wgrad(xd, dyd, g, bn_in=to_dev(bn)), TB.conv_wgrad(x, dy, g, bn_in=bn),
          rtol=2e-3, atol_rel=1e-3)


@pytest.mark.parametrize("name,g", GEOMS16, ids=[n for n, _ in GEOMS16])
def test_conv_family_bf16(name, g: Geom):
    _conv_case(name, g)


@pytest.mark.parametrize("name,g", [GEOMS16[0], GEOMS16[2], GEOMS16[9], GEOMS16[12]], ids=lambda v: v if isinstance(v, str) else "")
def test_conv_every_launch_plan_bf16(name, g: Geom):
    """every (tile, split) the bf16 family offers computes the same result"""
    for tile in range(5):
        for split in (1, 2, 3, 8):
            with ops.force_plan(tile, split):
                _conv_case(f"{name}/t{tile}s{split}", g)
    for tile in (0, 2, 5, 6):      # 5 / 6: the 128 / 64 tiles on LDS-DMA (64 pixels per stage)
        if tile == 5 and min(g.Cin, g.Cout) <= 64:     # the 128 tile on a narrow layer: refused (the tuner never offers it)
            with ops.force_plan(5, 1), pytest.raises(ops.MopoeHipError):
                ops.conv_wgrad(torch.zeros(g.in_shape, dtype=BF, device=DEV), torch.zeros(g.out_shape, dtype=BF, device=DEV), g)
            continue
        for split in (1, 2, 5):
            with ops.force_plan(tile, split):
                gen = torch.Generator().manual_seed(7)
                x = torch.randn(g.in_shape, generator=gen).to(BF)
                dy = torch.randn(g.out_shape, generator=gen).to(BF)
                check(f"{name}/wgrad_t{tile}s{split}", ops.conv_wgrad(x.to(DEV), dy.to(DEV), g), TB.conv_wgrad(x, dy, g),
                      rtol=3e-4, atol_rel=3e-4)
                bn = make_bn(g.Cin, x.numel() // g.Cin, 1, gen, x.float())
                check(f"{name}/wgrad_bn_t{tile}s{split}", ops.conv_wgrad(x.to(DEV), dy.to(DEV), g, bn_in=to_dev(bn)),
                      TB.conv_wgrad(x, dy, g, bn_in=bn), rtol=2e-3, atol_rel=1e-3)


MERGE_GEOMS = [
    ("enc_64to128_b9", Geom(9, 8, 8, 16, 16, 64, 128, 4, 4, 2, 2, 1, 1, False)),            # convs: Cin = 64, rows of the tile = (tap, ci)
    ("enc_64to192_odd_grid", Geom(3, 6, 5, 12, 10, 64, 192, 4, 4, 2, 2, 1, 1, False)),      # partial column tile, no power-of-two map
    ("enc_64to64_k4", Geom(2, 8, 8, 16, 16, 64, 64, 4, 4, 2, 2, 1, 1, False)),              # half of the columns empty
    ("dec_128to64_b5", Geom(5, 8, 8, 16, 16, 128, 64, 4, 4, 2, 2, 1, 1, True)),             # transposed: Cout = 64, columns = (tap, co)
    ("dec_64to64_b3", Geom(3, 16, 16, 32, 32, 64, 64, 4, 4, 2, 2, 1, 1, True)),             # half of the rows empty
    ("dec_192to64_odd_grid", Geom(2, 5, 6, 10, 12, 192, 64, 4, 4, 2, 2, 1, 1, True)),       # partial row tile
    ("text_convT1d_64", Geom(4, 1, 16, 1, 32, 128, 64, 1, 4, 1, 2, 0, 1, True)),            # 1-D, four taps
]


@pytest.mark.parametrize("name,g", MERGE_GEOMS, ids=[n for n, _ in MERGE_GEOMS])
def test_wgrad_two_taps_per_block_bf16(name, g: Geom):
    """wgrad tile 7: two taps per block on the side of the gathered operand (64 channels there) -- the (tap, channel) rows /
    columns of its 128 x 128 tile against the emulation, with and without a split of the pixel reduction; refused with BN on
    load and where the gathered side has another channel count"""
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    x = torch.randn(g.in_shape, generator=gen).to(BF)
    dy = torch.randn(g.out_shape, generator=gen).to(BF)
    ref = TB.conv_wgrad(x, dy, g)
    for split in (1, 2, 5):
        with ops.force_plan(7, split):
            check(f"{name}/wgrad_t7s{split}", ops.conv_wgrad(x.to(DEV), dy.to(DEV), g), ref, rtol=3e-4, atol_rel=3e-4)
    with ops.force_plan(7, 1):
        bn = make_bn(g.Cin, x.numel() // g.Cin, 1, gen, x.float())
        with pytest.raises(ops.MopoeHipError):
            ops.conv_wgrad(x.to(DEV), dy.to(DEV), g, bn_in=to_dev(bn))
        g2 = Geom(2, 4, 4, 8, 8, 128, 128, 4, 4, 2, 2, 1, 1, False)
        with pytest.raises(ops.MopoeHipError):
            ops.conv_wgrad(torch.zeros(g2.in_shape, dtype=BF, device=DEV), torch.zeros(g2.out_shape, dtype=BF, device=DEV), g2)


PARITY_GEOMS = [
    ("enc_64to128_b3", Geom(3, 16, 16, 32, 32, 64, 128, 4, 4, 2, 2, 1, 1, False)),      # conv, S tile 128 (rb1's shape)
    ("enc_128to192_b5", Geom(5, 8, 8, 16, 16, 128, 192, 4, 4, 2, 2, 1, 1, False)),      # two G tiles, S tile 64 (192 = 3 x 64)
    ("enc_64to64_b2", Geom(2, 8, 16, 16, 32, 64, 64, 4, 4, 2, 2, 1, 1, False)),         # S tile 64, non-square map
    ("enc_72to136_b2", Geom(2, 8, 8, 16, 16, 72, 136, 4, 4, 2, 2, 1, 1, False)),        # partial channel tiles on both sides
    ("enc_64to256_b2", Geom(2, 8, 8, 16, 16, 64, 256, 4, 4, 2, 2, 1, 1, False)),        # two S tiles of 128
    ("dec_128to64_b5", Geom(5, 8, 8, 16, 16, 128, 64, 4, 4, 2, 2, 1, 1, True)),         # transposed: G = dy (64), S = x (128)
    ("dec_64to64_b3", Geom(3, 16, 16, 32, 32, 64, 64, 4, 4, 2, 2, 1, 1, True)),         # transposed, S tile 64
    ("dec_192to128_b2", Geom(2, 8, 8, 16, 16, 192, 128, 4, 4, 2, 2, 1, 1, True)),       # transposed, two G tiles, S tile 64
]


@pytest.mark.parametrize("name,g", PARITY_GEOMS, ids=[n for n, _ in PARITY_GEOMS])
def test_wgrad_four_taps_per_block_bf16(name, g: Geom):
    """wgrad tile 8 (round 4): one parity class of a k4 s2 p1 kernel -- four taps -- per block, the 9 x 9 big-grid pixels of an
    8 x 8 tile of small-grid pixels staged once for all of them: every tap of every class against the emulation, with and
    without a split of the pixel tiles; refused with BN on load, for other kernel shapes and for maps that are not whole tiles"""
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    x = torch.randn(g.in_shape, generator=gen).to(BF)
    dy = torch.randn(g.out_shape, generator=gen).to(BF)
    ref = TB.conv_wgrad(x, dy, g)
    csm = g.Cin if g.transposed else g.Cout
    for tile in (8, 9):
        if tile == 9 and csm % 128:
            with ops.force_plan(9, 1), pytest.raises(ops.MopoeHipError):
                ops.conv_wgrad(x.to(DEV), dy.to(DEV), g)
            continue
        for split in (1, 2, 5):
            with ops.force_plan(tile, split):
                check(f"{name}/wgrad_t{tile}s{split}", ops.conv_wgrad(x.to(DEV), dy.to(DEV), g), ref, rtol=3e-4, atol_rel=3e-4)
    with ops.force_plan(8, 1):
        bn = make_bn(g.Cin, x.numel() // g.Cin, 1, gen, x.float())
        with pytest.raises(ops.MopoeHipError):
            ops.conv_wgrad(x.to(DEV), dy.to(DEV), g, bn_in=to_dev(bn))
        for bad in (Geom(2, 6, 5, 12, 10, 64, 64, 4, 4, 2, 2, 1, 1, False),           # not whole 8 x 8 tiles
                    Geom(2, 1, 16, 1, 32, 64, 64, 1, 4, 1, 2, 0, 1, True),            # 1-D
                    Geom(2, 4, 4, 16, 16, 64, 64, 4, 4, 4, 4, 1, 1, False)):          # stride 4
            with pytest.raises(ops.MopoeHipError):
                ops.conv_wgrad(torch.zeros(bad.in_shape, dtype=BF, device=DEV), torch.zeros(bad.out_shape, dtype=BF, device=DEV), bad)


def _glds_case(name, g: Geom):
    """the ops the LDS-DMA tiles serve (csrc/conv_gemm_bf16_glds.inc): forward convs whose operand needs no BN on load
    (plain, fp32 result, bias + element mask, bias + statistics = a projection shortcut) and every form of the input gradient"""
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    x = torch.randn(g.in_shape, generator=gen).to(BF)
    wp = (torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)).to(BF)
    bias = 0.1 * torch.randn(g.Cout, generator=gen)
    rows_in = x.numel() // g.Cin
    rps_out = math.prod(g.out_shape[:3]) // g.N
    xd, wd = x.to(DEV), wp.to(DEV)
    if g.Cin % 64 == 0:
        check16(f"{name}/fwd", ops.conv_fwd(xd, wd, g), TB.conv_fwd(x, wp, g))
        check16(f"{name}/fwd_f32out", ops.conv_fwd(xd, wd, g, out_dtype=torch.float32), TB.conv_fwd(x, wp, g, out_dtype=torch.float32))
        emask = Mask((torch.rand(g.out_shape, generator=gen) < 0.5).float() * 2, 2, rps_out)
        check16(f"{name}/fwd_emask", ops.conv_fwd(xd, wd, g, bias=bias.to(DEV), mask=to_dev(emask)),
                TB.conv_fwd(x, wp, g, bias=bias, mask=emask))
        st_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
        st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
        check16(f"{name}/fwd_shortcut", ops.conv_fwd(xd, wd, g, bias=bias.to(DEV), out_stats=st),
                TB.conv_fwd(x, wp, g, bias=bias, out_stats=st_ref))
        check(f"{name}/fwd_shortcut/stats", st, st_ref, rtol=2e-3, atol_rel=2e-3)
    if g.Cout % 64 == 0:
        dy = torch.randn(g.out_shape, generator=gen).to(BF)
        dyd = dy.to(DEV)
        check16(f"{name}/dgrad", ops.conv_dgrad(dyd, wd, g), TB.conv_dgrad(dy, wp, g))
        check16(f"{name}/dgrad_f32out", ops.conv_dgrad(dyd, wd, g, out_dtype=torch.float32), TB.conv_dgrad(dy, wp, g, out_dtype=torch.float32))
        for mode in (1, 2):
            bn = make_bn(g.Cin, rows_in, mode, gen, x.float() if mode == 1 else None)
            s_ref = torch.zeros(2, g.Cin, dtype=torch.float64)
            dx_ref = TB.conv_dgrad(dy, wp, g, relu_bn=bn, xin=x, bwd_sums=s_ref)
            s = torch.zeros(2, g.Cin, dtype=torch.float64, device=DEV)
            dx = ops.conv_dgrad(dyd, wd, g, relu_bn=to_dev(bn), xin=xd, bwd_sums=s)
            check16(f"{name}/dgrad_relubn{mode}", dx, dx_ref)
            check(f"{name}/dgrad_relubn{mode}/sums", s, s_ref, rtol=2e-3, atol_rel=2e-3)


def _glds_xform_case(name, g: Geom):
    """forward convs with BN -> ReLU on the gathered operand on the LDS-DMA tiles (weights by DMA, activations through
    registers): conv1 / conv2 of a residual block, with and without the residual mix in the epilogue"""
    if g.Cin % 64:
        return
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()) % 10000)
    x = torch.randn(g.in_shape, generator=gen).to(BF)
    wp = (torch.randn(g.taps, g.Cin, g.Cout, generator=gen) / math.sqrt(g.taps * g.Cin)).to(BF)
    bias = 0.1 * torch.randn(g.Cout, generator=gen)
    rows_in = x.numel() // g.Cin
    rows_out = math.prod(g.out_shape[:3])
    rps_out = rows_out // g.N
    xd, wd = x.to(DEV), wp.to(DEV)
    for mode in (1, 2):
        bn = make_bn(g.Cin, rows_in, mode, gen, x.float() if mode == 1 else None)
        cmask = Mask((torch.rand(g.N, g.Cout, generator=gen) < 0.5).float() * 2, 1, rps_out)
        st_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
        y_ref = TB.conv_fwd(x, wp, g, bn_in=bn, bias=bias, mask=cmask, out_stats=st_ref)
        st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
        y = ops.conv_fwd(xd, wd, g, bn_in=to_dev(bn), bias=bias.to(DEV), mask=to_dev(cmask), out_stats=st)
        check16(f"{name}/fwd_fused_bn{mode}", y, y_ref, atol_rel=1.5e-3)
        check(f"{name}/fwd_fused_bn{mode}/stats", st, st_ref, rtol=2e-3, atol_rel=2e-3)
        sres = torch.randn(g.out_shape, generator=gen).to(BF)
        bns = make_bn(g.Cout, rows_out, mode, gen, sres.float() if mode == 1 else None)
        st_ref = torch.zeros(2, g.Cout, dtype=torch.float64)
        y_ref = TB.conv_fwd(x, wp, g, bn_in=bn, bias=bias, mask=cmask, out_stats=st_ref, mix=(sres, bns))
        st = torch.zeros(2, g.Cout, dtype=torch.float64, device=DEV)
        y = ops.conv_fwd(xd, wd, g, bn_in=to_dev(bn), bias=bias.to(DEV), mask=to_dev(cmask), out_stats=st,
                         mix=(sres.to(DEV), to_dev(bns)))
        check16(f"{name}/fwd_mix_bn{mode}", y, y_ref, atol_rel=1.5e-3)
        check(f"{name}/fwd_mix_bn{mode}/stats", st, st_ref, rtol=2e-3, atol_rel=2e-3)


GLDS_GEOMS = [GEOMS16[i] for i in (0, 1, 2, 3, 4, 7, 9, 12, 13)] + [
    ("enc_k4s2p1_64to128_b9_ragged", Geom(9, 8, 8, 16, 16, 64, 128, 4, 4, 2, 2, 1, 1, False)),   # several M tiles, a partial last one
    ("dec_k4s2p1_64to64_b5", Geom(5, 16, 16, 32, 32, 64, 64, 4, 4, 2, 2, 1, 1, True)),
    ("odd_grid_k4s2p1_64to192", Geom(3, 6, 5, 12, 10, 64, 192, 4, 4, 2, 2, 1, 1, False)),
]


@pytest.mark.parametrize("name,g", GLDS_GEOMS, ids=[n for n, _ in GLDS_GEOMS])
def test_conv_lds_dma_tiles_bf16(name, g: Geom):
    """every LDS-DMA tile (two to four LDS buffers; 128x128, 256x128, 128x64, 64x64) with and without a split reduction against
    the emulation, and against the register-staged kernel's result bit for bit where the summation order is the same"""
    for tile in (5, 6, 7, 9, 10, 11):
        for split in (1, 3):
            with ops.force_plan(tile, split):
                _glds_case(f"{name}/t{tile}s{split}", g)
                if tile in (5, 7, 9, 11):
                    _glds_xform_case(f"{name}/t{tile}s{split}x", g)


@pytest.mark.parametrize("n,hs,ws", [(2, 16, 16), (3, 5, 16), (2, 6, 6)], ids=["b2_16x16", "b3_5x16_partial_tile", "b2_6x6_streaming"])
def test_edge_layers_bf16(n, hs, ws):
    """image stem / head with the wide tensor in bf16 (fp32 pixels, taps and tap gradients): MFMA forms (C = 64, rows of
    16 k pixels) and the streaming kernels"""
    gen = torch.Generator().manual_seed(5)
    gs = Geom(n, hs, ws, 2 * hs, 2 * ws, 1, 64, 3, 3, 2, 2, 1, 1, False)
    img = torch.rand(gs.in_shape, generator=gen)
    w = torch.randn(9, 1, 64, generator=gen) / 3
    st_ref = torch.zeros(2, 64, dtype=torch.float64)
    st = torch.zeros(2, 64, dtype=torch.float64, device=DEV)
    y = ops.conv_fwd(img.to(DEV), w.to(DEV), gs, out_stats=st, out_dtype=BF)
    check16("stem/fwd", y, TB.conv_fwd(img, w, gs, out_stats=st_ref, out_dtype=BF))
    check("stem/stats", st, st_ref, rtol=2e-3, atol_rel=2e-3)
    dy = torch.randn(gs.out_shape, generator=gen).to(BF)
    check("stem/wgrad", ops.conv_wgrad(img.to(DEV), dy.to(DEV), gs), TB.conv_wgrad(img, dy, gs), rtol=3e-4, atol_rel=3e-4)
    gh = Geom(n, hs, ws, 2 * hs, 2 * ws, 64, 1, 3, 3, 2, 2, 1, 1, True)
    x = torch.randn(gh.in_shape, generator=gen).to(BF)
    wh = torch.randn(9, 64, 1, generator=gen) / 8
    b = torch.tensor([0.3])
    out = ops.conv_fwd(x.to(DEV), wh.to(DEV), gh, bias=b.to(DEV))
    assert out.dtype == torch.float32
    check("head/fwd", out, TB.conv_fwd(x, wh, gh, bias=b), rtol=2e-4, atol_rel=2e-4)
    gimg = torch.randn(gh.out_shape, generator=gen)
    check16("head/dgrad", ops.conv_dgrad(gimg.to(DEV), wh.to(DEV), gh, out_dtype=BF), TB.conv_dgrad(gimg, wh, gh, out_dtype=BF))
    check("head/wgrad", ops.conv_wgrad(x.to(DEV), gimg.to(DEV), gh), TB.conv_wgrad(x, gimg, gh), rtol=3e-4, atol_rel=3e-4)


@pytest.mark.parametrize("rows,c", [(4096, 64), (300, 320), (7, 640), (70000, 128)])
def test_block_glue_bf16(rows, c):
    gen = torch.Generator().manual_seed(rows + c)
    s = torch.randn(rows, c, generator=gen).to(BF)
    m = torch.randn(rows, c, generator=gen).to(BF)
    g = torch.randn(rows, c, generator=gen).to(BF)
    bn = make_bn(c, rows, 1, gen, s.float())
    sd, md, gd, bnd = s.to(DEV), m.to(DEV), g.to(DEV), to_dev(bn)
    st_ref, st = torch.zeros(2, c, dtype=torch.float64), torch.zeros(2, c, dtype=torch.float64, device=DEV)
    check16("bn_relu_apply", ops.bn_relu_apply(sd, bnd), TB.bn_relu_apply(s, bn))
    check16("block_out_fwd", ops.block_out_fwd(sd, md, bnd, out_stats=st), TB.block_out_fwd(s, m, bn, out_stats=st_ref))
    check("block_out_fwd/stats", st, st_ref, rtol=2e-3, atol_rel=2e-3)
    sums_ref = TB.bn_bwd_reduce(g, s, bn)
    sums = ops.bn_bwd_reduce(gd, sd, bnd)
    check("bn_bwd_reduce", sums, sums_ref, rtol=1e-4, atol_rel=1e-4)
    n = 5 if rows % 5 == 0 else 1
    mask = Mask((torch.rand(n, c, generator=gen) < 0.5).float() * 2, 1, rows // n)
    ref = TB.block_out_bwd(g, s, bn, sums_ref, mask, want_colsum_dm=True)
    got = ops.block_out_bwd(gd, sd, bnd, sums_ref.to(DEV), to_dev(mask), want_colsum_dm=True)
    for nm, a, b in zip(("dm", "ds", "dgamma", "dbeta", "cdm", "cds"), got, ref):
        if nm == "cds":
            # the column sums of a BatchNorm backward are analytically zero: what is left is the rounding of the stored
            # elements, so the bound is one rounding step of an element times sqrt(rows), not a fraction of the sum
            bound = ULP * ref[1].float().abs().max().item() * rows ** 0.5
            assert (a.cpu() - b).abs().max().item() <= bound, (nm, (a.cpu() - b).abs().max().item(), bound)
            continue
        (check16 if a.dtype == BF else (lambda k, u, v: check(k, u, v, rtol=2e-3, atol_rel=2e-3)))(f"block_out_bwd/{nm}", a, b)
    x = torch.randn(rows, c, generator=gen).to(BF)
    add = torch.randn(rows, c, generator=gen).to(BF)
    bnx = make_bn(c, rows, 1, gen, x.float())
    dy = g
    sums_x = torch.stack([dy.float().double().sum(0), (dy.float() * ((x.float() - TB.bn_coef(bnx)[0]) * TB.bn_coef(bnx)[1])).double().sum(0)])
    nsum_ref, nsum = torch.zeros(2, c, dtype=torch.float64), torch.zeros(2, c, dtype=torch.float64, device=DEV)
    ref = TB.bn_bwd_apply(dy, x, bnx, sums_x, mask=mask, add=add, want_colsum=True, next_s=s, next_bn=bn, next_sums=nsum_ref)
    got = ops.bn_bwd_apply(gd, x.to(DEV), to_dev(bnx), sums_x.to(DEV), mask=to_dev(mask), add=add.to(DEV), want_colsum=True,
                           next_s=sd, next_bn=bnd, next_sums=nsum)
    check16("bn_bwd_apply/dx", got[0], ref[0])
    check("bn_bwd_apply/colsum", got[3], ref[3], rtol=2e-3, atol_rel=2e-3)
    check("bn_bwd_apply/next_sums", nsum, nsum_ref, rtol=2e-3, atol_rel=2e-3)
    check("colsum", ops.colsum(sd), TB.colsum(s), rtol=1e-4, atol_rel=1e-4)


@pytest.mark.parametrize("n,rps,drop", [(3, 64, True), (2, 1024, False), (5, 32, True), (1, 4096, True)])
def test_block_front_streaming_kernels_bf16(n, rps, drop):
    """csrc/pointwise.hip (round 4): the front of a residual block -- bn1 -> relu -> conv1 (1x1, 64 channels) -> Dropout2d -> bn2
    -> relu -- without ever writing d1: statistics pass, a2 pass and the fused backward (bn2 backward + conv1 input / weight
    gradients + bn1's backward sums) against the emulation, which composes exactly the ops these kernels replace; then the
    replaced HIP ops themselves (conv_fwd + bn_relu_apply / bn_bwd_apply + conv_dgrad + conv_wgrad) on the same inputs."""
    c, rows = 64, n * rps
    gen = torch.Generator().manual_seed(n * 1000 + rps)
    x = torch.randn(rows, c, generator=gen).to(BF).view(n, rps, 1, c)
    w1 = (torch.randn(1, c, c, generator=gen) / 8).to(BF)
    bias = 0.1 * torch.randn(c, generator=gen)
    bn1 = make_bn(c, rows, 1, gen, x.float())
    mask = Mask((torch.rand(n, c, generator=gen) < 0.5).float() * 2, 1, rps) if drop else None
    g1 = Geom(n, rps, 1, rps, 1, c, c, 1, 1, 1, 1, 0, 0, False)
    assert ops.block_front_supported(x.to(DEV), g1, to_dev(mask))
    xd, wd, bd, bn1d, md = x.to(DEV), w1.to(DEV), bias.to(DEV), to_dev(bn1), to_dev(mask)
    # statistics of d1
    st_ref, st = torch.zeros(2, c, dtype=torch.float64), torch.zeros(2, c, dtype=torch.float64, device=DEV)
    TB.block_front_stats(x, w1, bias, bn1, mask, st_ref)
    ops.block_front_stats(xd, wd, bd, bn1d, md, st)
    check("front/stats", st, st_ref, rtol=2e-3, atol_rel=2e-3)
    # ... which equal those of the conv kernel that used to write d1
    st_old = torch.zeros(2, c, dtype=torch.float64, device=DEV)
    d1_old = ops.conv_fwd(xd, wd, g1, bn_in=bn1d, bias=bd, mask=md, out_stats=st_old)
    check("front/stats_vs_conv_fwd", st, st_old, rtol=2e-3, atol_rel=2e-3)
    bn2 = Bn(torch.rand(c, generator=gen) + 0.5, 0.1 * torch.randn(c, generator=gen), 1, st_ref.clone(), rows)
    bn2d = to_dev(bn2)
    a2_ref = TB.block_front_apply(x, w1, bias, bn1, bn2, mask)
    a2 = ops.block_front_apply(xd, wd, bd, bn1d, bn2d, md)
    # (two rounding steps between the sums and a2: d1 is rounded to bf16, then relu(bn2(d1)) is -- a d1 that lands next to a
    # rounding boundary may round the other way on the two sides, which moves a2 by up to one more step)
    check("front/a2", a2.float(), a2_ref.float(), rtol=2.5 * ULP, atol_rel=2e-3)
    check("front/a2_vs_bn_relu_apply", a2.float(), ops.bn_relu_apply(d1_old, bn2d).float(), rtol=2.5 * ULP, atol_rel=2e-3)
    # backward
    dh2 = (torch.randn(rows, c, generator=gen) * (a2_ref.view(rows, c).float() > 0)).to(BF).view(x.shape)
    mean2, rstd2, _, _ = TB.bn_coef(bn2)
    d1_ref = TB._front_d1(x, w1, bias, bn1, mask)[1]
    sums2 = torch.stack([dh2.float().reshape(rows, c).double().sum(0),
                         (dh2.float() * ((d1_ref - mean2) * rstd2)).reshape(rows, c).double().sum(0)])
    out_ref = dict(s1=torch.zeros(2, c, dtype=torch.float64), dw=torch.zeros(1, c, c), db=torch.zeros(c), dg=torch.zeros(c), dbt=torch.zeros(c))
    dh1_ref = TB.block_front_bwd(x, dh2, w1, bias, bn1, bn2, mask, sums2, out_ref["s1"], out_ref["dw"], out_ref["db"], out_ref["dg"], out_ref["dbt"])
    s1 = torch.zeros(2, c, dtype=torch.float64, device=DEV)
    dw, db, dg, dbt = (torch.zeros(1, c, c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(c, device=DEV), torch.zeros(c, device=DEV))
    dh1 = ops.block_front_bwd(xd, dh2.to(DEV), wd, bd, bn1d, bn2d, md, sums2.to(DEV), s1, dw, db, dg, dbt)
    check("front/dh1", dh1.float(), dh1_ref.float(), rtol=2.5 * ULP, atol_rel=4e-3)
    check("front/sums1", s1, out_ref["s1"], rtol=5e-3, atol_rel=5e-3)
    check("front/dw1", dw, out_ref["dw"], rtol=2e-3, atol_rel=2e-3)
    # (the column sums of a BatchNorm backward are analytically zero: what is left is the rounding of the stored dc1, so the
    # bound is one rounding step of an element times sqrt(rows), not a fraction of the sum)
    dc1_scale = max(out_ref["dw"].abs().max().item() / max(rows ** 0.5, 1.0), 1e-3)
    noise = ULP * float(dh2.float().abs().max()) * float((bn2.gamma * TB.bn_coef(bn2)[1]).abs().max()) * rows ** 0.5 * 2
    assert (db.cpu() - out_ref["db"]).abs().max().item() <= noise, ((db.cpu() - out_ref["db"]).abs().max().item(), noise)
    check("front/dgamma2", dg, out_ref["dg"], rtol=1e-6, atol_rel=1e-6)
    check("front/dbeta2", dbt, out_ref["dbt"], rtol=1e-6, atol_rel=1e-6)
    # the ops it replaces, on the same inputs: bn_bwd_apply -> conv_dgrad (+ bn1 sums) -> conv_wgrad
    dc1_old, _, _, cdc1_old = ops.bn_bwd_apply(dh2.to(DEV), d1_old, bn2d, sums2.to(DEV), mask=md, want_colsum=True)
    s1_old = torch.zeros(2, c, dtype=torch.float64, device=DEV)
    dh1_old = ops.conv_dgrad(dc1_old, wd, g1, relu_bn=bn1d, xin=xd, bwd_sums=s1_old)
    check("front/dh1_vs_old_ops", dh1.float(), dh1_old.float(), rtol=2.5 * ULP, atol_rel=4e-3)
    check("front/sums1_vs_old_ops", s1, s1_old, rtol=5e-3, atol_rel=5e-3)
    check("front/dw1_vs_old_ops", dw, ops.conv_wgrad(xd, dc1_old, g1, bn_in=bn1d), rtol=2e-3, atol_rel=2e-3)
    assert (db - cdc1_old).abs().max().item() <= noise, ((db - cdc1_old).abs().max().item(), noise)
    # conv2's input gradient with its ReLU mask / x-hat taken from a2 (bn mode 3) == from d1 (mode 1)
    g2 = Geom(n, 1, rps // 2, 1, rps, c, 128, 1, 4, 1, 2, 0, 1, False) if rps % 2 == 0 else None
    if g2 is not None:
        w2 = (torch.randn(g2.taps, c, 128, generator=gen) / 16).to(BF).to(DEV)
        dm = torch.randn(g2.out_shape, generator=gen).to(BF).to(DEV)
        xa = d1_old.view(g2.in_shape)
        sa, sb = torch.zeros(2, c, dtype=torch.float64, device=DEV), torch.zeros(2, c, dtype=torch.float64, device=DEV)
        ga = ops.conv_dgrad(dm, w2, g2, relu_bn=bn2d, xin=xa, bwd_sums=sa)
        bn2y = Bn(bn2d.gamma, bn2d.beta, 3, sums=bn2d.sums, count=bn2d.count)
        gb = ops.conv_dgrad(dm, w2, g2, relu_bn=bn2y, xin=a2.view(g2.in_shape), bwd_sums=sb)
        check16("front/conv2_dgrad_mask_from_a2", gb, ga.cpu())
        check("front/conv2_dgrad_sums_from_a2", sb, sa, rtol=2e-2, atol_rel=2e-2)    # (x-hat through the bf16-rounded a2)


def test_embedding_bf16():
    gen = torch.Generator().manual_seed(3)
    table = torch.randn(101, 64, generator=gen)
    ids = torch.randint(0, 101, (6, 16), generator=gen).float()
    out = ops.embedding_fwd(ids.to(DEV), table.to(DEV), out_dtype=BF)
    assert out.dtype == BF and torch.equal(out.cpu(), TB.embedding_fwd(ids, table, out_dtype=BF))
    gout = torch.randn(6, 16, 64, generator=gen).to(BF)
    check("embedding_bwd", ops.embedding_bwd(ids.to(DEV), gout.to(DEV), 101, 0), TB.embedding_bwd(ids, gout, 101, 0), rtol=1e-5, atol_rel=1e-5)


def test_bf16_rejects_unsupported_channel_counts():
    g = Geom(2, 4, 4, 8, 8, 20, 24, 4, 4, 2, 2, 1, 1, False)     # K = 20 is not a multiple of 32
    x = torch.zeros(g.in_shape, dtype=BF, device=DEV)
    w = torch.zeros(g.taps, g.Cin, g.Cout, dtype=BF, device=DEV)
    with pytest.raises(ops.MopoeHipError):
        ops.conv_fwd(x, w, g)
    with pytest.raises(ops.MopoeHipError):
        ops.block_out_fwd(torch.zeros(8, 20, dtype=BF, device=DEV), torch.zeros(8, 20, dtype=BF, device=DEV),
                          to_dev(make_bn(20, 8, 2, torch.Generator().manual_seed(0))))


# ---------------------------------------------------------------------------------------------------------------
# whole model
# ---------------------------------------------------------------------------------------------------------------
def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)


def test_small_model_bf16_vs_oracle():
    """a small configuration the bf16 family accepts (every GEMM K a multiple of 32; 64 px, class_dim 32, DIM 32, vocab 200,
    B = 8): BatchNorm batch statistics, dropout (the oracle's seeded masks replayed), eval -- fixtures G7 small_bf16_*"""
    check_bf16_case("small_bf16_nodrop", small_batch=True)
    check_bf16_case("small_bf16_dropout", small_batch=True)
    check_bf16_case("small_bf16_eval", grads=False)


def test_c3_full_size_bf16(table_plans):
    """BASELINE config #3 exactly: 128 px, class_dim 128, B = 256, bf16 -- forward scalars against the bf16-mode oracle and
    the REFERENCE's fp32 run, EVERY parameter gradient against both (fixture G7 c3_b256_bf16: the CPU passes at B = 256 ran
    in the build container, oracle/gen_g7.py), then three Adam steps (finite gradients, decreasing loss)."""
    exp, cfg, _ = check_bf16_case("c3_b256_bf16")
    _three_steps(exp, cfg, 256, seed=92)


def test_c5_full_size_bf16_and_fp32(table_plans):
    """BASELINE config #5 exactly: 256 px (the stride-4 block), class_dim 256, B = 32: the bf16 family's forward scalars and
    every parameter gradient (fixture G7 c5_b32_bf16), and the fp32 family's forward scalars on the same inputs against the
    reference's fp32 run (only B = 4 had run)."""
    exp, cfg, g = check_bf16_case("c5_b32_bf16")
    _three_steps(exp, cfg, 32, seed=96)
    del exp
    _, sd, batch, eps, _ = g7_inputs(g)
    exp = build_exp(cfg, sd, "cuda", "train_nodrop", eps=eps)
    got = RE.basic_routine_epoch(exp, ({k: v.cuda() for k, v in batch.items()}, None))
    assert _rel(got["total_loss"].item(), float(g["fp32/total_loss"])) <= 1e-4
    for k, v in got["klds"].items():
        ref = float(g[f"fp32/klds/{k}"])
        assert _rel(v.item(), ref) <= 1e-4 + 1e-6 / abs(ref), k
    for k, v in got["log_probs"].items():
        assert _rel(v.item(), float(g[f"fp32/log_probs/{k}"])) <= 1e-4, k
    _three_steps(exp, cfg, 32, seed=99)


@pytest.mark.parametrize("form", ["eager", "graph"])
def test_bf16_trajectory_vs_fp32_reference(form):
    """ten Adam steps of the bf16 family at BASELINE config #3's architecture (B = 16) against the REFERENCE's own fp32
    trajectory (fixture G7 traj_c3_b16: its module under torch.optim.Adam, run in the build container), loss by loss at
    SURVEY 8c's bf16 tolerance (rtol 2e-2): the bf16 rounding points must not bend the optimisation trajectory.  Both step
    forms: eager train_step and the captured hipGraph."""
    g = load("g7_traj_c3_b16")
    cfg = cfg_from(g["cfg"])
    order, lr, ref = [int(i) for i in g["order"]], float(g["lr"]), [float(v) for v in g["losses"]]
    sd = R.init_state(cfg, seed=int(g["seed_weights"]))
    np.testing.assert_allclose(weights_fingerprint(sd), g["weights_fingerprint"], rtol=1e-12)
    batches = [R.synthetic_batch(cfg, 16, seed=int(g["seed_batch0"]) + i) for i in range(max(order) + 1)]
    eps = batches[0][1]
    exp = build_exp(cfg, {k: v.clone() for k, v in sd.items()}, "cuda", "train_nodrop", eps=eps, compute_dtype="bf16")
    exp.flags.initial_learning_rate = lr
    exp.set_optimizer(capturable=(form == "graph"))
    pack = RE.ScalarPack(exp.flags.device)
    dev = lambda i: ({k: v.cuda() for k, v in batches[i][0].items()}, None)
    got = []
    if form == "eager":
        for i in order:
            RE.train_step(exp, dev(i), None, pack)
            got.append(pack.read()["total_loss"])
    else:
        step = RE.GraphedTrainStep(exp, dev(0), pack, None, warmup=2)     # (= the first two steps of `order`, eager)
        got = [None, None]
        for i in order[2:]:
            step(dev(i))
            got.append(pack.read()["total_loss"])
    for k, (gv, r) in enumerate(zip(got, ref)):
        if gv is None:
            continue
        _log(f"traj_{form} step {k}: hip_bf16={gv:.6g} reference_fp32={r:.6g} rel={_rel(gv, r):.2e}")
        assert _rel(gv, r) <= 2e-2, (form, k, gv, r)
    assert ref[-1] < ref[0]      # (the trajectory moves: the comparison is not of ten copies of one number)


def _three_steps(exp, cfg, nrow, seed):
    batch, _ = R.synthetic_batch(cfg, nrow, seed=seed)
    exp.flags.initial_learning_rate = 1e-5
    exp.set_optimizer(capturable=False)
    losses = []
    for _ in range(3):
        out = RE.train_step(exp, ({k: v.cuda() for k, v in batch.items()}, None))
        losses.append(out["total_loss"].item())
        bad = [n for n, p in exp.mm_vae.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        assert not bad, bad[:5]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_graphed_bf16_step_matches_eager():
    """the captured step in bf16 (weight-copy refresh inside the graph) follows the eager bf16 trajectory"""
    cfg = R.Cfg(img_size=64, class_dim=32, DIM_img=32, DIM_text=32, vocab_size=200, batch_size=8)
    sd = R.init_state(cfg, seed=4)
    batches = [R.synthetic_batch(cfg, 8, seed=10 + i) for i in range(4)]
    eps = batches[0][1]
    dev = lambda b: ({k: v.cuda() for k, v in b[0].items()}, None)
    out = {}
    for kind in ("eager", "graph"):
        exp = build_exp(cfg, {k: v.clone() for k, v in sd.items()}, "cuda", "train_nodrop", eps=eps, compute_dtype="bf16")
        exp.flags.initial_learning_rate = 1e-3
        exp.set_optimizer(capturable=(kind == "graph"))
        pack = RE.ScalarPack(exp.flags.device)
        losses = []
        if kind == "eager":
            for i in (0, 1, 2, 3):
                RE.train_step(exp, dev(batches[i]), None, pack)
                losses.append(pack.read()["total_loss"])
        else:
            step = RE.GraphedTrainStep(exp, dev(batches[0]), pack, warmup=1)
            losses.append(pack.read()["total_loss"])
            for i in (1, 2, 3):
                step(dev(batches[i]))
                losses.append(pack.read()["total_loss"])
        out[kind] = losses
    for a, b in zip(out["eager"], out["graph"]):
        assert np.isfinite(a) and abs(a - b) <= 2e-3 * abs(a), out
