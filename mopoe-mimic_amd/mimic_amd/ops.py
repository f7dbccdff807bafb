"""ctypes binding of libmopoe_hip.so (include/mopoe_hip.h): the ONLY compute backend of mimic_amd.

There is deliberately no CPU / eager fallback here: if the HIP library is missing or the tensors are
not on a GPU the ops raise.  (CPU tests of the host logic monkeypatch this module's functions with
the torch emulation under tests/, which is test infrastructure, not a product path.)

Every function takes/returns channels-last tensors (see the header for the layout contract): fp32, or -- the bf16
storage family of BASELINE configs #3 / #5 -- bfloat16 activations and bf16 weight copies with fp32 accumulation; the
entry point is chosen by the dtype of the activation operand.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from functools import lru_cache
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOPOE_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libmopoe_hip.so")  # env override: A/B builds
ABI_VERSION = 19

RES_A, RES_B = 2.0, 0.3
BN_EPS = 1e-5
SUBSET_MASKS = (1, 2, 4, 3, 5, 6, 7)  # bit0 PA, bit1 Lateral, bit2 text; reference subset order


class MopoeHipError(RuntimeError):
    pass


# ----------------------------------------------------------------------------------------------
# plain-data descriptors shared with the C ABI
# ----------------------------------------------------------------------------------------------
class _Geom(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("N", "Hs", "Ws", "Hb", "Wb", "Cin", "Cout", "kh", "kw", "sh", "sw", "ph", "pw", "transposed")]


class _BnRef(C.Structure):
    _fields_ = [("sums", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("rmean", C.c_void_p),
                ("rvar", C.c_void_p), ("inv_count", C.c_double), ("eps", C.c_float), ("C", C.c_int32),
                ("mode", C.c_int32)]


class _MaskRef(C.Structure):
    _fields_ = [("mask", C.c_void_p), ("kind", C.c_int32), ("rows_per_sample", C.c_int32)]


class _MixRef(C.Structure):   # mopoe_mix_ref
    _fields_ = [("s", C.c_void_p), ("bn", _BnRef), ("a", C.c_float), ("b", C.c_float)]


class _RunDesc(C.Structure):
    _fields_ = [("sums", C.c_void_p), ("rmean", C.c_void_p), ("rvar", C.c_void_p), ("C", C.c_int32),
                ("count", C.c_int32)]


class _AdamSeg(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("p16", C.c_void_p),
                ("n", C.c_int64)]


@dataclass(frozen=True)
class Geom:
    """Conv / ConvTranspose geometry; 'small' grid = conv output / convT input (header: mopoe_conv_geom)."""
    N: int
    Hs: int
    Ws: int
    Hb: int
    Wb: int
    Cin: int
    Cout: int
    kh: int
    kw: int
    sh: int
    sw: int
    ph: int
    pw: int
    transposed: bool

    def __post_init__(self):
        object.__setattr__(self, "_hash", hash((self.N, self.Hs, self.Ws, self.Hb, self.Wb, self.Cin, self.Cout, self.kh,
                                                self.kw, self.sh, self.sw, self.ph, self.pw, self.transposed)))

    def __hash__(self):   # Geoms key the plan / struct caches on every launch: hash once
        return self._hash

    @property
    def in_shape(self):
        return (self.N, self.Hs, self.Ws, self.Cin) if self.transposed else (self.N, self.Hb, self.Wb, self.Cin)

    @property
    def out_shape(self):
        return (self.N, self.Hb, self.Wb, self.Cout) if self.transposed else (self.N, self.Hs, self.Ws, self.Cout)

    @property
    def taps(self):
        return self.kh * self.kw

    def with_batch(self, n):
        return _with_batch(self, n)

    def c(self):
        return _geom_struct(self)


@lru_cache(maxsize=None)
def _with_batch(g: "Geom", n: int) -> "Geom":
    return Geom(n, *[getattr(g, f) for f in
                     ("Hs", "Ws", "Hb", "Wb", "Cin", "Cout", "kh", "kw", "sh", "sw", "ph", "pw", "transposed")])


@lru_cache(maxsize=None)
def _geom_struct(g: "Geom") -> "_Geom":
    return _Geom(g.N, g.Hs, g.Ws, g.Hb, g.Wb, g.Cin, g.Cout, g.kh, g.kw, g.sh, g.sw, g.ph, g.pw, int(g.transposed))


@dataclass
class Bn:
    """A BatchNorm to be applied / inverted inside another kernel (header: mopoe_bn_ref)."""
    gamma: torch.Tensor
    beta: torch.Tensor
    mode: int                               # 1 batch statistics, 2 running statistics, 3 = 1 with the ACTIVATION as xin (conv_dgrad)
    sums: Optional[torch.Tensor] = None     # double [2, C] (mode 1)
    count: int = 0                          # rows the sums were taken over (mode 1)
    rmean: Optional[torch.Tensor] = None
    rvar: Optional[torch.Tensor] = None
    eps: float = BN_EPS

    def c(self):
        return _BnRef(_p(self.sums), _p(self.gamma), _p(self.beta), _p(self.rmean), _p(self.rvar),
                      (1.0 / self.count) if self.count else 0.0, self.eps, self.gamma.numel(), self.mode)


@dataclass
class Mask:
    """Dropout multiplier (values 0 or 2).  kind 1: [N, C] per-sample channel mask (nn.Dropout2d);
    kind 2: [rows, C] elementwise (nn.Dropout)."""
    mask: torch.Tensor
    kind: int
    rows_per_sample: int = 1

    def c(self):
        return _MaskRef(_p(self.mask), self.kind, self.rows_per_sample)


_NO_BN = _BnRef(None, None, None, None, None, 0.0, 0.0, 0, 0)
_NO_MASK = _MaskRef(None, 0, 1)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# ----------------------------------------------------------------------------------------------
# library loading
# ----------------------------------------------------------------------------------------------
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MopoeHipError(f"{LIB_PATH} not built: run `python __graft_entry__.py build` "
                                "(mimic_amd has no CPU fallback)")
        _lib = C.CDLL(LIB_PATH)
        _lib.mopoe_last_error.restype = C.c_char_p
        _lib.mopoe_abi_version.restype = C.c_int
        if _lib.mopoe_abi_version() != ABI_VERSION:
            raise MopoeHipError("libmopoe_hip.so ABI version mismatch")
    return _lib


def _check(rc: int):
    if rc != 0:
        raise MopoeHipError(f"libmopoe_hip error {rc}: {lib().mopoe_last_error().decode()}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device=None) -> int:
    """current HIP stream of the device as an integer (fast path: torch.cuda.current_stream() costs ~7 us)"""
    if _raw_stream is not None:
        idx = torch.cuda.current_device() if device is None or device.index is None else device.index
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def _stream():
    return C.c_void_p(_stream_ptr())


def _dev(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise MopoeHipError("mimic_amd ops need GPU tensors (no CPU fallback)")
            if not t.is_contiguous():
                raise MopoeHipError("mimic_amd ops need contiguous tensors")


def _bn(bn: Optional[Bn]):
    return C.byref(bn.c()) if bn is not None else C.byref(_NO_BN)


def _mask(m: Optional[Mask]):
    return C.byref(m.c()) if m is not None else C.byref(_NO_MASK)


_conv_ws = {}


def _workspace(device):
    """Per-device scratch for the split reductions of small-grid conv layers (C ABI: caller-owned)."""
    key = (device.index, _stream_ptr(device))
    if key not in _conv_ws:
        lib().mopoe_conv_workspace_bytes.restype = C.c_size_t
        nbytes = int(lib().mopoe_conv_workspace_bytes())
        _conv_ws[key] = (torch.zeros(nbytes, dtype=torch.uint8, device=device), nbytes)   # (arrival counters start at zero)
    return _conv_ws[key]


def new_stats(n_channels: int, device) -> torch.Tensor:
    return torch.zeros(2, n_channels, dtype=torch.float64, device=device)


# ----------------------------------------------------------------------------------------------
# convolution family
# ----------------------------------------------------------------------------------------------
class _Plan(C.Structure):
    """header: mopoe_conv_plan"""
    _fields_ = [("tile", C.c_int32), ("split", C.c_int32)]


# Launch plans.  The library has a static tile / split heuristic; the mirror can do better by measuring: the
# first time a (op, geometry, fusion) triple is met (i.e. during the first warm-up step) every candidate plan
# is launched a few times on scratch accumulators, timed with events on the current stream, and the fastest is
# kept for the life of the process.  All plans compute the same sums (in a different association order).
# MOPOE_AUTOTUNE=0 keeps the static heuristic (plan = NULL).
AUTOTUNE = os.environ.get("MOPOE_AUTOTUNE", "1") not in ("0", "table")
# MOPOE_AUTOTUNE=table: the committed plans where the table has the triple, the static heuristic elsewhere, no timing at all
# (the GPU tests of the BASELINE shapes run in this mode: the kernels the product launches, without a tuner pass per test)
TABLE_ONLY = os.environ.get("MOPOE_AUTOTUNE", "1") == "table"
_TUNE_REPS = int(os.environ.get("MOPOE_TUNE_REPS", "3"))
# Optional: leave the first N conv launches of the process on the static heuristic and start tuning afterwards
# (a GPU that has just left idle ranks candidates differently from steady state).  Default 0 = tune at first use,
# so that one warm-up step settles every plan.
TUNE_AFTER_CALLS = int(os.environ.get("MOPOE_TUNE_AFTER_CALLS", "0"))
# Objective of the ranking: duration x (alpha + (1 - alpha) x the fraction of the chip the launch occupies).  1.0 = isolated
# latency alone.  0.7 (default since round 4): inside the step three networks' kernels share the chip, and a plan that is a few
# per cent slower alone but leaves CUs (and LDS) to the other branches wins there -- force-tuned bench.py, same box:
# C2 6 305 / 6 312 (1.0) -> 6 444 / 6 451 / 6 473 (0.7), 6 440 (0.6), 6 342 (0.5), 6 018 (0.3); C5 5 100 -> 5 239; c2d128 2 671 -> 2 707;
# C3 22 677 -> 22 761
TUNE_ALPHA = float(os.environ.get("MOPOE_TUNE_ALPHA", "0.7"))
# tiles 8..11 (LDS-free kernel) among the tuner's candidates: off by default -- they win on many small layers when timed
# alone but the step as a whole (three modalities' kernels running side by side) is not faster with them
DIRECT_TILES = os.environ.get("MOPOE_DIRECT_TILES", "0") != "0"
_conv_calls = 0
_plans = {}
_GATHER_TILES = ((128, 128), (256, 64), (64, 64), (256, 128), (128, 64), (64, 64), (128, 64), (128, 128),
                 (128, 128), (256, 64), (64, 64), (128, 64),   # 8..11: the LDS-free kernel
                 (128, 128), (128, 64), (64, 64), (256, 128),  # 12..15: the LDS-DMA family (csrc/conv_gemm_glds.inc)
                 (128, 128), (128, 64), (64, 64), (256, 128))  # 16..19: the same with the fp32 products on the bf16 matrix pipe
F32_GLDS = os.environ.get("MOPOE_F32_GLDS", "1") != "0"       # A/B switch: keep the tuner off the LDS-DMA tiles
# A/B switch: tiles 16..19 (plain operand) offered.  The products are the same fp32 numbers -- each operand is split exactly
# into three bf16 parts and six of the nine partial products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the error
# against fp64 is lower than that of v_mfma_f32_32x32x2_f32 (csrc/conv_gemm_glds.inc, tests: test_f32_products_on_the_bf16_pipe)
F32_SPLIT_BF16 = os.environ.get("MOPOE_F32_SPLIT_BF16", "1") != "0"
_SPLITS = (2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128)


_forced_plan = None

# Committed launch plans: the tuner's choices for the BASELINE configurations, measured on an MI355X
# by tests/tools/make_plan_table.py and kept in mimic_amd/plans_gfx950.json.  A triple found in the table takes its plan
# from it; the tuner only runs for triples the table does not hold (other shapes / batch sizes), or for everything under
# MOPOE_AUTOTUNE=force.  Two processes therefore launch the same kernels, and a bench set-up does not spend its first
# step timing ~30 candidates per layer.
PLAN_TABLE_PATH = os.environ.get("MOPOE_PLAN_TABLE", os.path.join(os.path.dirname(os.path.abspath(__file__)), "plans_gfx950.json"))
USE_PLAN_TABLE = os.environ.get("MOPOE_AUTOTUNE", "1") != "force" and os.environ.get("MOPOE_PLAN_TABLE", "") != "0"
_plan_table = None


def plan_key_str(key) -> str:
    """the table's key: op name, the 14 geometry numbers, the fusion flags"""
    g = key[1]
    geo = ",".join(str(int(getattr(g, f))) for f in ("N", "Hs", "Ws", "Hb", "Wb", "Cin", "Cout", "kh", "kw", "sh", "sw", "ph", "pw", "transposed"))
    return f"{key[0]}|{geo}|" + ",".join(str(int(f)) if isinstance(f, bool) else str(f) for f in key[2:])


def _table_plan(key):
    """(found, plan or None) from the committed table"""
    global _plan_table
    if _plan_table is None:
        _plan_table = {}
        if USE_PLAN_TABLE and os.path.exists(PLAN_TABLE_PATH):
            import json
            with open(PLAN_TABLE_PATH) as f:
                _plan_table = json.load(f).get("plans", {})
    v = _plan_table.get(plan_key_str(key), False)
    if v is False and key[0] in ("fwd", "fwd16") and len(key) >= 4 and isinstance(key[3], bool):
        # a forward conv with / without a dropout mask in its epilogue (train / eval, or dropout disabled): the table was measured
        # on the training step; the mask multiplies the finished tile and does not move the choice of tile or split
        v = _plan_table.get(plan_key_str(key[:3] + (not key[3],) + key[4:]), False)
    if v is False:
        return False, None
    if v is not None and not F32_SPLIT_BF16:
        # the switch is off: a committed plan on the bf16 matrix pipe falls back to the same tile on the fp32 MFMA
        if key[0] in ("fwd", "dgrad") and v[0] >= 16:
            v = [v[0] - 4, v[1]]
        elif key[0] == "wgrad" and v[0] in (7, 8):
            v = [v[0] - 2, v[1]]
        elif key[0] == "wgrad" and v[0] in (9, 10):
            v = [2, v[1]]
    return True, (None if v is None else _Plan(int(v[0]), int(v[1])))


_plan_sources = {"table": 0, "tuned": 0, "static": 0}


def plan_source_summary():
    """how the launch plans in use were obtained: counts of (op, geometry, fusion) triples from the committed table, from the
    in-process tuner, and left on the library's static heuristic"""
    return dict(_plan_sources, table_file=os.path.basename(PLAN_TABLE_PATH) if USE_PLAN_TABLE and os.path.exists(PLAN_TABLE_PATH) else None)


def clear_plans():
    _plans.clear()


class force_plan:
    """context manager (tests): every conv op inside uses this (tile, split) instead of the tuned table"""

    def __init__(self, tile: int, split: int):
        self.plan = _Plan(tile, split)

    def __enter__(self):
        global _forced_plan
        self.prev, _forced_plan = _forced_plan, self.plan
        return self

    def __exit__(self, *exc):
        global _forced_plan
        _forced_plan = self.prev


_plan_log = {}


def plan_report():
    """[(op, geometry, fusion flags, chosen (tile, split), its us, {candidate: us})] of every tuned triple"""
    out = []
    for key, timings in _plan_log.items():
        p = _plans.get(key)
        chosen = None if p is None else (p.tile, p.split)
        out.append((key[0], key[1], key[2:], chosen, timings.get(chosen), timings))
    return out


def plan_table():
    """{(op, Geom, fusion flags): (tile, split) or None (static heuristic)} -- what the autotuner chose"""
    return {k: (None if v is None else (v.tile, v.split)) for k, v in _plans.items()}


def _gather_shape(kind: str, g: Geom):
    """(rows per phase, phases, taps per phase, K channels, N channels) of the implicit GEMM (conv_gemm.hip)"""
    ck, cn = (g.Cin, g.Cout) if kind == "fwd" else (g.Cout, g.Cin)
    dest_on_small = (kind == "fwd") != bool(g.transposed)
    if dest_on_small:
        return g.N * g.Hs * g.Ws, 1, g.kh * g.kw, ck, cn
    rows = g.N * -(-g.Hb // g.sh) * -(-g.Wb // g.sw)
    return rows, g.sh * g.sw, max(1, (g.kh // g.sh) * (g.kw // g.sw)), ck, cn


WS_COUNTER_BYTES = 64 << 10   # head of the workspace: arrival counters of in-kernel split reductions (header: workspace)


def _gather_candidates(kind: str, g: Geom, ws_bytes: int, plain_operand: bool = False):
    """plain_operand: no BN -> ReLU on the gathered operand (every LDS-DMA tile applies; with it tiles 12, 14 and 15)"""
    if min(g.Cin, g.Cout) == 1:
        return {}   # image-side edge layers run on the streaming edge kernels: nothing to choose
    ws_bytes -= WS_COUNTER_BYTES
    rows, nphase, taps, ck, cn = _gather_shape(kind, g)
    iters = taps * -(-ck // 16)
    per = rows * nphase * cn * 4
    cands = {}   # (tile, split) -> fraction of the chip the launch occupies
    for tile, (bm, bn) in enumerate(_GATHER_TILES):
        blocks = -(-rows // bm) * -(-cn // bn) * nphase
        if bm == 256 and rows < 256:
            continue
        if tile >= 3 and (g.Cin % 4 or g.Cout % 4):
            continue   # vector-path-only tiles
        if tile in (5, 6) and ck % 32:
            continue   # 32-deep K chunk
        if 8 <= tile < 12 and (ck % 8 or not DIRECT_TILES):
            continue   # LDS-free kernel: 8-deep K steps
        if tile >= 12 and not (F32_GLDS and ck % 32 == 0 and g.Cin % 4 == 0 and g.Cout % 4 == 0 and (plain_operand or tile != 13)):
            continue   # LDS-DMA family: 32-deep stages, vector path (with BN on load: the tiles with 2 or 4 buffers)
        if tile >= 16 and not (F32_SPLIT_BF16 and plain_operand):
            continue   # products on the bf16 pipe: plain operand forms
        cap = (256 if tile in (15, 19) else 512) if tile >= 12 else (512 if tile in (0, 1, 3) else 768)   # blocks resident at once
        cands[(tile, 1)] = min(1.0, blocks / cap)
        for s in _SPLITS:
            if s * 2 <= iters and blocks * s <= 2048 and s * per <= ws_bytes:
                cands[(tile, s)] = min(1.0, blocks * s / cap)
    return cands


def _wgrad_candidates(g: Geom, bf16: bool = False, plain_operand: bool = False):
    if min(g.Cin, g.Cout) == 1:
        return {}
    ms = g.N * g.Hs * g.Ws
    cands = {}
    # tile 7 (bf16, plain operand): two taps per block on the gathered operand's side when that side has 64 channels
    if bf16 and BF16_GLDS and plain_operand and g.taps % 2 == 0 and (g.Cout if g.transposed else g.Cin) == 64:
        tiles = -(-(g.Cin if g.transposed else g.Cout) // 128) * (g.taps // 2)
        seen = set()
        for target in (256, 512, 768, 1024, 1536, 2048, 4096):
            s = max(1, min(-(-target // tiles), -(-ms // 128)))
            if s not in seen:
                seen.add(s)
                cands[(7, s)] = min(1.0, tiles * s / 768)
    # tile 8 (bf16, plain operand, k4 s2 p1, small grid of whole 8 x 8 tiles): four taps -- one parity class -- per block
    if (bf16 and BF16_GLDS and WGRAD_PARITY and plain_operand and (g.kh, g.kw, g.sh, g.sw, g.ph, g.pw) == (4, 4, 2, 2, 1, 1)
            and g.Hs % 8 == 0 and g.Ws % 8 == 0 and g.Hb == 2 * g.Hs and g.Wb == 2 * g.Ws):
        cg, csm = (g.Cout, g.Cin) if g.transposed else (g.Cin, g.Cout)
        ntiles = g.N * (g.Hs // 8) * (g.Ws // 8)
        for tile, cs in ((8, 64), (9, 128)):       # 64 gathered channels x 64 / 128 channels of the small-grid operand
            if csm % cs:
                continue
            tiles = -(-cg // 64) * (csm // cs) * 4
            seen = set()
            for target in (256, 512, 768, 1024, 2048):
                s = max(1, min(-(-target // tiles), ntiles))
                if s not in seen:
                    seen.add(s)
                    cands[(tile, s)] = min(1.0, tiles * s / (256 if cs == 128 else 512))
    # fp32 tiles 9 / 10 (plain operand, k4 s2 p1, small grid of whole 8 x 8 tiles): four taps per block with the products on the
    # bf16 matrix pipe (csrc/conv_gemm_glds_parity.inc); 64 gathered channels x 64 / 128 channels of the small-grid operand
    if (not bf16 and F32_GLDS and F32_SPLIT_BF16 and plain_operand and (g.kh, g.kw, g.sh, g.sw, g.ph, g.pw) == (4, 4, 2, 2, 1, 1)
            and g.Hs % 8 == 0 and g.Ws % 8 == 0 and g.Hb == 2 * g.Hs and g.Wb == 2 * g.Ws and g.Cin % 4 == 0 and g.Cout % 4 == 0):
        cg, csm = (g.Cout, g.Cin) if g.transposed else (g.Cin, g.Cout)
        ntiles = g.N * (g.Hs // 8) * (g.Ws // 8)
        for tile, cs in ((9, 64), (10, 128)):
            if csm % cs:
                continue
            tiles = -(-cg // 64) * (csm // cs) * 4
            seen = set()
            # (one block per CU: the largest split that still fits ONE round of 256 blocks is usually the best -- rb2 of
            # config #2, 24 channel-tile x class blocks: split 10 = 240 blocks 92 us, split 8 = 192 blocks 105, split 11 = 264 blocks 148)
            for s in [max(1, min(256 // tiles, ntiles))] + [max(1, min(-(-target // tiles), ntiles)) for target in (64, 128, 192, 256, 384, 512, 1024)]:
                if s not in seen:
                    seen.add(s)
                    cands[(tile, s)] = min(1.0, tiles * s / 256)
    # tiles 5 / 6 = the 128 / 64 tiles on LDS-DMA (csrc/conv_gemm_glds.inc, conv_gemm_bf16_glds.inc)
    glds = (BF16_GLDS if bf16 else F32_GLDS and g.Cin % 4 == 0 and g.Cout % 4 == 0)
    tile_list = ((0, 128), (2, 64), (5, 128), (6, 64)) if glds else ((0, 128), (2, 64))
    if glds and not bf16 and F32_SPLIT_BF16 and plain_operand:
        tile_list += ((7, 128), (8, 64))        # fp32 tiles 5 / 6 with the products on the bf16 matrix pipe
    for tile, tsz in tile_list:
        if tsz == 128 and (g.Cin <= 64 or g.Cout <= 64):
            continue
        tiles = -(-g.Cin // tsz) * -(-g.Cout // tsz) * g.taps
        seen = set()
        for target in (256, 512, 768, 1024, 1536, 2048, 4096):
            s = max(1, min(-(-target // tiles), -(-ms // 128)))
            if s not in seen:
                seen.add(s)
                cands[(tile, s)] = min(1.0, tiles * s / 768)
    return cands


_scratch = {}


def _scratch_like(t):
    """zero-initialised stand-in for an accumulating output while candidates are being timed"""
    if t is None:
        return None
    k = (t.dtype, t.numel(), t.device)
    if k not in _scratch:
        _scratch[k] = torch.zeros(t.numel(), dtype=t.dtype, device=t.device)
    return _scratch[k]


def _tuned_plan(key, cands_fn, launch):
    """launch(plan_ref) enqueues the op once on scratch accumulators -> ctypes byref of the fastest plan (or None)"""
    if _forced_plan is not None:
        return C.byref(_forced_plan)
    if key in _plans:
        p = _plans[key]
        return None if p is None else C.byref(p)
    if not AUTOTUNE and not TABLE_ONLY:
        return None
    found, p = _table_plan(key)
    if found:
        _plans[key] = p
        _plan_sources["table"] += 1
        return None if p is None else C.byref(p)
    if TABLE_ONLY and not AUTOTUNE:
        _plans[key] = None
        _plan_sources["static"] += 1
        return None
    if torch.cuda.is_current_stream_capturing():
        return None
    global _conv_calls
    _conv_calls += 1
    if _conv_calls <= TUNE_AFTER_CALLS:
        return None   # still warming up: static heuristic, nothing cached
    cands = cands_fn()
    best, timings = None, {}
    if len(cands) > 1:
        # candidates are timed alone on the device: work still queued on the other streams (the other modalities'
        # networks, the weight-gradient lane) would otherwise run beside them and decide the ranking
        torch.cuda.synchronize()
        def time_plan(plan, reps):
            ref = C.byref(plan)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                launch(ref)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3

        plans = {c: _Plan(*c) for c in cands}
        # objective: duration x (alpha + (1 - alpha) x fraction of the chip occupied).  alpha = 1 ranks by latency
        # alone; below 1 a launch that leaves CUs free for the other modalities' branches of the step graph is
        # charged less for its duration
        score = lambda c: timings[c] * (TUNE_ALPHA + (1.0 - TUNE_ALPHA) * cands[c])
        for c, plan in plans.items():          # pass 1: everything, briefly (first launch = warm-up)
            launch(C.byref(plan))
            timings[c] = time_plan(plan, _TUNE_REPS)
        finalists = sorted(timings, key=score)[:3]
        for c in finalists:                    # pass 2: the three best, longer; best of three batches each
            timings[c] = min(time_plan(plans[c], 2 * _TUNE_REPS) for _ in range(3))
        best = plans[min(finalists, key=score)]
    _plans[key] = best
    _plan_log[key] = timings
    _plan_sources["tuned" if len(cands) > 1 else "static"] += 1
    return None if best is None else C.byref(best)


BF16 = torch.bfloat16
_GATHER_TILES_BF16 = ((128, 128), (256, 64), (64, 64), (256, 128), (128, 64),
                      # 5..11: the LDS-DMA family (csrc/conv_gemm_bf16_glds.inc): operands without a transform on load, K % 64 == 0
                      (128, 128), (128, 128), (256, 128), (256, 128), (128, 64), (128, 64), (64, 64))
_GLDS_TILES = {5: 512, 6: 256, 7: 256, 9: 768, 10: 512, 11: 512}     # tile -> blocks resident at once (8 spills: not offered)
BF16_GLDS = os.environ.get("MOPOE_BF16_GLDS", "1") != "0"      # A/B switch: keep the tuner on the register-staged tiles
WGRAD_PARITY = os.environ.get("MOPOE_WGRAD_PARITY", "1") != "0"  # A/B switch: wgrad tile 8 (four taps per block) offered


def _is16(t):
    return t is not None and t.dtype == BF16


def _gather_candidates_bf16(kind: str, g: Geom, ws_bytes: int, plain_operand: bool = False):
    """plain_operand: the gathered operand enters the MFMA as it lies in memory (no BN -> ReLU on load): the LDS-DMA tiles apply"""
    ws_bytes -= WS_COUNTER_BYTES
    rows, nphase, taps, ck, cn = _gather_shape(kind, g)
    iters = taps * (ck // 32)
    per = rows * nphase * cn * 4
    cands = {}
    for tile, (bm, bn) in enumerate(_GATHER_TILES_BF16):
        blocks = -(-rows // bm) * -(-cn // bn) * nphase
        if bm == 256 and rows < 256:
            continue
        if tile >= 5 and not (BF16_GLDS and ck % 64 == 0 and tile in _GLDS_TILES and (plain_operand or tile in (5, 7, 9, 11))):
            continue     # (with BN -> ReLU on load: the two-buffer tiles and the four-buffer small one)
        if tile >= 5 and taps * (ck // 64) < 2:
            continue     # a single 64-deep chunk per tile: nothing to pipeline
        cap = _GLDS_TILES[tile] if tile >= 5 else (512 if tile == 3 else 768)
        cands[(tile, 1)] = min(1.0, blocks / cap)
        for sp in _SPLITS:
            if sp * 2 <= iters and blocks * sp <= 2048 and sp * per <= ws_bytes:
                cands[(tile, sp)] = min(1.0, blocks * sp / cap)
    return cands


def conv_mix_supported(x, g: Geom) -> bool:
    """can conv_fwd(..., mix=) take this layer?  (the residual mix lives in the vector path's row-major epilogue)"""
    return _is16(x) or (g.Cin % 4 == 0 and g.Cout % 4 == 0)


def conv_fwd(x, wp, g: Geom, bn_in: Optional[Bn] = None, bias=None, mask: Optional[Mask] = None,
             out_stats=None, out_dtype=None, mix=None):
    """out_dtype: dtype of the result (default: the activation operand's).  bf16 family: x bf16 + wp bf16 (the
    weight copy); the single-channel image-side layers take the fp32 image / fp32 taps and a bf16 wide tensor.
    mix = (s, bn_s[, a, b]): the residual mix of a block in the epilogue, y = a * bn_s(s) + b * mask * (conv + bias)
    (include/mopoe_hip.h: mopoe_conv_fwd_mix); out_stats are then those of y."""
    _dev(x, wp, bias, out_stats)
    assert tuple(x.shape) == g.in_shape and tuple(wp.shape) == (g.taps, g.Cin, g.Cout)
    out_dtype = out_dtype or x.dtype
    gc = g.c()
    stream = _stream()
    mixr = None
    if mix is not None:
        ms, mbn = mix[0], mix[1]
        _dev(ms)
        assert tuple(ms.shape) == g.out_shape and ms.dtype == out_dtype and ms.is_contiguous() and conv_mix_supported(x, g)
        mixc = _MixRef(ms.data_ptr(), mbn.c(), float(mix[2]) if len(mix) > 2 else RES_A, float(mix[3]) if len(mix) > 3 else RES_B)
        mixr = C.byref(mixc)
    if _is16(x) or out_dtype == BF16:
        plain = bn_in is None and mask is None
        if not g.transposed and g.Cin == 1:      # image stem: fp32 pixels x fp32 taps -> bf16 features
            assert plain and bias is None and x.dtype == torch.float32 and wp.dtype == torch.float32
            y = torch.empty(g.out_shape, dtype=BF16, device=x.device)
            _check(lib().mopoe_edge_expand_bf16(_p(x), _p(wp), _p(y), C.byref(gc), g.Cout, _p(out_stats), stream))
            return y
        if g.transposed and g.Cout == 1:         # image head: bf16 features x fp32 taps -> fp32 pixels
            assert plain and out_stats is None and _is16(x) and wp.dtype == torch.float32
            y = torch.empty(g.out_shape, dtype=torch.float32, device=x.device)
            _check(lib().mopoe_edge_reduce_bf16(_p(x), _p(wp), _p(bias), _p(y), C.byref(gc), g.Cin, stream))
            return y
        if not (_is16(x) and _is16(wp)):
            raise MopoeHipError("bf16 conv needs a bf16 activation and the bf16 copy of the packed weight")
        y = torch.empty(g.out_shape, dtype=out_dtype, device=x.device)
        ws, nbytes = _workspace(x.device)
        fn = lib().mopoe_conv_fwd_bf16
        bnr, mr, f32 = _bn(bn_in), _mask(mask), C.c_int32(int(out_dtype == torch.float32))

        def launch16(plan, stats=None):
            if mixr is not None:
                _check(lib().mopoe_conv_fwd_mix_bf16(_p(x), _p(wp), _p(bias), _p(y), C.byref(gc), bnr, mr, mixr, _p(stats),
                                                     plan, _p(ws), C.c_size_t(nbytes), stream))
                return
            _check(fn(_p(x), _p(wp), _p(bias), _p(y), f32, C.byref(gc), bnr, mr, _p(stats), plan, _p(ws),
                      C.c_size_t(nbytes), stream))

        key = ("fwd16", g, bn_in is not None, mask is not None, out_stats is not None, out_dtype) + (("mix",) if mix is not None else ())
        plan = _tuned_plan(key, lambda: _gather_candidates_bf16("fwd", g, nbytes, plain_operand=bn_in is None),
                           lambda ref: launch16(ref, _scratch_like(out_stats)))
        launch16(plan, out_stats)
        return y
    y = torch.empty(g.out_shape, dtype=torch.float32, device=x.device)
    ws, nbytes = _workspace(x.device)
    fn = lib().mopoe_conv_fwd
    bnr, mr = _bn(bn_in), _mask(mask)

    def launch(plan, stats=None):
        if mixr is not None:
            _check(lib().mopoe_conv_fwd_mix(_p(x), _p(wp), _p(bias), _p(y), C.byref(gc), bnr, mr, mixr, _p(stats), plan,
                                            _p(ws), C.c_size_t(nbytes), stream))
            return
        _check(fn(_p(x), _p(wp), _p(bias), _p(y), C.byref(gc), bnr, mr, _p(stats), plan, _p(ws), C.c_size_t(nbytes),
                  stream))

    key = ("fwd", g, bn_in is not None, mask is not None, out_stats is not None) + (("mix",) if mix is not None else ())
    plan = _tuned_plan(key, lambda: _gather_candidates("fwd", g, nbytes, plain_operand=bn_in is None),
                       lambda ref: launch(ref, _scratch_like(out_stats)))
    launch(plan, out_stats)
    return y


def conv_dgrad(dy, wp, g: Geom, relu_bn: Optional[Bn] = None, xin=None, bwd_sums=None, out_dtype=None):
    _dev(dy, wp, xin, bwd_sums)
    assert tuple(dy.shape) == g.out_shape
    out_dtype = out_dtype or dy.dtype
    gc = g.c()
    stream = _stream()
    if _is16(dy) or out_dtype == BF16:
        if g.transposed and g.Cout == 1:         # image head: fp32 pixel gradient x fp32 taps -> bf16 feature gradient
            assert relu_bn is None and dy.dtype == torch.float32 and wp.dtype == torch.float32
            dx = torch.empty(g.in_shape, dtype=BF16, device=dy.device)
            _check(lib().mopoe_edge_expand_bf16(_p(dy), _p(wp), _p(dx), C.byref(gc), g.Cin, None, stream))
            return dx
        if not (_is16(dy) and _is16(wp)) or (xin is not None and not _is16(xin)):
            raise MopoeHipError("bf16 conv_dgrad needs bf16 dy / xin and the bf16 copy of the packed weight")
        dx = torch.empty(g.in_shape, dtype=out_dtype, device=dy.device)
        ws, nbytes = _workspace(dy.device)
        fn = lib().mopoe_conv_dgrad_bf16
        bnr, f32 = _bn(relu_bn), C.c_int32(int(out_dtype == torch.float32))

        def launch16(plan, sums=None):
            _check(fn(_p(dy), _p(wp), _p(dx), f32, C.byref(gc), bnr, _p(xin), _p(sums), plan, _p(ws), C.c_size_t(nbytes),
                      stream))

        key = ("dgrad16", g, relu_bn is not None, bwd_sums is not None, out_dtype)
        plan = _tuned_plan(key, lambda: _gather_candidates_bf16("dgrad", g, nbytes, plain_operand=True),
                           lambda ref: launch16(ref, _scratch_like(bwd_sums)))
        launch16(plan, bwd_sums)
        return dx
    dx = torch.empty(g.in_shape, dtype=torch.float32, device=dy.device)
    ws, nbytes = _workspace(dy.device)
    fn = lib().mopoe_conv_dgrad
    bnr = _bn(relu_bn)

    def launch(plan, sums=None):
        _check(fn(_p(dy), _p(wp), _p(dx), C.byref(gc), bnr, _p(xin), _p(sums), plan, _p(ws), C.c_size_t(nbytes), stream))

    key = ("dgrad", g, relu_bn is not None, bwd_sums is not None)
    plan = _tuned_plan(key, lambda: _gather_candidates("dgrad", g, nbytes, plain_operand=True),
                       lambda ref: launch(ref, _scratch_like(bwd_sums)))
    launch(plan, bwd_sums)
    return dx


def conv_wgrad(x, dy, g: Geom, bn_in: Optional[Bn] = None, out=None):
    """out: optional ZERO-FILLED [taps, Cin, Cout] destination (e.g. a slice of a per-network gradient arena).
    The result is fp32 for either family (it feeds Adam on the fp32 master weights)."""
    _dev(x, dy, out)
    assert tuple(x.shape) == g.in_shape and tuple(dy.shape) == g.out_shape
    dwp = out if out is not None else torch.empty((g.taps, g.Cin, g.Cout), dtype=torch.float32, device=x.device)
    assert tuple(dwp.shape) == (g.taps, g.Cin, g.Cout) and dwp.dtype == torch.float32
    gc = g.c()
    stream = _stream()
    bnr = _bn(bn_in)
    if _is16(x) or _is16(dy):
        if min(g.Cin, g.Cout) == 1:              # image stem / head: wide tensor bf16, image side fp32
            assert bn_in is None
            vec, scal = (dy, x) if g.Cin == 1 else (x, dy)
            assert _is16(vec) and scal.dtype == torch.float32
            _check(lib().mopoe_edge_wgrad_bf16(_p(vec), _p(scal), _p(dwp), C.byref(gc), max(g.Cin, g.Cout),
                                               C.c_int32(int(out is not None)), stream))
            return dwp
        if not (_is16(x) and _is16(dy)):
            raise MopoeHipError("bf16 conv_wgrad needs both operands in bf16")
        fn = lib().mopoe_conv_wgrad_bf16
        key = ("wgrad16", g, bn_in is not None)
    else:
        fn = lib().mopoe_conv_wgrad
        key = ("wgrad", g, bn_in is not None)

    def launch(plan, dst, is_zero):
        _check(fn(_p(x), _p(dy), _p(dst), C.byref(gc), bnr, C.c_int32(is_zero), plan, stream))

    plan = _tuned_plan(key, lambda: _wgrad_candidates(g, bf16=key[0] == "wgrad16", plain_operand=bn_in is None),
                       lambda ref: launch(ref, _scratch_like(dwp), 1))
    launch(plan, dwp, int(out is not None))
    return dwp


# ----------------------------------------------------------------------------------------------
# residual-block glue
# ----------------------------------------------------------------------------------------------
def _rows(t):
    return t.numel() // t.shape[-1]


def block_out_fwd(s, m, bn_s: Bn, a=RES_A, b=RES_B, out_stats=None):
    _dev(s, m, out_stats)
    out = torch.empty_like(s)
    fn = lib().mopoe_block_out_fwd_bf16 if _is16(s) else lib().mopoe_block_out_fwd
    _check(fn(_p(s), _p(m), _p(out), C.c_int64(_rows(s)), s.shape[-1], _bn(bn_s),
                                     C.c_float(a), C.c_float(b), _p(out_stats), _stream()))
    return out


def bn_relu_apply(x, bn: Bn):
    """relu(bn(x)) written out (the operand of a block's second conv, where that beats the conv's BN -> ReLU-on-load form)"""
    _dev(x)
    out = torch.empty_like(x)
    fn = lib().mopoe_bn_relu_apply_bf16 if _is16(x) else lib().mopoe_bn_relu_apply
    _check(fn(_p(x), _p(out), C.c_int64(_rows(x)), x.shape[-1], _bn(bn), _stream()))
    return out


# ---- the front of a residual block as streaming kernels (csrc/pointwise.hip): bn1 -> relu -> conv1 (1x1) -> dropout -> bn2 -> relu
BLOCK_FRONT = os.environ.get("MOPOE_BLOCK_FRONT", "1") != "0"     # A/B switch
BLOCK_FRONT_F32 = os.environ.get("MOPOE_BLOCK_FRONT_F32", "1") != "0"     # A/B switch for the fp32 family alone
# fp32: the two forward passes each pay conv1's 64 MFMAs per 32 pixels at the fp32 rate (one MFMA = 64 cycles for 2 k) and come
# out slower than conv_fwd + bn_relu_apply (rb1 at config #2: 101.9 against 91.7 us); the fused backward wins (100.0 against
# 129.9 us).  So the fp32 family keeps the forward of round 3 (d1 written) and takes only the backward kernel.
BLOCK_FRONT_F32_FWD = os.environ.get("MOPOE_BLOCK_FRONT_F32_FWD", "0") != "0"


def block_front_supported(x, g1: "Geom", mask1: Optional[Mask], forward: bool = False) -> bool:
    """64 channels (either storage family), a 1x1 conv, dropout absent or per (sample, channel) on whole 32-row tiles.
    forward=True: is the FORWARD pair (statistics + apply passes, d1 never written) to be used as well?"""
    if forward and not _is16(x) and not BLOCK_FRONT_F32_FWD:
        return False
    return (BLOCK_FRONT and x.dtype in (BF16, torch.float32) and (_is16(x) or BLOCK_FRONT_F32) and g1.Cin == 64 and g1.Cout == 64
            and g1.taps == 1 and _rows(x) % 32 == 0
            and (mask1 is None or (mask1.kind == 1 and mask1.rows_per_sample % 32 == 0)))


def block_front_stats(x, w1, bias, bn1: Bn, mask1: Optional[Mask], out_stats):
    """out_stats [2, C] += {sum, sumsq} of d1 = mask1 * (conv1(relu(bn1(x))) + bias), rounded to bf16; d1 itself is not written"""
    _dev(x, w1, bias, out_stats)
    fn = lib().mopoe_block_front_stats_bf16 if _is16(x) else lib().mopoe_block_front_stats
    _check(fn(_p(x), _p(w1), _p(bias), C.c_int64(_rows(x)), x.shape[-1], _bn(bn1), _mask(mask1),
                                              _p(out_stats), _stream()))
    return out_stats


def block_front_apply(x, w1, bias, bn1: Bn, bn2: Bn, mask1: Optional[Mask]):
    """a2 = relu(bn2(d1)) with d1 recomputed from x"""
    _dev(x, w1, bias)
    a2 = torch.empty_like(x)
    fn = lib().mopoe_block_front_apply_bf16 if _is16(x) else lib().mopoe_block_front_apply
    _check(fn(_p(x), _p(w1), _p(bias), _p(a2), C.c_int64(_rows(x)), x.shape[-1], _bn(bn1), _bn(bn2),
                                              _mask(mask1), _stream()))
    return a2


def block_front_bwd(x, dh2, w1, bias, bn1: Bn, bn2: Bn, mask1: Optional[Mask], sums2, sums1, dw1, dbias=None, dgamma2=None,
                    dbeta2=None):
    """dh2 (conv2's input gradient, ReLU mask of bn2 applied) + sums2 -> dh1 (returned), sums1 [2, C] +=, dw1 [1, C, C] +=,
    dbias [C] += (pre-zeroed accumulators); dgamma2 / dbeta2 [C] = bn2's affine gradients (= sums2[1] / sums2[0])"""
    _dev(x, dh2, w1, bias, sums2, sums1, dw1, dbias, dgamma2, dbeta2)
    dh1 = torch.empty_like(x)
    fn = lib().mopoe_block_front_bwd_bf16 if _is16(x) else lib().mopoe_block_front_bwd
    _check(fn(_p(x), _p(dh2), _p(w1), _p(bias), _p(dh1), C.c_int64(_rows(x)), x.shape[-1], _bn(bn1),
                                            _bn(bn2), _mask(mask1), _p(sums2), _p(sums1), _p(dw1), _p(dbias), _p(dgamma2), _p(dbeta2),
                                            _stream()))
    return dh1


def bn_bwd_reduce(g, s, bn_s: Bn, sums=None):
    """sums: optional pre-zeroed double [2, C] (callers batch these allocations)."""
    _dev(g, s, sums)
    if sums is None:
        sums = new_stats(s.shape[-1], s.device)
    fn = lib().mopoe_bn_bwd_reduce_bf16 if _is16(s) else lib().mopoe_bn_bwd_reduce
    _check(fn(_p(g), _p(s), C.c_int64(_rows(s)), s.shape[-1], _bn(bn_s), _p(sums), _stream()))
    return sums


def block_out_bwd(g, s, bn_s: Bn, sums, mask: Optional[Mask], a=RES_A, b=RES_B, want_colsum_dm=False,
                  want_colsum_ds=True, small=None):
    """-> dm, ds, dgamma_s, dbeta_s, colsum_dm (or None), colsum_ds (or None)
    small: optional pre-zeroed float [4, C] that receives dgamma, dbeta and the two column sums."""
    _dev(g, s, sums)
    c = s.shape[-1]
    dm, ds = torch.empty_like(g), torch.empty_like(g)
    if small is None:
        small = torch.zeros(4, c, dtype=torch.float32, device=g.device)
    cdm = small[2] if want_colsum_dm else None
    cds = small[3] if want_colsum_ds else None
    fn = lib().mopoe_block_out_bwd_bf16 if _is16(s) else lib().mopoe_block_out_bwd
    _check(fn(_p(g), _p(s), _p(dm), _p(ds), C.c_int64(_rows(s)), c, _bn(bn_s), _p(sums), _mask(mask), C.c_float(a),
              C.c_float(b), _p(small[0]), _p(small[1]), _p(cdm), _p(cds), _stream()))
    return dm, ds, small[0], small[1], cdm, cds


def bn_bwd_apply(dy, x, bn: Bn, sums, mask: Optional[Mask] = None, add=None, want_colsum=False, small=None,
                 next_s=None, next_bn: Optional[Bn] = None, next_sums=None):
    """-> dx, dgamma, dbeta, colsum_dx (or None); small: optional pre-zeroed float [3, C].
    next_s / next_bn / next_sums: also accumulate bn_bwd_reduce(dx, next_s, next_bn) into the pre-zeroed next_sums."""
    _dev(dy, x, sums, add, next_s, next_sums)
    c = x.shape[-1]
    dx = torch.empty_like(x)
    if small is None:
        small = torch.zeros(3, c, dtype=torch.float32, device=x.device)
    cs = small[2] if want_colsum else None
    nb = _bn(next_bn) if next_s is not None else None
    fn = lib().mopoe_bn_bwd_apply_bf16 if _is16(x) else lib().mopoe_bn_bwd_apply
    _check(fn(_p(dy), _p(x), _p(add), _p(dx), C.c_int64(_rows(x)), c, _bn(bn), _p(sums), _mask(mask), _p(small[0]),
              _p(small[1]), _p(cs), _p(next_s), nb, _p(next_sums), _stream()))
    return dx, small[0], small[1], cs


def bn_running_update(entries: Sequence, momentum=0.1):
    """entries: iterable of (sums double[2,C], running_mean, running_var, count).  One launch per 32 layers; the
    records travel in the kernel arguments, so there is no device-side table to build or keep alive."""
    entries = list(entries)
    if not entries:
        return
    arr = (_RunDesc * len(entries))()
    for i, (sums, rm, rv, count) in enumerate(entries):
        _dev(sums, rm, rv)
        arr[i] = _RunDesc(sums.data_ptr(), rm.data_ptr(), rv.data_ptr(), rm.numel(), count)
    _check(lib().mopoe_bn_running_update(arr, len(entries), C.c_float(momentum), _stream()))


# timeline stamps (tuning aid, tests/tools/net_timeline.py): when STAMPS is a dict, stamp(name) appends a device timestamp
# taken when the current stream reaches this point; the launches are captured like any other node of the step
STAMPS = None


def stamp(name):
    st = STAMPS
    if st is None:
        return
    if "buf" not in st:
        st["buf"] = torch.zeros(256, dtype=torch.int64, device="cuda")
        st["names"] = []
    i = len(st["names"])
    if i >= st["buf"].numel():
        return
    st["names"].append(name)
    _check(lib().mopoe_prof_stamp(C.c_void_p(st["buf"].data_ptr() + 8 * i), _stream()))


def adam_step(params, grads, ms, vs, step, lr, beta1, beta2, eps, coef, lowp=None, prep=True):
    """One Adam step over all tensors (header: mopoe_adam_step).  params / grads / ms / vs: equally long lists of fp32
    tensors (a grad may be None: that tensor is skipped, like optim.Adam does); step: device scalar (float), incremented;
    lr: float or device scalar; coef: float[2] device scratch; lowp: optional list of bf16 copies (or None entries)
    rewritten in the same pass."""
    n = len(params)
    arr = (_AdamSeg * n)()
    for i in range(n):
        p_, g_ = params[i], grads[i]
        if g_ is None:
            arr[i] = _AdamSeg(p_.data_ptr(), None, ms[i].data_ptr(), vs[i].data_ptr(), None, p_.numel())
            continue
        if g_.dtype != torch.float32 or p_.dtype != torch.float32 or g_.numel() != p_.numel():
            raise MopoeHipError("adam_step: parameters and gradients must be fp32 tensors of equal size")
        _dev(p_, g_, ms[i], vs[i])
        lp = lowp[i] if lowp is not None else None
        arr[i] = _AdamSeg(p_.data_ptr(), g_.data_ptr(), ms[i].data_ptr(), vs[i].data_ptr(),
                          None if lp is None else lp.data_ptr(), p_.numel())
    lr_dev = lr if isinstance(lr, torch.Tensor) else None
    _check(lib().mopoe_adam_step(arr, n, _p(step) if prep else None, _p(lr_dev), C.c_double(0.0 if lr_dev is not None else float(lr)),
                                 C.c_double(beta1), C.c_double(beta2), C.c_double(eps), _p(coef), _stream()))


def colsum(x, out=None):
    """out: optional ZERO-FILLED [C] fp32 destination (a slice of a per-network gradient arena: no memset node)"""
    _dev(x, out)
    zero = out is not None
    if out is None:
        out = torch.empty(x.shape[-1], dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.numel() == x.shape[-1]
    fn = lib().mopoe_colsum_bf16 if _is16(x) else lib().mopoe_colsum
    _check(fn(_p(x), _p(out), C.c_int64(_rows(x)), x.shape[-1], C.c_int32(int(zero)), _stream()))
    return out


# ----------------------------------------------------------------------------------------------
# latent space
# ----------------------------------------------------------------------------------------------
def _ptr3(ts):
    arr = (C.c_void_p * 3)()
    for i, t in enumerate(ts):
        arr[i] = None if t is None else t.data_ptr()
    return arr


_kl_ws = {}


def _ws(device, n=8):
    key = (device.index, n, _stream_ptr(device) if device.type == "cuda" else 0)
    if key not in _kl_ws:
        _kl_ws[key] = torch.zeros(n, dtype=torch.float64, device=device)
    return _kl_ws[key]


def latent_fwd(mu_in, lv_in, eps, row_start, w, norm):
    """mu_in/lv_in: 3-lists (PA, Lateral, text) of [B,D] or None.
    -> mus [K,B,D], lvs [K,B,D], joint_mu, joint_lv, z, klds [K], joint_div [1]"""
    present = [t for t in mu_in if t is not None]
    _dev(*present, *[t for t in lv_in if t is not None], eps)
    b, d = present[0].shape
    k = len(row_start) - 1
    dev = present[0].device
    mus = torch.empty(k, b, d, dtype=torch.float32, device=dev)
    lvs = torch.empty_like(mus)
    jm, jl, z = (torch.empty(b, d, dtype=torch.float32, device=dev) for _ in range(3))
    klds = torch.empty(k, dtype=torch.float32, device=dev)
    jd = torch.empty(1, dtype=torch.float32, device=dev)
    rs = (C.c_int32 * (k + 1))(*row_start)
    wa = (C.c_float * k)(*w)
    _check(lib().mopoe_latent_fwd(_ptr3(mu_in), _ptr3(lv_in), _p(eps), b, d, rs, wa, C.c_float(norm), _p(mus),
                                  _p(lvs), _p(jm), _p(jl), _p(z), _p(klds), _p(jd), _p(_ws(dev)),
                                  _stream()))
    return mus, lvs, jm, jl, z, klds, jd


def latent_bwd(mu_in, lv_in, eps, row_start, w, norm, g_mus, g_lvs, g_jm, g_jl, g_z, g_klds, g_jd):
    """-> (d_mu_in[3], d_lv_in[3]) with None for absent modalities."""
    present = [t for t in mu_in if t is not None]
    _dev(*present, eps, *[t for t in (g_mus, g_lvs, g_jm, g_jl, g_z, g_klds, g_jd) if t is not None])
    b, d = present[0].shape
    dmu = [None if t is None else torch.empty_like(t) for t in mu_in]
    dlv = [None if t is None else torch.empty_like(t) for t in lv_in]
    k = len(row_start) - 1
    rs = (C.c_int32 * (k + 1))(*row_start)
    wa = (C.c_float * k)(*w)
    _check(lib().mopoe_latent_bwd(_ptr3(mu_in), _ptr3(lv_in), _p(eps), b, d, rs, wa, C.c_float(norm), _p(g_mus),
                                  _p(g_lvs), _p(g_jm), _p(g_jl), _p(g_z), _p(g_klds), _p(g_jd), _ptr3(dmu),
                                  _ptr3(dlv), _stream()))
    return dmu, dlv


# ----------------------------------------------------------------------------------------------
# likelihoods, embedding
# ----------------------------------------------------------------------------------------------
def laplace_nll_fwd(x_hat, x, scale, norm):
    _dev(x_hat, x)
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _check(lib().mopoe_laplace_nll_fwd(_p(x_hat), _p(x), C.c_int64(x.numel()), C.c_float(scale), C.c_float(norm),
                                       _p(out), _p(_ws(x.device)), _stream()))
    return out


def laplace_nll_bwd(x_hat, x, g, scale, norm):
    _dev(x_hat, x, g)
    dx = torch.empty_like(x_hat)
    _check(lib().mopoe_laplace_nll_bwd(_p(x_hat), _p(x), _p(g), C.c_int64(x.numel()), C.c_float(scale),
                                       C.c_float(norm), _p(dx), _stream()))
    return dx


def logsoftmax_fwd(x, inplace=False):
    _dev(x)
    y = x if inplace else torch.empty_like(x)
    _check(lib().mopoe_logsoftmax_fwd(_p(x), _p(y), C.c_int64(_rows(x)), x.shape[-1], _stream()))
    return y


def logsoftmax_bwd(dy, y, inplace=False, out_dtype=None):
    """out_dtype=bfloat16: the gradient of the logits leaves as a bf16 GEMM operand (dy, y are fp32)"""
    _dev(dy, y)
    if out_dtype == BF16:
        dx = torch.empty(dy.shape, dtype=BF16, device=dy.device)
        _check(lib().mopoe_logsoftmax_bwd_bf16out(_p(dy), _p(y), _p(dx), C.c_int64(_rows(y)), y.shape[-1], _stream()))
        return dx
    dx = dy if inplace else torch.empty_like(dy)
    _check(lib().mopoe_logsoftmax_bwd(_p(dy), _p(y), _p(dx), C.c_int64(_rows(y)), y.shape[-1], _stream()))
    return dx


def token_softmax_grad(logp, ids, g, norm, out_dtype=None):
    """gradient of the logits for loss = -sum_r logp[r, ids[r]] / norm with upstream gradient g (1 element), one pass"""
    _dev(logp, ids, g)
    out_dtype = out_dtype or torch.float32
    dx = torch.empty(logp.shape, dtype=out_dtype, device=logp.device)
    _check(lib().mopoe_token_softmax_grad(_p(logp), _p(ids), _p(g), C.c_int64(ids.numel()), logp.shape[-1], C.c_float(norm),
                                          _p(dx), C.c_int32(int(out_dtype == BF16)), _stream()))
    return dx


def lse_rows(logits):
    """logits [..., V] (fp32 or bf16, contiguous, V a multiple of 4 / 8) -> log-sum-exp per row, fp32 [...]"""
    _dev(logits)
    assert logits.is_contiguous()
    lse = torch.empty(logits.shape[:-1], dtype=torch.float32, device=logits.device)
    _check(lib().mopoe_lse_rows(_p(logits), C.c_int32(int(logits.dtype == BF16)), C.c_int64(_rows(logits)), logits.shape[-1],
                                _p(lse), _stream()))
    return lse


def token_nll_logits_fwd(logits, lse, ids, norm):
    """sum_r (lse[r] - logits[r, ids[r]]) / norm: the token NLL from the head's logits and their row log-sum-exp"""
    _dev(logits, lse, ids)
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    _check(lib().mopoe_token_nll_logits_fwd(_p(logits), C.c_int32(int(logits.dtype == BF16)), _p(lse), _p(ids),
                                            C.c_int64(ids.numel()), logits.shape[-1], C.c_float(norm), _p(out),
                                            _p(_ws(logits.device)), _stream()))
    return out


def token_softmax_grad_logits(logits, lse, ids, g, norm, inplace=False):
    """gradient of the logits for the token NLL from the logits themselves: g / norm * (softmax - onehot), in logits' dtype"""
    _dev(logits, lse, ids, g)
    dx = logits if inplace else torch.empty_like(logits)
    _check(lib().mopoe_token_softmax_grad_logits(_p(logits), C.c_int32(int(logits.dtype == BF16)), _p(lse), _p(ids), _p(g),
                                                 C.c_int64(ids.numel()), logits.shape[-1], C.c_float(norm), _p(dx), _stream()))
    return dx


def token_nll_fwd(logp, ids, norm):
    _dev(logp, ids)
    out = torch.empty(1, dtype=torch.float32, device=logp.device)
    _check(lib().mopoe_token_nll_fwd(_p(logp), _p(ids), C.c_int64(ids.numel()), logp.shape[-1], C.c_float(norm),
                                     _p(out), _p(_ws(logp.device)), _stream()))
    return out


def token_nll_bwd(ids, g, shape, norm):
    _dev(ids, g)
    dlogp = torch.empty(shape, dtype=torch.float32, device=ids.device)
    _check(lib().mopoe_token_nll_bwd(_p(ids), _p(g), C.c_int64(ids.numel()), shape[-1], C.c_float(norm),
                                     _p(dlogp), _stream()))
    return dlogp


def dense_nll_fwd(logp, target, norm):
    """-sum(target * logp) / norm (char text encoding: dense / one-hot targets)"""
    _dev(logp, target)
    assert logp.shape == target.shape
    out = torch.empty(1, dtype=torch.float32, device=logp.device)
    _check(lib().mopoe_dense_nll_fwd(_p(logp), _p(target), C.c_int64(logp.numel()), C.c_float(norm), _p(out),
                                     _p(_ws(logp.device)), _stream()))
    return out


def dense_nll_bwd(target, g, norm):
    _dev(target, g)
    dlogp = torch.empty_like(target)
    _check(lib().mopoe_dense_nll_bwd(_p(target), _p(g), C.c_int64(target.numel()), C.c_float(norm), _p(dlogp), _stream()))
    return dlogp


def laplace_logprob_rows(x_hat, target, scale: float):
    """x_hat [R, ...], target [B, ...] with R a multiple of B -> float [R]: row r scored against target row r % B"""
    _dev(x_hat, target)
    rows, tb = x_hat.shape[0], target.shape[0]
    per_row = x_hat.numel() // rows
    assert target.numel() // tb == per_row and rows % tb == 0
    out = torch.empty(rows, dtype=torch.float32, device=x_hat.device)
    _check(lib().mopoe_laplace_logprob_rows(_p(x_hat), _p(target), C.c_int64(rows), C.c_int64(per_row), C.c_int64(tb),
                                            C.c_float(scale), _p(out), _stream()))
    return out


def token_logprob_rows(logp, ids):
    """logp [R, L, V] log-probabilities, ids [B, L] float token ids, R a multiple of B -> float [R]"""
    _dev(logp, ids)
    rows, L, V = logp.shape
    tb = ids.shape[0]
    assert ids.shape[1] == L and rows % tb == 0
    out = torch.empty(rows, dtype=torch.float32, device=logp.device)
    _check(lib().mopoe_token_logprob_rows(_p(logp), _p(ids), C.c_int64(rows), L, V, C.c_int64(tb), _p(out), _stream()))
    return out


def dense_logprob_rows(logp, target):
    """logp [R, L, F] log-probabilities, target [B, L, F] dense / one-hot, R a multiple of B -> float [R]"""
    _dev(logp, target)
    rows, tb = logp.shape[0], target.shape[0]
    per_row = logp[0].numel()
    assert target[0].numel() == per_row and rows % tb == 0
    out = torch.empty(rows, dtype=torch.float32, device=logp.device)
    _check(lib().mopoe_dense_logprob_rows(_p(logp), _p(target), C.c_int64(rows), C.c_int64(per_row), C.c_int64(tb), _p(out), _stream()))
    return out


def embedding_fwd(ids, table, out_dtype=None):
    _dev(ids, table)
    out_dtype = out_dtype or torch.float32
    out = torch.empty(*ids.shape, table.shape[1], dtype=out_dtype, device=table.device)
    fn = lib().mopoe_embedding_fwd_bf16 if out_dtype == BF16 else lib().mopoe_embedding_fwd
    _check(fn(_p(ids), _p(table), _p(out), C.c_int64(ids.numel()), table.shape[0], table.shape[1], _stream()))
    return out


def embedding_bwd(ids, gout, vocab, padding_idx=0):
    _dev(ids, gout)
    dtable = torch.empty(vocab, gout.shape[-1], dtype=torch.float32, device=gout.device)
    fn = lib().mopoe_embedding_bwd_bf16 if _is16(gout) else lib().mopoe_embedding_bwd
    _check(fn(_p(ids), _p(gout), _p(dtable), C.c_int64(ids.numel()), vocab, gout.shape[-1], padding_idx, _stream()))
    return dtable


# ----------------------------------------------------------------------------------------------
# profiling hooks used by bench.py
# ----------------------------------------------------------------------------------------------
def prof_enable(on: bool):
    _check(lib().mopoe_prof_enable(int(on)))


_TILE_TEMPLATES = ("128, 128, 2, 4, 16", "256, 64, 4, 2, 16", "64, 64, 2, 2, 16", "256, 128, 4, 2, 16", "128, 64, 2, 2, 16",
                   "64, 64, 2, 2, 32", "128, 64, 2, 2, 32", "128, 128, 2, 2, 16")


def _prof_kind_names():
    """kind index -> the kernel's template name exactly as rocprofv3 prints it (header: MOPOE_PROF_KINDS)"""
    names = [None] * 144
    names[140], names[141] = "wgrad_parity_f32_kernel<64, true>", "wgrad_parity_f32_kernel<64, false>"
    names[142], names[143] = "wgrad_parity_f32_kernel<128, true>", "wgrad_parity_f32_kernel<128, false>"
    names[138], names[139] = "wgrad_gemm_f32_glds_kernel<128, false, 2, 1>", "wgrad_gemm_f32_glds_kernel<64, false, 4, 1>"
    for i, tt in enumerate(("128, 128, 2, 2, {}, 2, 1", "128, 64, 2, 2, {}, 3, 1", "64, 64, 2, 2, {}, 4, 1", "256, 128, 4, 2, {}, 3, 1")):
        for k, spec in enumerate((1, 3)):
            names[130 + 2 * i + k] = f"gather_gemm_f32_glds_kernel<{tt.format(spec)}>"
    names[127], names[128], names[129] = "pw_front_fwd_f32_kernel<false>", "pw_front_fwd_f32_kernel<true>", "pw_front_bwd_f32_kernel"
    names[124], names[125], names[126] = "pw_front_fwd_bf16_kernel<64, false>", "pw_front_fwd_bf16_kernel<64, true>", "pw_front_bwd_bf16_kernel<64>"
    names[122], names[123] = "wgrad_parity_bf16_kernel<64, *>", "wgrad_parity_bf16_kernel<128, *>"
    for m in (1, 2):
        names[120 + m - 1] = f"wgrad_gemm_bf16_glds_kernel<128, false, true, {m}>"
    for i, (tt, st) in enumerate((("128", (2, 2)), ("64", (4, 2)))):
        for xf in (0, 1):
            names[116 + 2 * i + xf] = f"wgrad_gemm_f32_glds_kernel<{tt}, {'true' if xf else 'false'}, {st[xf]}, 0>"
    for i, tt in enumerate(("128, 128, 2, 2, {}, 2, 0", "128, 64, 2, 2, {}, 3, 0", "64, 64, 2, 2, {}, 4, 0", "256, 128, 4, 2, {}, 2, 0")):
        for spec in (1, 2, 3):
            names[104 + 3 * i + spec - 1] = f"gather_gemm_f32_glds_kernel<{tt.format(spec)}>"
    for i, tt in enumerate(("128", "64")):
        for xf in (0, 1):
            names[100 + 2 * i + xf] = f"wgrad_gemm_bf16_glds_kernel<{tt}, {'true' if xf else 'false'}, true, 0>"
    for i, tt in enumerate(("128, 128, 2, 2, 2, 2", "256, 128, 4, 2, 2, 2", "128, 64, 4, 1, 2, 2")):
        names[94 + i] = f"gather_gemm_bf16_glds_kernel<{tt}>"
    names[98] = "gather_gemm_bf16_glds_kernel<64, 64, 2, 2, 2, 4>"
    for i, tt in enumerate(("128, 128, 2, 2, {}, 2", "128, 128, 2, 2, {}, 3", "256, 128, 4, 2, {}, 2", "256, 128, 4, 2, {}, 3",
                            "128, 64, 4, 1, {}, 2", "128, 64, 4, 1, {}, 3", "64, 64, 2, 2, {}, 4")):
        for k, spec in enumerate((1, 3)):
            names[80 + 2 * i + k] = f"gather_gemm_bf16_glds_kernel<{tt.format(spec)}>"
    for tile, tt in enumerate(("128, 128, 2, 2", "256, 64, 4, 1", "64, 64, 2, 2", "256, 128, 4, 2", "128, 64, 4, 1")):
        for spec in (1, 2, 3):
            names[60 + tile * 3 + spec - 1] = f"gather_gemm_bf16_kernel<{tt}, {spec}>"
    for i, tt in enumerate(("128, 128", "64, 64")):
        for xf in (0, 1):
            names[75 + 2 * i + xf] = f"wgrad_gemm_bf16_kernel<{tt}, {'true' if xf else 'false'}, true>"
    for tile, tt in enumerate(_TILE_TEMPLATES):
        for spec in range(4):
            names[tile * 4 + spec] = f"gather_gemm_kernel<{tt}, true, {spec}>"
    for i, tt in enumerate(("128, 128, 2, 2, 16", "256, 64, 4, 1, 16", "64, 64, 2, 2, 16")):
        names[32 + i] = f"gather_gemm_kernel<{tt}, false, 0>"
    for i, tt in enumerate(("128, 128", "64, 64")):
        for spec in range(3):
            names[36 + 3 * i + spec] = f"wgrad_gemm_kernel<{tt}, true, {spec}>"
        names[42 + i] = f"wgrad_gemm_kernel<{tt}, false, 0>"
    for i, tt in enumerate(("2, 2, 2, 2", "4, 1, 2, 2", "2, 2, 1, 1", "2, 1, 2, 2")):
        for spec in range(1, 4):
            names[44 + 4 * i + spec] = f"direct_gemm_kernel<{tt}, {spec}>"
    return names


PROF_KINDS = _prof_kind_names()


def prof_collect():
    """-> {kernel kind: (launches, total ms, total algorithmic flops, total algorithmic bytes)} since the last collect"""
    k = len(PROF_KINDS)
    n, ms, fl, by = (C.c_int64 * k)(), (C.c_double * k)(), (C.c_double * k)(), (C.c_double * k)()
    _check(lib().mopoe_prof_collect(n, ms, fl, by))
    return {name: (n[i], ms[i], fl[i], by[i]) for i, name in enumerate(PROF_KINDS) if name is not None and n[i] > 0}
