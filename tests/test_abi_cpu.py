"""The C-ABI library loads without a GPU and exports every symbol include/mopoe_hip.h declares
(no compute calls here); CPU tensors are rejected instead of silently falling back."""
import ctypes
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "mopoe_hip.h")).read()
    return sorted(set(re.findall(r"\b(mopoe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from mimic_amd import ops
    if not os.path.exists(ops.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(ops.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 23
    for s in syms:
        assert hasattr(lib, s), s
    lib.mopoe_abi_version.restype = ctypes.c_int
    assert lib.mopoe_abi_version() == ops.ABI_VERSION


def test_no_cpu_fallback():
    from mimic_amd import ops
    with pytest.raises(ops.MopoeHipError):
        ops.colsum(torch.zeros(4, 4))
    with pytest.raises(ops.MopoeHipError):
        ops.laplace_nll_fwd(torch.zeros(8), torch.zeros(8), 0.75, 1.0)


def test_product_package_does_not_import_oracle():
    pkg = os.path.join(REPO, "mopoe-mimic_amd")
    for root, _d, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "mopoe_ref" not in text and "torch_backend" not in text, os.path.join(root, f)


def test_every_collective_is_asynchronous():
    """mimic_amd.parallel's rule: a SYNCHRONOUS collective records its completion event on the caller's
    stream, the process group's watchdog queries it while run_epochs.GraphedTrainStep is capturing that stream, and the
    HIP runtime answers with hipErrorCapturedEvent (capture invalidated, process aborted).  So every torch.distributed
    collective in the product package and in bench.py must carry async_op=True (and dist.barrier, which has no such form
    on the caller's stream, must not appear at all)."""
    import ast
    names = {"all_reduce", "broadcast", "all_gather", "all_gather_into_tensor", "reduce", "reduce_scatter",
             "reduce_scatter_tensor", "all_to_all", "all_to_all_single", "gather", "scatter", "send", "recv", "barrier",
             "monitored_barrier", "broadcast_object_list", "all_gather_object"}
    files = [os.path.join(REPO, "bench.py")]
    for root, _dirs, fs in os.walk(os.path.join(REPO, "mopoe-mimic_amd", "mimic_amd")):
        files += [os.path.join(root, f) for f in fs if f.endswith(".py")]
    seen, bad = 0, []
    for path in files:
        for node in ast.walk(ast.parse(open(path).read(), path)):
            if not (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in names):
                continue
            base = node.func.value
            if not (isinstance(base, ast.Name) and base.id == "dist"
                    or isinstance(base, ast.Attribute) and base.attr == "distributed"):
                continue
            seen += 1
            is_async = any(kw.arg == "async_op" and isinstance(kw.value, ast.Constant) and kw.value.value is True
                           for kw in node.keywords)
            if node.func.attr in ("barrier", "monitored_barrier", "broadcast_object_list", "all_gather_object") or not is_async:
                bad.append(f"{os.path.relpath(path, REPO)}:{node.lineno} dist.{node.func.attr}")
    assert seen >= 5, seen
    assert not bad, bad


def test_committed_plan_table_is_well_formed():
    """mimic_amd/plans_gfx950.json (tests/tools/make_plan_table.py on an MI355X): every key is op|14 geometry numbers|fusion
    flags, every value null (static heuristic) or [tile, split] inside the ranges include/mopoe_hip.h documents; a known
    BASELINE triple is found through ops._table_plan, an unknown one is not (the tuner then runs for it)."""
    import json
    from mimic_amd import ops
    with open(ops.PLAN_TABLE_PATH) as f:
        table = json.load(f)
    plans = table["plans"]
    assert len(plans) >= 500 and set(table["meta"]["configs"]) >= {"c2", "c3", "c5"}
    for key, v in plans.items():
        op, geo, _flags = key.split("|")
        assert op in ("fwd", "dgrad", "wgrad", "fwd16", "dgrad16", "wgrad16"), key
        assert len(geo.split(",")) == 14 and all(x.lstrip("-").isdigit() for x in geo.split(",")), key
        top = {"fwd": 19, "dgrad": 19, "wgrad": 10, "fwd16": 11, "dgrad16": 11, "wgrad16": 9}[op]     # the header's tile ranges
        assert v is None or (len(v) == 2 and 0 <= v[0] <= top and 1 <= v[1] <= 4096), (key, v)
    # rb1's shortcut conv at config #2 (B = 64): forward, no BN on load, no mask, statistics
    g = ops.Geom(64, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False)
    found, _ = ops._table_plan(("fwd", g, False, False, True))
    assert found
    found, _ = ops._table_plan(("fwd", ops.Geom(3, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False), False, False, True))
    assert not found
    # conv2 of rb1 in the training step: dropout mask + statistics + residual mix; without dropout (eval, or dropout disabled)
    # the same launch has no mask -- the table's plan for the masked form is used
    masked = ops._table_plan(("fwd", g, False, True, True, "mix"))
    unmasked = ops._table_plan(("fwd", g, False, False, True, "mix"))
    assert masked[0] and unmasked[0] and (masked[1].tile, masked[1].split) == (unmasked[1].tile, unmasked[1].split)


def test_split_bf16_switch_maps_committed_plans_back(monkeypatch):
    """MOPOE_F32_SPLIT_BF16=0: a committed fp32 plan on the bf16 matrix pipe (tiles 16..19, wgrad 7 / 8) is launched as the same
    tile on the fp32 MFMA (12..15, 5 / 6); bf16-family plans are untouched"""
    import json
    from mimic_amd import ops
    with open(ops.PLAN_TABLE_PATH) as f:
        plans = json.load(f)["plans"]
    ops._table_plan(("fwd", ops.Geom(64, 32, 32, 64, 64, 64, 128, 4, 4, 2, 2, 1, 1, False), False, False, True))   # loads the table
    def lookup(key_str):
        v = ops._plan_table[key_str]
        op = key_str.split("|")[0]
        class K(tuple):
            pass
        # _table_plan needs the structured key only through plan_key_str: patch that to hand back the string
        monkeypatch.setattr(ops, "plan_key_str", lambda key: key_str)
        found, p = ops._table_plan((op,))
        assert found
        return v, p
    on_pipe = [k for k, v in plans.items() if v is not None and ((k.split("|")[0] in ("fwd", "dgrad") and v[0] >= 16)
                                                               or (k.split("|")[0] == "wgrad" and v[0] in (7, 8, 9, 10)))]
    assert on_pipe, "the committed table holds no plan on the bf16 matrix pipe"
    monkeypatch.setattr(ops, "F32_SPLIT_BF16", False)
    for k in on_pipe[:20]:
        v, p = lookup(k)
        want = v[0] - 4 if k.split("|")[0] != "wgrad" else (v[0] - 2 if v[0] in (7, 8) else 2)
        assert p.tile == want and p.split == v[1], (k, v, p.tile)
    k16 = next(k for k, v in plans.items() if k.startswith("wgrad16") and v is not None and v[0] in (7, 8))
    v, p = lookup(k16)
    assert p.tile == v[0]
