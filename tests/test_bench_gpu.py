"""bench.py's output contract on the GPU (one short run per storage family): one JSON line with the driver's keys, the
roofline object measured live, no CPU baseline when asked not to."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg,dtype,bound", [("c2", "f32", "mfma"), ("c5", "bf16", "hbm")])
def test_bench_line(cfg, dtype, bound):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--config", cfg, "--steps", "4", "--warmup", "2",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["metric"] == "samples/sec" and line["n_gpus"] == 1 and line["steps"] == 4 and line["dtype"] == dtype
    assert line["value"] > 0 and line["vs_baseline"] is None and line["cpu_baseline"] is None and line["scaling"] == "weak"
    assert "workload" in line["config"] and line["config"]["hip_graph"] is True
    r = line["roofline"]
    assert r["bound"] == bound and r["unit"] == ("TFLOP/s" if bound == "mfma" else "GB/s")
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["avg_launch_us"] > 0
    assert abs(line["value"] - line["config"]["global_batch"] / line["ms_per_step"] * 1e3) <= 0.01 * line["value"]
