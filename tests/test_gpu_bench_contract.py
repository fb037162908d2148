"""bench.py keeps its one-line JSON contract (run on a small grid; the CPU baseline leg uses the oracle port when
oracle/_ref is not on the box)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = _run(["--grid", "24", "--steps", "2", "--warmup", "1", "--cpu-sample-grid", "16"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "GFLOP/s"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["residual"] < 1e-10
    # socket power / shader clock during the timed steps (nulls where rocm-smi is absent or silent)
    assert set(d["config"]["power"]) >= {"socket_power_w_avg", "sclk_mhz_avg", "samples"}
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0


@pytest.mark.parametrize("facto", ["ldlt", "lu"])
def test_bench_other_factorizations(facto):
    d = _run(["--grid", "20", "--steps", "1", "--warmup", "0", "--facto", facto, "--no-cpu-baseline"])
    assert d["config"]["residual"] < 1e-10 and d["value"] > 0
