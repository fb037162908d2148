"""Host-side pieces of bench.py that need no GPU: the power / clock sampler parses rocm-smi's text (the shape seen on the
MI355X boxes of the pool) and reports nulls when the tool is absent or silent."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

SMI_TEXT = """
============================ ROCm System Management Interface ============================
=============================== Current clock frequencies ================================
GPU[0]		: fclk clock level: 0: (1250Mhz)
GPU[0]		: mclk clock level: 0: (2000Mhz)
GPU[0]		: sclk clock level: S: (2364Mhz)
GPU[0]		: socclk clock level: S: (38Mhz)
==========================================================================================
=================================== Power Consumption ====================================
GPU[0]		: Current Socket Graphics Package Power (W): 1296.0
==========================================================================================
"""


def test_power_watch_parses_rocm_smi_text(monkeypatch):
    class R:
        stdout = SMI_TEXT

    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R())
    w = bench.PowerWatch(True)
    w.exe = "rocm-smi"                       # (whether or not the tool is installed here)
    with w:
        time.sleep(0.7)
    s = w.summary()
    assert s["samples"] >= 1 and s["sclk_mhz_avg"] == 2364.0 and s["socket_power_w_avg"] == 1296.0


def test_power_watch_is_silent_without_the_tool(monkeypatch):
    w = bench.PowerWatch(False)
    with w:
        pass
    s = w.summary()
    assert s["samples"] == 0 and s["sclk_mhz_avg"] is None and s["socket_power_w_avg"] is None and s["source"] is None
