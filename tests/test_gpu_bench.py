"""The driver-facing entry points on a real GPU: bench.py's one-line JSON contract and __graft_entry__.smoke()."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"] and d["value"] > 1.0
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert d["roofline_hbm_class"]["heun_step_cfg2_4MiB"]["frac_hbm_peak"] > 0.2
    assert d["ms_per_step_min"] <= d["ms_per_step_median"] <= d["ms_per_step_max"] and "gpu_state" in d
    # configs 3 and 5 ride on the default line: neither may have failed silently, each carries three timed replays and a roofline
    oc = d["other_configs"]
    for name in ("config3_adm128", "config5_share_cond_punetg64"):
        assert "error" not in oc[name], oc[name]
        assert oc[name]["timed_replays"] >= 3 and oc[name]["ms_per_run"]["min"] <= oc[name]["ms_per_run"]["max"]
        r = oc[name]["roofline"]
        assert r["bound"] == "mfma" and 0.05 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3


def test_graft_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()
