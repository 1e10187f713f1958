"""The bench line's contract, checked on the line committed with the latest profile (bench.py itself needs an
MI355X): one JSON object with the driver's keys, `roofline` and `cpu_baseline` objects, no model keys."""
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def latest_line():
    files = sorted((ROOT / "profiles").glob("r*/bench_line_unprofiled_v*.json"),
                   key=lambda p: (p.parent.name, int(p.stem.rsplit("v", 1)[1])))
    assert files, "no committed bench line under profiles/"
    text = files[-1].read_text().strip().splitlines()
    return files[-1], json.loads(text[-1])


def test_committed_bench_line_has_the_contract_keys():
    path, d = latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, f"{path.name}: missing {k}"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "distances/s" and d["value"] > 5e9                      # BASELINE.json's target
    assert abs(d["value"] - d["config"]["distances_total"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] <= 1
    assert "traffic" in r and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["unit"] == d["unit"]
    assert c.get("gpu_vs_oracle_mismatching_pairs", 0) == 0
    v = d.get("verify")
    if v:
        assert v["unit"] == "pairs/s" and v["value"] > 0 and v["cpu_baseline"]["gpu_vs_oracle_mismatching_pairs"] == 0


def test_host_cores_reads_the_cgroup_quota():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.host_cores()
    import os
    assert 1 <= n <= (os.cpu_count() or 1)
