"""The bench line's contract, checked on the line committed with the latest profile (bench.py itself needs an
MI355X): one JSON object with the driver's keys, `roofline` and `cpu_baseline` objects, no model keys."""
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _latest(pattern):
    files = sorted((ROOT / "profiles").glob(pattern), key=lambda p: (p.parent.name, int(p.stem.rsplit("v", 1)[1])))
    assert files, f"no committed {pattern} under profiles/"
    text = files[-1].read_text().strip().splitlines()
    return files[-1], json.loads(text[-1])


def latest_line():
    """The line as the driver sees it on stdout (round 6: the COMPACT line, floats at 5 significant digits)."""
    return _latest("r*/bench_line_unprofiled_v*.json")


def detail_line():
    """The detailed objects behind it (`--detail-json`)."""
    return _latest("r*/bench_detail_v*.json")


def test_committed_bench_line_has_the_contract_keys():
    path, d = latest_line()
    assert len(json.dumps(d)) < 6144, f"{path.name}: the printed line is {len(json.dumps(d))} bytes (the driver keeps the tail of stdout)"
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, f"{path.name}: missing {k}"
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["unit"] == "distances/s" and d["value"] > 5e9                      # BASELINE.json's target
    assert abs(d["value"] - d["config"]["distances_total"] / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] <= 1
    assert "traffic" in r and (r["traffic"] is None or r["traffic"] > 0) and 0 < r["whole_step_frac"] < r["frac"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and c["unit"] == d["unit"]
    assert c.get("gpu_vs_oracle_mismatching_pairs", 0) == 0
    # every leg's value is IN the printed line (the driver's stdout tail had lost verify.value)
    for leg in ("verify", "pipeline", "dense", "ragged", "sift_stats", "config3", "config4", "db"):
        assert d[leg]["value"] > 0, leg
    for leg in ("verify", "pipeline", "dense", "ragged", "sift_stats", "config3", "config4"):
        assert d[leg].get("gpu_vs_oracle_mismatching_pairs", 0) == 0, leg
    v = d["verify"]
    assert v["unit"] == "pairs/s" and abs(v["roofline"]["frac"] - v["roofline"]["achieved"] / v["roofline"]["peak"]) < 1e-4
    assert v["roofline"]["traffic"] is None or v["roofline"]["traffic"] > 0
    pl = d["pipeline"]
    assert pl["unit"] == "verified pairs/s" and pl["pairs_verified"] > 1000 and pl["pairs_total"] == 124750
    assert abs(pl["value"] - pl["pairs_verified"] / (pl["ms_per_step"] * 1e-3)) < 1e-3 * pl["value"]
    assert abs(pl["non_scan_ms"] - (pl["ms_per_step"] - pl["scan_ms"])) < 0.05 and pl["verified_pairs_mismatching"] == 0
    h = pl["host"]                                             # the call's own timeline adds up
    assert abs(h["c_call_ms"] - (h["verify_setup_ms"] + h["match_call_ms"] + h["close_and_launch_ms"] + h["verify_wait_pack_download_ms"])) < 0.05
    assert h["c_call_ms"] <= pl["ms_per_step"] + 0.05
    c4 = d["config4"]
    assert set(c4["parts"]) == {"sequential", "loop_voting_512x512", "loop_matches"} and c4["parts"]["sequential"]["scan_frac"] > 0.55
    dn = d["dense"]
    assert dn["unit"] == "distances/s" and dn["matches_per_pair"] > 500
    assert abs(dn["value"] - d["config"]["distances_total"] / (dn["ms_per_step"] * 1e-3)) < 1e-3 * dn["value"]


def test_committed_detail_line_is_consistent():
    path, d = detail_line()
    assert abs(d["value"] - d["config"]["distances_total"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    v = d["verify"]
    assert v["cpu_baseline"]["gpu_vs_oracle_mismatching_pairs"] == 0
    vr = v["roofline"]                                         # the second metric's own roofline (FP64 vector)
    assert abs(vr["frac"] - vr["achieved"] / vr["peak"]) < 1e-9 and 0 < vr["frac"] < 1
    assert abs(vr["frac_of_no_fma_ceiling"] - 2 * vr["frac"]) < 1e-6 and vr["kernel"] in ("tvg_kernel", "tvg_e_kernel + tvg_fh_kernel")
    pl = d["pipeline"]                                         # configs[2] chained on the device
    assert abs(pl["value"] - pl["pairs_verified"] / (pl["ms_per_step"] * 1e-3)) < 1e-6 * pl["value"]
    assert pl["cpu_baseline"]["gpu_vs_oracle_mismatching_pairs"] == 0
    assert pl["cpu_baseline"]["verified_pairs_mismatching"] == 0 and pl["cpu_baseline"]["verified_pairs_checked"] > 0
    assert 0 < pl["roofline"]["frac"] < 1
    g = pl["guided"]                                           # the guided re-match of the verified pairs
    assert g["unit"] == "entries/s" and g["value"] > 1.6e12                    # >= 3 x 5.3e11
    assert g["pairs"] == sum(g["pairs_by_kernel"].values()) and g["pairs_by_kernel"]["candidate_generation"] > 0
    assert g["cpu_baseline"]["gpu_vs_oracle_mismatching_pairs"] == 0
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_compact", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    c = bench.compact_line(d)                                  # the printed line is a function of the detailed one
    assert c["verify"]["value"] == bench._num(d["verify"]["value"]) and c["pipeline"]["value"] == bench._num(d["pipeline"]["value"])
    assert len(json.dumps(c)) < 6144


def test_committed_scaled_config_lines():
    """bench.py --config 3 / 4 at N=1 (profiles/r02): the contract keys, strong scaling, the stated workloads."""
    for name, pairs in (("bench_config3_n1_v2.json", 1999000), ("bench_config4_n1_v2.json", None)):
        path = ROOT / "profiles" / "r02" / name
        d = json.loads(path.read_text().strip().splitlines()[-1])
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline"):
            assert k in d, f"{name}: missing {k}"
        assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["unit"] == "distances/s" and d["value"] > 5e9
        assert "REDUCED" not in d["config"]["workload"]
        if pairs:
            assert d["config"]["pairs_total"] == pairs
        else:
            assert d["config"]["loop_queries"] == 1000 and d["config"]["pairs_total"] > 500000
        assert abs(d["value"] - d["config"]["distances_per_step_all_ranks"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_host_cores_reads_the_cgroup_quota():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.host_cores()
    import os
    assert 1 <= n <= (os.cpu_count() or 1)


def _run_bench(*argv, timeout=600):
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)          # the entry must find out by itself that it has to launch its ranks
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]     # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_multi_rank_entry_launches_itself_and_carries_the_configs3_leg():
    """`python bench.py --gpus 2` (no launcher around it, no --config): the entry starts its two ranks under
    torch.distributed.run; the line is the weak-scaled configs[1] headline (the metric at every N) and carries
    BASELINE configs[3] - fixed pair set, sharded, strong scaling: the curve north_star states its ">= 7x at 8 GPUs"
    on - as the `config3` leg, with its scalars repeated at the top level.  Run here without a GPU (--cpu-dry-run:
    gloo, CPU oracle in place of the kernels, tiny sizes): a plumbing check, every value is null."""
    d = _run_bench("--gpus", "2", "--cpu-dry-run", "--images", "9", "--feats", "64", "--steps", "1", "--warmup", "1")
    assert d["dry_run"] is True and d["value"] is None and d["metric"].startswith("DRY RUN")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "configs[1]" in d["config"]["workload"] and d["config"]["pairs_total"] == 13 * 12 // 2   # 9 x sqrt(2) images
    assert 0 < d["config"]["pairs_per_rank"] < d["config"]["pairs_total"] and d["config"]["gather_path"] == "padded"
    l3 = d["config3"]
    for k in ("value", "ms_per_step", "exchange_ms_per_step", "per_rank_kernel_ms", "rccl_ranks", "gather_path"):
        assert k in l3, k
    assert l3["scaling"] == "strong" and l3["n_gpus"] == 2 and l3["value"] is None and l3["dry_run"] is True
    assert d["config3_value"] is None and d["config3_ms_per_step"] == l3["ms_per_step"] > 0
    assert d["config3_exchange_ms_per_step"] == l3["exchange_ms_per_step"] > 0
    assert len(l3["per_rank_kernel_ms"]) == 2 and l3["rccl_ranks"] == 0 and l3["gather_path"] == "padded"
    c = l3["line"]["config"]
    assert "configs[3]" in c["workload"] and c["workload"].startswith("REDUCED 9 x 64")
    assert c["pairs_total"] == 36 and 0 < c["pairs_rank0"] < 36
    assert c["distances_per_step_all_ranks"] == 36 * 64 * 64      # every pair matched exactly once over the two ranks
    assert c["backend"] == "gloo" and c["rccl_ranks"] == 0
    assert len(c["kernel_ms_per_step_by_rank"]) == 2 and len(c["exchange_ms_per_step_by_rank"]) == 2
    assert c["exchange_ms_per_step"] == max(c["exchange_ms_per_step_by_rank"]) > 0
    # what a measured multi-rank run PRINTS is the compact form of such a line (dry runs print the detailed one):
    # the scaling curve's scalars survive it
    spec = importlib.util.spec_from_file_location("bench_for_compact2", ROOT / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    k = bench.compact_line(d)
    assert len(json.dumps(k)) < 6144 and k["n_gpus"] == 2 and k["scaling"] == "weak" and "roofline" in k and k["config"]["pairs_total"] == 78
    k3 = k["config3"]
    assert k3["n_gpus"] == 2 and k3["scaling"] == "strong" and len(k3["per_rank_kernel_ms"]) == 2 and k3["gather_path"] == "padded"
    assert k3["exchange_ms_per_step"] > 0 and abs(k["config3_ms_per_step"] - l3["ms_per_step"]) < 1e-3 * l3["ms_per_step"] and "config3_value" in k


def test_configs3_alone_is_still_a_line_of_its_own():
    d = _run_bench("--gpus", "2", "--cpu-dry-run", "--config", "3", "--images", "9", "--feats", "64", "--steps", "1",
                   "--warmup", "0")
    assert d["dry_run"] and d["scaling"] == "strong" and d["n_gpus"] == 2 and "config3" not in d
    assert d["config"]["pairs_total"] == 36 and d["config"]["gather_path"] == "padded"


def test_multi_rank_entry_configs4_dry_run():
    d = _run_bench("--gpus", "2", "--cpu-dry-run", "--config", "4", "--images", "24", "--feats", "64", "--steps", "1",
                   "--warmup", "0")
    c = d["config"]
    assert d["dry_run"] and "configs[4]" in c["workload"] and c["loop_queries"] >= 1 and c["loop_pairs"] > 0
    assert c["pairs_total"] == 24 * 23 // 2          # overlap 50 >= 24 images: every pair is a sequential pair


def test_bench_refuses_gloo_for_a_measurement():
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--backend", "gloo"], capture_output=True,
                       text=True, timeout=120, cwd=str(ROOT))
    assert r.returncode != 0 and "cpu-dry-run" in (r.stdout + r.stderr)


def test_eight_rank_entry_dry_run():
    """`python bench.py --gpus 8` as the driver will launch it on the 8-GPU node, rehearsed without a GPU: eight ranks,
    the weak-scaled headline (500 * sqrt(8) images' worth of pairs at the reduced size) and the configs[3] strong-scaling
    leg, every pair matched exactly once over the eight ranks, one JSON line from rank 0."""
    d = _run_bench("--gpus", "8", "--cpu-dry-run", "--images", "6", "--feats", "32", "--steps", "1", "--warmup", "0",
                   timeout=900)
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] is None
    n_img = d["config"]["images"] if "images" in d["config"] else None
    assert d["config"]["pairs_total"] >= 8 and 0 <= d["config"]["pairs_per_rank"] <= d["config"]["pairs_total"]
    l3 = d["config3"]
    assert l3["n_gpus"] == 8 and l3["scaling"] == "strong" and len(l3["per_rank_kernel_ms"]) == 8
    c = l3["line"]["config"] if "line" in l3 else l3["config"]
    assert c["pairs_total"] == 15 and c["distances_per_step_all_ranks"] == 15 * 32 * 32
    assert len(c["kernel_ms_per_step_by_rank"]) == 8
