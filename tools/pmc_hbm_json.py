#!/usr/bin/env python3
"""tools/pmc_r05.sh's HBM passes -> the JSON bench.py reads roofline.traffic from (with the hash of the kernel source the
counters were taken on).  usage: pmc_hbm_json.py pmc_hbm_r04_<tag>.txt out.json"""
import hashlib, json, re, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
txt = Path(sys.argv[1]).read_text()
cnt, calls, total = {}, None, None
for line in txt.splitlines():
    m = re.match(r"\s+(\S.*?)\s+calls=(\d+)\s+total=(\S+)\s+avg=(\S+)", line)
    if m and "match_mfma_kernel<0" in m.group(1):
        calls, total = int(m.group(2)), float(m.group(3))
    m = re.match(r"\s+(\S.*?)\s+(\w+)\s+n=(\d+)\s+sum=(\S+)\s+avg=(\S+)", line)
    if m and "match_mfma_kernel<0" in m.group(1):
        cnt[m.group(2)] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
box = next((ln[4:].strip() for ln in txt.splitlines() if ln.startswith("box:")), None)
out = {"source": f"{Path(sys.argv[1]).name} (tools/pmc_r06.sh: rocprofv3 --kernel-trace --pmc, ONE counter per pass, every pass on the box named "
                 "in `box`; python bench.py --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 0 --no-pipeline --no-dense --no-ragged "
                 "--no-sift-stats --no-config3 --no-config4)",
       "box": box, "passes_ok": txt.count("rc=0"), "passes": txt.count("rc="),
       "kernel": "match_mfma_kernel<0, 8, 4>"}
# one step, no warm-up: the kernel's calls are the step's launches (two full batches, or - since the last batch of a
# multi-batch call is a quarter-size one - three); the counters below are averages over them, like bench.py's
# avg_kernel_ms
out["launches_per_step"] = calls
out["pairs_per_launch"] = 124750 // calls if calls else None
out["workload"] = f"500 images x 4096 descriptors, 124750 pairs, {calls} launches per step (per-launch figures are averages over them)"
if "FETCH_SIZE" not in cnt and "TCC_EA0_RDREQ_sum" in cnt:
    # rocprofv3 dies on the derived FETCH_SIZE on some boxes (rc 139); the guide's HBM section gives its definition -
    # FETCH_SIZE = TCC_EA0_RDREQ x 64 B, in KB - so the raw request counter of a separate pass stands in for it
    cnt["FETCH_SIZE"] = (cnt["TCC_EA0_RDREQ_sum"][0], cnt["TCC_EA0_RDREQ_sum"][1] * 64 / 1024, cnt["TCC_EA0_RDREQ_sum"][2] * 64 / 1024)
    out["fetch_size_from"] = "TCC_EA0_RDREQ_sum x 64 B (the definition of FETCH_SIZE, MI355X_MICROARCH.md HBM section); the derived counter's pass died"
    out["read_requests_per_launch"] = cnt["TCC_EA0_RDREQ_sum"][2]
    if "TCC_EA0_RDREQ_32B_sum" in cnt:
        out["read_requests_32B_per_launch"] = cnt["TCC_EA0_RDREQ_32B_sum"][2]
if "FETCH_SIZE" in cnt:
    out["fetch_size_kb_per_launch"] = cnt["FETCH_SIZE"][2]
    out["fetch_bytes_per_launch_corrected"] = cnt["FETCH_SIZE"][2] * 1024 * 2
    out["correction"] = ("FETCH_SIZE is in KB and counts 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM section): "
                         "bytes = FETCH_SIZE * 1024 * 2")
if "TCC_EA0_WRREQ_sum" in cnt:
    w, w64 = cnt["TCC_EA0_WRREQ_sum"][2], cnt.get("TCC_EA0_WRREQ_64B_sum", (0, 0, 0))[2]
    out["write_requests_per_launch"] = w
    out["write_requests_64B_per_launch"] = w64
    out["write_bytes_per_launch"] = w64 * 64 + (w - w64) * 32
if "GRBM_GUI_ACTIVE" in cnt:
    out["grbm_gui_active_per_launch"] = cnt["GRBM_GUI_ACTIVE"][2]
    if calls and total:
        ms = total / calls / 1e3   # top_kernels totals are in microseconds
        out["kernel_ms_under_pmc"] = ms
        out["clock_ghz"] = cnt["GRBM_GUI_ACTIVE"][2] / 8 / (ms * 1e6)   # cycles per XCD / (ms * 1e6 ns/ms) = GHz
out["kernel_source_sha256"] = {f: hashlib.sha256((ROOT / f).read_bytes()).hexdigest()
                               for f in ("pycolmap_amd/csrc/match_mfma.hip",)}
Path(sys.argv[2]).write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
