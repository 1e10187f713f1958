#!/usr/bin/env python3
"""profiles/rNN/pmc_tvg_rNN.json from the text summary profiles/r03/recipes/pmc_tvg_r03.sh / pmc_tvg_r05.sh write (per-call averages of the SQ counters
of the two verification kernels + the derived ratios + the sha256 of the kernel sources the counters belong to;
bench.py's verify.roofline.executed reads it while the sources still hash to the same values).
    python tools/pmc_tvg_json.py gpurun_out/r05/pmc_tvg_r05_v1.txt [out.json]"""
import hashlib
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SOURCES = ["pycolmap_amd/csrc/tvg_core.h", "pycolmap_amd/csrc/tvg_e.hip", "pycolmap_amd/csrc/tvg_fh.hip",
           "pycolmap_amd/csrc/tvg_math.h"]
PAIRS = 16384


def main(txt):
    out = {"source": f"{txt} (tools/pmc_tvg_r0N.sh: rocprofv3 --kernel-trace --pmc, one counter group per pass; AMC_TVG_SLICES=1 python bench.py "
                     "--images 40 --steps 1 --warmup 0 --no-cpu-baseline --verify-pairs 16384 --no-pipeline --no-dense: one E and one F/H "
                     "launch per call, behind each other - the per-kernel occupancy figures are those of a kernel alone on the machine)",
           "workload": "16,384 pairs of bench.py's verify leg (4096 distinct seeded calibrated scenes, ~420 matches)",
           "pairs_per_call": PAIRS,
           "units": "per-call averages; SQ_* cycle counters are in quad-cycles (x4 = shader clocks), summed over all SIMDs; "
                    "GRBM_GUI_ACTIVE is summed over the 8 XCDs"}
    pat = re.compile(r"amc::(tvg_(?:e|fh)_kernel)\S*.*?\s(SQ_\w+|GRBM_\w+|FETCH_SIZE|TCC_\w+)\s+n=(\d+)\s+sum=(\S+)\s+avg=(\S+)")
    tim = re.compile(r"amc::(tvg_(?:e|fh)_kernel)\(.*calls=(\d+)\s+total=(\S+)\s+avg=(\S+)")
    for line in Path(txt).read_text().splitlines():
        m = pat.search(line)
        if m:
            out.setdefault(m.group(1), {})[m.group(2)] = float(m.group(5))
            continue
        m = tim.search(line)
        if m:
            out.setdefault(m.group(1), {}).setdefault("kernel_us_under_counters", float(m.group(4)))
    total = 0.0
    for k, waves in (("tvg_e_kernel", None), ("tvg_fh_kernel", None)):
        c = out[k]
        simd_quads = c["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0      # quad-cycles available on the 1024 SIMDs during the call
        c["wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        c["mean_waves_per_simd"] = c["SQ_WAVE_CYCLES"] / simd_quads
        c["valu_busy_share_of_simd_cycles"] = c["SQ_ACTIVE_INST_VALU"] / simd_quads
        c["valu_busy_share_while_waves_resident"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] * round(c["mean_waves_per_simd"] + 0.49)
        c["salu_per_valu"] = c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"]
        c["lds_bank_conflict_over_active_lds"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_ACTIVE_INST_LDS"]
        total += c["SQ_INSTS_VALU"]
        # round 6: share of the 64 lanes that take part in the vector instructions issued (averaged over issue cycles)
        if "SQ_THREAD_CYCLES_VALU" in c:
            c["lane_utilisation"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
        # ... and the bytes through the L2's memory side per call: FETCH_SIZE is in KB and reports half of a wide read on
        # gfx950 (MI355X_MICROARCH.md, HBM section: x 1024 x 2); writes are 64-byte requests but for the few 32-byte ones
        if "FETCH_SIZE" in c:
            c["read_bytes_per_call"] = c["FETCH_SIZE"] * 1024.0 * 2.0
        if "TCC_EA0_WRREQ_sum" in c:
            w64 = c.get("TCC_EA0_WRREQ_64B_sum", c["TCC_EA0_WRREQ_sum"])
            c["write_bytes_per_call"] = w64 * 64.0 + (c["TCC_EA0_WRREQ_sum"] - w64) * 32.0
    out["valu_wave_instructions_per_pair"] = total / PAIRS
    e, f = out["tvg_e_kernel"], out["tvg_fh_kernel"]
    if "lane_utilisation" in e and "lane_utilisation" in f:
        out["lane_utilisation"] = (e["SQ_THREAD_CYCLES_VALU"] + f["SQ_THREAD_CYCLES_VALU"]) / (64.0 * (e["SQ_ACTIVE_INST_VALU"] + f["SQ_ACTIVE_INST_VALU"]))
    if "read_bytes_per_call" in e and "read_bytes_per_call" in f and "write_bytes_per_call" in e and "write_bytes_per_call" in f:
        out["read_bytes_per_pair"] = (e["read_bytes_per_call"] + f["read_bytes_per_call"]) / PAIRS
        out["write_bytes_per_pair"] = (e["write_bytes_per_call"] + f["write_bytes_per_call"]) / PAIRS
        out["traffic_note"] = ("both kernels of one call of 16,384 pairs; the reads and writes are the waves' workspaces and spill "
                               "slots streaming through the L2 (footprint: hundreds of MB, 4 MB of L2 per XCD), not the inputs")
    out["kernel_source_sha256"] = {f: hashlib.sha256((ROOT / f).read_bytes()).hexdigest() for f in SOURCES}
    dst = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "profiles" / "r06" / "pmc_tvg_r06.json"
    dst.write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({k: out[k] for k in ("valu_wave_instructions_per_pair",)}, indent=1))
    for k in ("tvg_e_kernel", "tvg_fh_kernel"):
        print(k, {x: round(out[k][x], 3) for x in ("wait_any_over_wave_cycles", "mean_waves_per_simd", "valu_busy_share_of_simd_cycles",
                                                     "salu_per_valu", "lds_bank_conflict_over_active_lds", "lane_utilisation") if x in out[k]},
              {x: "%.3g" % out[k][x] for x in ("read_bytes_per_call", "write_bytes_per_call") if x in out[k]})


if __name__ == "__main__":
    main(sys.argv[1])
