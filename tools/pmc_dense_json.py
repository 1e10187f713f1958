#!/usr/bin/env python3
"""profiles/r06/pmc_dense_r06.json from tools/pmc_dense_r06.sh's text (per-dispatch counter sums; the SECOND half of a
kernel's dispatches are the dense leg's: bench.py runs the sparse headline step first).
    python tools/pmc_dense_json.py gpurun_out/r06/pmc_dense_r06_v2.txt [out.json]"""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
NPAIRS, N = 124750, 4096


def main(txt):
    lines = Path(txt).read_text().splitlines()
    box = next((ln[4:].strip() for ln in lines if ln.startswith("box:")), None)
    pat = re.compile(r"^(.*?)\s+(\S+)\s+dispatches=\s*(\d+)\s+first_half_sum=(\S+)\s+second_half_sum=(\S+)")
    k = {}
    for ln in lines:
        m = pat.match(ln)
        if m:
            k.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
    leg = next((ln for ln in lines if ln.startswith("dense leg under the profiler")), "")
    out = {"source": f"{Path(txt).name} (tools/pmc_dense_r06.sh: rocprofv3 --kernel-trace --pmc, ONE counter per pass, kernel filter on the "
                     "stage's kernels; python bench.py --steps 1 --warmup 0 with the dense leg on; the dense leg's dispatches are the second half "
                     "of each kernel's)",
           "box": box, "passes_ok": sum(1 for ln in lines if ln.startswith("rc=0")), "passes": sum(1 for ln in lines if ln.startswith("rc=")),
           "workload": "500 images x 4096 descriptors, every pair overlapping (1,064 matches per pair), one step = 124,750 pairs in 3 batches",
           "units": "per step of the dense leg (sums over its launches); bytes: reads = TCC_EA0_RDREQ x 64 B (raw; FETCH_SIZE x 1024 is the same "
                    "number; the guide's x2 correction holds for 16-B-per-lane streaming reads only and is NOT applied), writes = 64-B requests x 64 "
                    "+ the others x 32",
           "dense_leg_under_the_profiler": leg.split(":", 1)[1].strip() if leg else None, "kernels": {}}
    for name, c in sorted(k.items()):
        d = {"launches": c["duration_ms"][0] // 2 if "duration_ms" in c else None,
             "ms": c["duration_ms"][2] if "duration_ms" in c else None}
        if "TCC_EA0_RDREQ_sum" in c:
            d["read_bytes"] = c["TCC_EA0_RDREQ_sum"][2] * 64.0
        if "FETCH_SIZE" in c:
            d["fetch_size_kb"] = c["FETCH_SIZE"][2]
        if "TCC_EA0_WRREQ_sum" in c:
            w, w64 = c["TCC_EA0_WRREQ_sum"][2], c.get("TCC_EA0_WRREQ_64B_sum", (0, 0, 0))[2]
            d["write_bytes"] = w64 * 64.0 + (w - w64) * 32.0
            d["write_requests"], d["write_requests_64B"] = w, w64
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            d["l2_hit_rate"] = c["TCC_HIT_sum"][2] / max(c["TCC_HIT_sum"][2] + c["TCC_MISS_sum"][2], 1.0)
        if "SQ_INSTS_VALU_MFMA_I8" in c:
            d["mfma_i8_wave_instructions"] = c["SQ_INSTS_VALU_MFMA_I8"][2]
            if d["ms"] and d["mfma_i8_wave_instructions"]:
                # one v_mfma_i32_32x32x32_i8 = 32768 MACs = 65536 int8 ops
                d["int8_tops"] = d["mfma_i8_wave_instructions"] * 65536.0 / (d["ms"] * 1e-3) / 1e12
                d["frac_of_int8_peak"] = d["int8_tops"] / 5000.0
        if d.get("ms") and d.get("read_bytes") is not None and d.get("write_bytes") is not None:
            d["memory_side_tb_s"] = (d["read_bytes"] + d["write_bytes"]) / (d["ms"] * 1e-3) / 1e12
        out["kernels"][name] = d
    fw = next((v for n, v in out["kernels"].items() if n.startswith("match_mfma_kernel<0")), None)
    rv = next((v for n, v in out["kernels"].items() if n.startswith("match_mfma_kernel<1")), None)
    rs = out["kernels"].get("resolve_index_mfma_kernel")
    if rv and rv.get("mfma_i8_wave_instructions"):
        # X rows the reverse scan covers per pair (candidate lists padded to 128-row segments): MACs / (128 x 4096 per row)
        rows = rv["mfma_i8_wave_instructions"] * 32768.0 / (128.0 * N) / NPAIRS
        out["reverse_scan"] = {"candidate_rows_per_pair_padded": rows, "share_of_image_2": rows / N,
                               "read_bytes_per_candidate_row": rv["read_bytes"] / (rows * NPAIRS) if rv.get("read_bytes") else None,
                               "rate_vs_forward_scan": (rv["frac_of_int8_peak"] / fw["frac_of_int8_peak"]) if fw and fw.get("frac_of_int8_peak") else None,
                               "frac_of_int8_peak": rv.get("frac_of_int8_peak"), "forward_frac_of_int8_peak": fw.get("frac_of_int8_peak") if fw else None}
    if rs and rs.get("write_requests"):
        # resolve writes one 32-byte request per row it settles (8 bytes of the 16-byte record): accepted rows of both sides
        acc = (rs["write_requests"] - rs.get("write_requests_64B", 0.0))
        out["resolve"] = {"rows_settled_per_pair_both_sides": acc / NPAIRS,
                          "read_bytes_per_row": rs["read_bytes"] / acc if rs.get("read_bytes") else None,
                          "write_bytes_per_row": rs["write_bytes"] / acc}
    dst = Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "profiles" / "r06" / "pmc_dense_r06.json"
    dst.write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps({k_: out[k_] for k_ in ("reverse_scan", "resolve") if k_ in out}, indent=1))
    for n, v in out["kernels"].items():
        print(n, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms", "read_bytes", "write_bytes", "l2_hit_rate", "frac_of_int8_peak", "memory_side_tb_s")})


if __name__ == "__main__":
    main(sys.argv[1])
