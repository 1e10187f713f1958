"""The compiler must not leave vector instructions where EXEC is zero (tools/check_exec0_reloads.py).

Round 6: after a change of the sampler, LLVM's register allocator placed the spill RELOAD of a live value (the watermark
RANSAC's max_num_trials) in the exit block of a lane-retiring loop, before the `s_or_b64 exec` that ends the region: it
ran for no lane, and report.num_trials came back as whatever the loop had left in the register.  The GPU tests caught
it; this test finds the pattern without a GPU, in the assembly of every kernel source, so that a change that provokes
it again fails here first.
"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))


def _compile_to_asm(src: Path, out: Path) -> None:
    from pycolmap_amd import build
    flags = [f for f in build.HIPCC_FLAGS if f not in ("-fPIC",)]
    cmd = [build._hipcc(), *flags, "-S", "--cuda-device-only", str(src), "-o", str(out), f"-I{ROOT / 'include'}"]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def test_no_vector_instruction_runs_under_exec_zero_behind_a_loop():
    from pycolmap_amd import build
    import check_exec0_reloads as chk
    try:
        build._hipcc()
    except RuntimeError:
        pytest.skip("hipcc not found")
    srcs = [build.CSRC / n for n in build.HIP_SOURCES if (build.CSRC / n).exists()]
    with tempfile.TemporaryDirectory() as td:
        outs = [Path(td) / (s.stem + ".s") for s in srcs]
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            list(pool.map(lambda so: _compile_to_asm(*so), zip(srcs, outs)))
        found = {o.name: chk.scan(str(o)) for o in outs}
    bad = {k: v[:5] for k, v in found.items() if v}
    assert not bad, f"vector instructions under EXEC = 0 behind a loop: {bad}"


def test_the_scanner_sees_the_pattern(tmp_path):
    import check_exec0_reloads as chk
    asm = tmp_path / "x.s"
    asm.write_text("\n".join([
        "f:",
        ".LBB0_1:",
        "\tglobal_load_dwordx4 v[6:9], v[2:3], off",
        "\ts_andn2_b64 exec, exec, s[2:3]",
        "\ts_cbranch_execnz .LBB0_1",
        ".LBB0_2:",
        "\ts_waitcnt lgkmcnt(0)",
        "\tscratch_load_dword v6, off, s33 offset:264 ; 4-byte Folded Reload",
        ".LBB0_3:",
        "\ts_or_b64 exec, exec, s[0:1]",
        "\tv_cndmask_b32_e64 v0, v6, v4, s[60:61]",
        ""]))
    hits = chk.scan(str(asm))
    assert len(hits) == 1 and "scratch_load_dword v6" in hits[0][1]
    ok = tmp_path / "y.s"
    ok.write_text("\n".join([
        "f:", ".LBB0_1:", "\ts_andn2_b64 exec, exec, s[2:3]", "\ts_cbranch_execnz .LBB0_1",
        "\ts_or_b64 exec, exec, s[0:1]", "\tscratch_load_dword v6, off, s33 offset:264", ""]))
    assert chk.scan(str(ok)) == []
