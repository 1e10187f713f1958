"""Which use of amc_comm_* leaves a process that cannot exit cleanly?  Runs each variant in its own process and prints
its exit status (tools-only diagnostic)."""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
VARIANTS = {
    "ctx_only": "ctx = _capi.Context(0); ctx.close()",
    "id_only": "_capi.comm_unique_id()",
    "create_destroy": "ctx = _capi.Context(0); c = ctx.comm_create(1, 0, _capi.comm_unique_id()); c.close(); ctx.close()",
    "create_leak": "ctx = _capi.Context(0); c = ctx.comm_create(1, 0, _capi.comm_unique_id()); c._h = None; ctx._comms = []",
    "gather_destroy": ("ctx = _capi.Context(0); c = ctx.comm_create(1, 0, _capi.comm_unique_id());\n"
                       "off = np.array([0, 2, 3], np.uint64); m = np.arange(6, dtype=np.uint32).reshape(3, 2)\n"
                       "print(c.allgather_match_tables(None, off, m)[1].tolist()); c.close(); ctx.close()"),
    "torch_first_create_destroy": ("import torch; torch.zeros(1).cuda();\n"
                                   "ctx = _capi.Context(0); c = ctx.comm_create(1, 0, _capi.comm_unique_id()); c.close(); ctx.close()"),
}
VARIANTS["gather_then_torch"] = VARIANTS["gather_destroy"] + "\nimport torch; print(torch.cuda.device_count())"
VARIANTS["gather_resident_then_torch"] = (
    "from pycolmap_amd import synth\nrng = np.random.default_rng(12); imgs = synth.scene_images(rng, 5, 640)\n"
    "ctx = _capi.Context(0); ctx.reserve_slots(5)\n"
    "for k, im in enumerate(imgs): ctx.upload_descriptors(k, im)\n"
    "s1, s2 = synth.exhaustive_pairs(5); off, m, _ = ctx.match_pairs(s1, s2)\n"
    "c = ctx.comm_create(1, 0, _capi.comm_unique_id()); print(c.allgather_match_tables(None, off, None)[2]['num_matches'])\n"
    "try:\n    c.allgather_match_tables(np.zeros(len(s1), np.uint64), off, m)\nexcept _capi.AmcError as e: print('expected', e.code)\n"
    "c.close(); import torch; print(torch.cuda.device_count()); ctx.close()")
for name, body in VARIANTS.items():
    code = f"import sys; sys.path.insert(0, {ROOT!r}); import numpy as np\nfrom pycolmap_amd import _capi\n{body}\nprint('body done', flush=True)"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    tail = (r.stdout + r.stderr).strip().splitlines()[-3:]
    print(f"{name}: rc={r.returncode} | " + " | ".join(t[:120] for t in tail), flush=True)
for name, cmd in {
    "pytest_c_abi_test": [sys.executable, "-m", "pytest", "tests/test_multigpu_gpu.py", "-q", "-k", "c_abi_without"],
    "pytest_c_abi_plus_skips": [sys.executable, "-m", "pytest", "tests/test_multigpu_gpu.py", "-q", "-k", "c_abi_without or two_rank"],
    "bench_config3_mid": [sys.executable, "bench.py", "--images", "120", "--feats", "4096", "--steps", "1", "--warmup", "0", "--verify-pairs", "0",
                          "--no-pipeline", "--no-ragged", "--no-sift-stats", "--no-dense", "--no-db", "--no-cpu-baseline"],
    "bench_no_config3_mid": [sys.executable, "bench.py", "--images", "120", "--feats", "4096", "--steps", "1", "--warmup", "0", "--verify-pairs", "0",
                             "--no-pipeline", "--no-ragged", "--no-sift-stats", "--no-dense", "--no-db", "--no-cpu-baseline", "--no-config3"],
}.items():
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    tail = [t for t in (r.stderr).strip().splitlines() if "free" in t or "Abort" in t or "rror" in t][-3:]
    print(f"{name}: rc={r.returncode} | " + " | ".join(t[:160] for t in tail), flush=True)
print("LD_LIBRARY_PATH =", os.environ.get("LD_LIBRARY_PATH"))
