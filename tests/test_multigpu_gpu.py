"""The multi-GPU path on real devices (SiftMatchingOptions.gpu_index,
/root/reference/pycolmap/pipeline/match_features.h:76-81).  The N > 1 logic is covered on CPU by the gloo tests
(tests/test_distributed_cpu.py); these run the same functions over RCCL:

* with ONE visible GPU (the round-end test box): a one-rank RCCL run through torch.distributed.run - process group,
  device-resident exchange, reassembly - so that the harness itself is known to work;
* with TWO or more (skipped otherwise - they run the day such a box is handed to the suite): two ranks against the
  single-process result, `bench.py --gpus 2` on the sharded configuration, and `gpu_index="0,1"` through the drop-in
  API against `gpu_index="0"`, byte for byte."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ngpus() -> int:
    import torch
    return torch.cuda.device_count()


def _torchrun(nproc: int, script: str, *args: str, timeout: int = 600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=str(ROOT))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script, *args]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_one_rank_rccl_exchange_through_torchrun():
    r = _torchrun(1, "tools/dist_smoke.py")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "dist smoke ok" in r.stdout and "world 1" in r.stdout
    assert "gather path uneven" in r.stdout   # the RCCL branch of distributed._gather_rows, not the gloo padding
    assert "C ABI exchange" in r.stdout       # ... and amc_allgather_match_tables (RCCL called by the library) agreed with it


def test_one_rank_exchange_through_the_c_abi_without_torch_distributed(amc_ctx):
    """amc_comm_* / amc_allgather_match_tables with a world of one, no process group anywhere: the library's own RCCL
    calls (ncclCommInitRank, ncclAllGather of the sizes), the reorder into the global CSR, the download."""
    from pycolmap_amd import _capi, synth
    rng = np.random.default_rng(12)
    imgs = synth.scene_images(rng, 5, 640)
    amc_ctx.reserve_slots(len(imgs))
    for k, im in enumerate(imgs):
        amc_ctx.upload_descriptors(k, im)
    s1, s2 = synth.exhaustive_pairs(len(imgs))
    off, m, _ = amc_ctx.match_pairs(s1, s2)
    assert len(m) > 100
    comm = amc_ctx.comm_create(1, 0, _capi.comm_unique_id())
    perm = np.random.default_rng(1).permutation(len(s1)).astype(np.uint64)    # the caller's global order differs from the rank's
    g_off, g_m, st = comm.allgather_match_tables(perm, off, None)              # rows from the resident table
    assert st["world_size"] == 1 and st["num_matches"] == len(m) and st["rows_sent"] == 0 and st["device_ptr"] != 0
    h_off, h_m, _ = comm.allgather_match_tables(perm, off, m)                  # rows from the host
    np.testing.assert_array_equal(g_off, h_off)
    np.testing.assert_array_equal(g_m, h_m)
    for k, g in enumerate(perm):
        np.testing.assert_array_equal(g_m[int(g_off[g]):int(g_off[g + 1])], m[int(off[k]):int(off[k + 1])])
    a_off, a_m, _ = comm.allgather_match_tables(None, off, None)
    np.testing.assert_array_equal(a_off, off)
    np.testing.assert_array_equal(a_m, m)
    with pytest.raises(_capi.AmcError) as ei:                                  # positions must be a permutation
        comm.allgather_match_tables(np.zeros(len(s1), np.uint64), off, m)
    assert ei.value.code == _capi.AMC_E_INVALID
    with pytest.raises(_capi.AmcError):                                        # offsets must describe the resident table
        bad = off.copy()
        bad[-1] += 1
        comm.allgather_match_tables(perm, bad, None)
    comm.close()


def _fake_rccl() -> Path:
    """tests/shim/fake_rccl.cc -> tests/shim/_build/libfakerccl.so (built here when missing or older than its source)."""
    src = ROOT / "tests" / "shim" / "fake_rccl.cc"
    out = ROOT / "tests" / "shim" / "_build" / "libfakerccl.so"
    if not out.exists() or out.stat().st_mtime < src.stat().st_mtime:
        out.parent.mkdir(exist_ok=True)
        subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        str(src), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", str(out)], check=True)
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
def test_several_ranks_of_the_c_abi_exchange_on_one_device(world):
    """The N > 1 logic of amc_allgather_match_tables with N ranks as threads on the ONE device this box has: RCCL itself
    refuses two ranks per device, so a stand-in transport (tests/shim/fake_rccl.cc: rendezvous + device-to-device copies,
    loaded through AMC_RCCL_LIBRARY) carries the bytes; displacements, exact per-rank counts, the reorder into the
    global CSR, appended lists, downloads on one rank and the poisoned size exchange are the product's
    (tests/comm_threads.py checks them against the single-context result)."""
    env = dict(os.environ, AMC_RCCL_LIBRARY=str(_fake_rccl()), PYTHONPATH=str(ROOT))
    # world 8 = the node the exchange is written for: 5 images are 10 pairs, so at least one rank of eight is empty
    r = subprocess.run([sys.executable, "tests/comm_threads.py", str(world), "--images", "7" if world == 2 else "5"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert f"comm threads ok: world {world}" in r.stdout


@pytest.mark.parametrize("world,images,chunk", [(2, 6, None), (3, 6, "5"), (8, 5, None), (8, 7, "64")])
def test_verification_half_of_the_exchange_on_one_device(world, images, chunk):
    """amc_allgather_pair_records + amc_allgather_inlier_tables + amc_allgather_match_tables of a sharded match + verify
    run (tests/comm_threads.py --verify): the geometries and the inlier lists of every rank from device memory, in the
    global pair order, equal to the single-context call; empty ranks (8 ranks, 10 pairs); the allocation-failure
    agreement (every rank returns AMC_E_NOMEM, the communicator survives); a record size that differs between ranks and
    a short row array on one rank as collective errors.  chunk: AMC_COMM_CHUNK_WORDS - the pieces a transfer is cut into
    (1 GiB in production, so that no byte count above 2^31 reaches a transport call) are exercised with small inputs."""
    env = dict(os.environ, AMC_RCCL_LIBRARY=str(_fake_rccl()), PYTHONPATH=str(ROOT))
    if chunk:
        env["AMC_COMM_CHUNK_WORDS"] = chunk
    r = subprocess.run([sys.executable, "tests/comm_threads.py", str(world), "--images", str(images), "--verify"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert f"comm threads verify ok: world {world}" in r.stdout


def test_more_ranks_than_pairs_on_one_device():
    """Two images = one pair over two ranks: one rank's table is empty (no rows sent, nothing received from it)."""
    env = dict(os.environ, AMC_RCCL_LIBRARY=str(_fake_rccl()), PYTHONPATH=str(ROOT))
    r = subprocess.run([sys.executable, "tests/comm_threads.py", "2", "--images", "2"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "comm threads ok: world 2, 1 pairs as" in r.stdout


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")
def test_two_rank_rccl_exchange_equals_single_process():
    r = _torchrun(2, "tools/dist_smoke.py")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "dist smoke ok" in r.stdout and "world 2" in r.stdout
    assert "gather path uneven" in r.stdout and "C ABI exchange" in r.stdout


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")
def test_two_ranks_with_an_empty_rank_take_the_same_gather_path():
    """A rank whose shard is empty (more ranks than pairs with matches) sends its dummy row through the same uneven
    all-gather: no second code path on RCCL."""
    r = _torchrun(2, "tools/dist_smoke.py", "--images", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "dist smoke ok" in r.stdout and "gather path uneven" in r.stdout


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")
def test_bench_two_gpus_sharded_configuration():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "3", "--images", "64", "--steps", "1",
                        "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "strong"
    cfg = line["config"]
    assert cfg["rccl_ranks"] == 2 and len(cfg["kernel_ms_per_step_by_rank"]) == 2 and cfg["exchange_ms_per_step"] >= 0
    assert cfg["gather_path"] == "c-abi" and cfg["gpu_vs_oracle_mismatching_pairs"] == 0


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")
def test_bench_two_gpus_default_line_carries_both_curves():
    """`python bench.py --gpus 2`: the weak-scaled configs[1] headline and the configs[3] strong-scaling leg in one line,
    both exchanging through amc_allgather_match_tables."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--images", "48", "--feats", "1024", "--steps", "1",
                        "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0 and line["config"]["gather_path"] == "c-abi"
    l3 = line["config3"]
    assert l3["scaling"] == "strong" and l3["value"] == line["config3_value"] > 0 and l3["rccl_ranks"] == 2
    assert l3["gather_path"] == "c-abi" and l3["gpu_vs_oracle_mismatching_pairs"] == 0


def test_bench_one_gpu_line_carries_the_configs3_leg_through_the_c_abi():
    """The same line at N = 1 (reduced sizes): the configs[3] leg runs its exchange step through the library's
    communicator with one rank, and its sample of pairs agrees with the oracle."""
    r = subprocess.run([sys.executable, "bench.py", "--images", "24", "--feats", "1024", "--steps", "1", "--warmup", "0",
                        "--verify-pairs", "0", "--no-pipeline", "--no-ragged", "--no-sift-stats", "--no-dense", "--no-db"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    l3 = line["config3"]
    assert line["n_gpus"] == 1 and l3["rccl_ranks"] == 1 and l3["gather_path"] == "c-abi" and l3["value"] > 0
    assert l3["gpu_vs_oracle_mismatching_pairs"] == 0 and l3["pairs_checked"] > 0
    assert line["cpu_baseline"]["optimised_value"] is None or line["cpu_baseline"]["optimised_value"] > 0
    assert line["cpu_baseline"]["default_cpu_matcher_pairs_per_s"] > 0 and line["roofline"]["whole_step_frac"] > 0


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")
def test_gpu_index_two_devices_writes_the_same_database(tmp_path):
    import colmap_db
    import pycolmap
    from pycolmap_amd import synth
    from test_pipeline_gpu import _dump_tables
    rng = np.random.default_rng(8)
    images = synth.multiview_scene(rng, num_images=6, n_feats=512) + synth.multiview_scene(rng, num_images=4, n_feats=700)
    for k, im in enumerate(images):
        im["name"] = f"im{k:03d}.jpg"
        im["prior"] = k % 2 == 0
    dumps = {}
    for tag, idx in (("single", "0"), ("two", "0,1"), ("all", "-1")):
        db = tmp_path / f"{tag}.db"
        colmap_db.create(db, images)
        pycolmap.match_exhaustive(db, sift_options={"gpu_index": idx, "guided_matching": True},
                                  matching_options={"block_size": 5}, verification_options={"compute_relative_pose": True})
        dumps[tag] = _dump_tables(db)
    assert dumps["single"] == dumps["two"] == dumps["all"]
    assert len(dumps["single"][0]) == 45
