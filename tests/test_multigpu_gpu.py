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


@pytest.mark.skipif("_ngpus() < 2", reason="needs two GPUs")
def test_two_rank_rccl_exchange_equals_single_process():
    r = _torchrun(2, "tools/dist_smoke.py")
    assert r.returncode == 0, r.stderr[-2000:]
    assert "dist smoke ok" in r.stdout and "world 2" in r.stdout
    assert "gather path uneven" in r.stdout


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
