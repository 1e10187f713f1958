#!/usr/bin/env python3
"""round 6: the FIRST match call of a fresh context on the dense set (530 MB of matches): how much of it is the growth of
the result buffers?   python tools/r06_first_call_dense.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pycolmap_amd import _capi, synth
import bench
dev = torch.device("cuda", 0)
for trial in range(3):
    arena = bench.make_arena_torch(500, 4096, seed=1, device=dev, overlap="all")
    ctx = _capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.reserve_slots(500)
    for i in range(500):
        ctx.upload_descriptors_device(i, arena[i].data_ptr(), 4096)
    torch.cuda.synchronize()
    s1, s2 = synth.exhaustive_pairs(500)
    walls = []
    for k in range(3):
        r = None
        t = time.perf_counter(); r = ctx.match_pairs(s1, s2, copy=False); walls.append(time.perf_counter() - t)
    print("trial", trial, "matches", int(r[0][-1]), "call walls ms", [round(1e3 * w, 1) for w in walls], flush=True)
    r = None
    ctx.close()
    del arena
