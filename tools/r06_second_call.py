#!/usr/bin/env python3
"""round 6: wall time of the first calls of a fresh context (one copy=True call, then result views), with the library's
own host phases (AMC_MATCH_PROFILE=1) - where does a slow early call spend its time?   python tools/r06_second_call.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pycolmap_amd import _capi, synth
import bench
for trial in range(4):
    arena = bench.make_arena_torch(500, 4096, seed=1 + trial, device=torch.device("cuda", 0))
    ctx = _capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.reserve_slots(500)
    for i in range(500):
        ctx.upload_descriptors_device(i, arena[i].data_ptr(), 4096)
    torch.cuda.synchronize()
    s1, s2 = synth.exhaustive_pairs(500)
    walls = []
    t = time.perf_counter(); r = ctx.match_pairs(s1, s2); walls.append(time.perf_counter() - t); r = None
    for k in range(5):
        r = None
        t = time.perf_counter(); r = ctx.match_pairs(s1, s2, copy=False); walls.append(time.perf_counter() - t)
    print("trial", trial, "call walls ms", [round(1e3 * w, 1) for w in walls], flush=True)
    r = None
    ctx.close()
    del arena
