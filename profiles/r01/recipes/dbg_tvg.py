import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ["AMC_TVG_PROFILE"] = "1"
import numpy as np
sys.argv = ["bench"]
import bench
from pycolmap_amd import _capi
for n in (1024, 4096):
    out = bench.verify_leg(lambda: _capi.Context(0), 0, n, 1, 1, 0)
    print(n, "pairs/s %.0f" % out["value"], "ms", out["ms_per_step"], out["mean_trials_E_F_H"], flush=True)
