import sys, time, subprocess
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
if len(sys.argv) > 1:
    from pycolmap_amd import _capi, synth
    import test_verify_gpu as T
    idx = int(sys.argv[1])
    rng = np.random.default_rng(2)
    cfgs=[(20, 5, False), (14, 0, False), (15, 0, False), (700, 300, False), (33, 31, True), (64, 0, False), (65, 63, False), (128, 128, True)]
    priors = [False, False, True, False, True, False, True, False]
    scenes = [synth.two_view_scene(rng, num_inliers=ni, num_outliers=no, planar=pl) for ni, no, pl in cfgs]
    ctx = _capi.Context(0)
    sc, pr = scenes[idx], priors[idx]
    slots, cams = T.build_batch([sc], [pr])
    ctx.reserve_slots(2)
    for i, (kp, cam) in enumerate(zip(slots, cams)):
        ctx.upload_keypoints(i, kp.astype(np.float32)); ctx.upload_camera(i, cam["model"], cam["width"], cam["height"], cam["params"], cam["prior"])
    t0 = time.time()
    tv, mk, st = ctx.verify_pairs([0], [1], [0, len(sc["matches"])], sc["matches"])
    print(idx, cfgs[idx], pr, "wall %.3f s kernel %.1f ms trials %s inl %s cfg %d" % (time.time() - t0, st["kernel_ms"], tv["num_trials"][0], tv["model_inliers"][0], tv["config"][0]), flush=True)
else:
    for i in range(8):
        try:
            r = subprocess.run([sys.executable, __file__, str(i)], capture_output=True, text=True, timeout=40)
            print(r.stdout.strip()[-300:] or r.stderr.strip()[-300:], flush=True)
        except subprocess.TimeoutExpired:
            print(i, "TIMEOUT >40s", flush=True)
