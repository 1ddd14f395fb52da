"""Single-frame extraction latency through gh_orb_extract_host (what FeatureDetector::detectAndCompute calls):
   python tools/orb_latency_probe.py      run with GSLAM_HIP_ORB_GRAPH=0 / 1 for the A/B."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gslam_amd import hip  # noqa: E402
from gslam_amd.orb import OrbExtractor, synth_frames  # noqa: E402

ctx = hip.Context()
for (w, h, K) in ((640, 480, 1000), (1241, 376, 2000), (1920, 1080, 2000)):
    ex = OrbExtractor(ctx, w, h, max_batch=1, n_features=K)
    img = synth_frames(ctx, 1, w, h)[0].cpu().numpy()
    for _ in range(20):
        ex.extract_host(img)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter()
        ex.extract_host(img)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print("graph=%s %dx%d K=%d: p50 %.3f ms  p99 %.3f ms" % (os.environ.get("GSLAM_HIP_ORB_GRAPH", "default"), w, h, K,
                                                             np.percentile(ts, 50), np.percentile(ts, 99)), flush=True)
    ex.close()
