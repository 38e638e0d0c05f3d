#!/usr/bin/env python3
"""Create / score / destroy a context 150 times: the records must not change and neither device memory nor the
process RSS may grow (run it with PQA_STAGING_CACHE=0 as well: pinned staging is then allocated and freed per context)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pqa2_amd import _native as N, synth
from pqa2_amd.engine import FeatureEngine
w, h = 1280, 720
refs, diss = synth.make_clip(w, h, 3, 8, chroma=True)
def cycle(surf=False):
    with FeatureEngine(w, h, n_planes=3, features=N.FEAT_ALL, max_batch=4) as eng:
        for i in range(3):
            eng.submit(i, refs[i], diss[i])
        r = eng.collect(0, 3)
    return r
base = cycle()
torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
import resource
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for k in range(150):
    r = cycle()
    assert np.array_equal(r.view(np.uint64), base.view(np.uint64))
torch.cuda.synchronize()
free1, _ = torch.cuda.mem_get_info()
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print(f"device free before {free0 >> 20} MiB, after 150 create/score/destroy cycles {free1 >> 20} MiB (delta {(free0 - free1) >> 20} MiB); max RSS {rss0 >> 10} -> {rss1 >> 10} MiB")
