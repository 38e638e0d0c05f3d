import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from pqa2_amd import _native as N, model as M, synth_torch
from pqa2_amd.engine import FeatureEngine
for (w,h,mn) in [(1920,1080,"vmaf_v0.6.1"),(3840,2160,"vmaf_4k_v0.6.1")]:
    F=300
    clip=synth_torch.make_clip_cuda(w,h,F,8)
    R,D=clip["ref"][0],clip["dis"][0]; torch.cuda.synchronize()
    model=M.load_model(mn)
    eng=FeatureEngine(w,h,max_batch=32,result_capacity=1024)
    ts=np.zeros(5)
    for it in range(6):
        t0=time.perf_counter(); eng.reset()
        t1=time.perf_counter(); eng.submit_resident(0,F,[R.data_ptr()],[D.data_ptr()],[w],[w*h])
        t2=time.perf_counter(); rec=eng.collect(0,F)
        t3=time.perf_counter(); met=M.metrics_from_records(rec,w,h,"integer_"); sc=M.score_frames(model,met); p=M.pool(sc["vmaf"])
        t4=time.perf_counter()
        if it>0: ts+=np.array([t1-t0,t2-t1,t3-t2,t4-t3,t4-t0])
    print(w,h,"ms: reset %.3f submit(launch) %.3f collect(wait+D2H) %.3f epilogue %.3f total %.3f -> fps %.0f"%(*(ts/5*1e3), F/(ts[4]/5)))
    eng.close(); del R,D,clip; torch.cuda.empty_cache()
