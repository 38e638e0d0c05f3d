"""Oracle-backed stand-in for pqa2_amd.engine.FeatureEngine, for CPU-only tests of the HOST logic
(file I/O, sharding + halo + gather, SVM, writers, analyzer signals).  Test infrastructure: it is
injected explicitly through `engine_factory`; the product path never falls back to it."""
import numpy as np

from oracle.oracle import Oracle
from pqa2_amd import _native as N


class OracleEngine:
    def __init__(self, width, height, bit_depth=8, n_planes=1, chroma_shift=(1, 1), features=N.FEAT_VMAF,
                 device=0, max_batch=8, result_capacity=16384, n_subsample=1,
                 vif_enhn_gain_limit=100.0, adm_enhn_gain_limit=100.0, vif_border=0):
        self.o = Oracle("f32")
        self.bpc, self.n_planes, self.features, self.k = bit_depth, n_planes, features, max(1, n_subsample)
        self.gl = (vif_enhn_gain_limit, adm_enhn_gain_limit, bool(vif_border))
        self.rec, self.prev_blur, self.cancelled = {}, None, False

    def set_motion_halo(self, prev):
        self.prev_blur = None if prev is None else self.o._blur_only(np.ascontiguousarray(prev), self.bpc)[1]

    def submit(self, index, ref_planes, dis_planes):
        if self.cancelled:
            raise N.PqaCancelled(N.PQA_ECANCELLED, "cancelled")
        r = np.zeros(N.RECORD_DOUBLES)
        feat, self.prev_blur = self.o.frame_features(np.ascontiguousarray(ref_planes[0]), np.ascontiguousarray(dis_planes[0]),
                                                     self.bpc, self.prev_blur, *self.gl)
        if index % self.k == 0:
            r[:16] = feat[:16]
        r[16] = feat[16]
        sse = np.zeros(3, np.uint64)
        for p in range(self.n_planes):
            if self.features & N.FEAT_PSNR:
                sse[p] = self.o.sse_plane(dis_planes[p], ref_planes[p], self.bpc)
            if self.features & N.FEAT_SSIM:
                r[N.REC_SSIM + p] = self.o.ssim_plane(dis_planes[p], ref_planes[p], self.bpc)
        r[N.REC_SSE:N.REC_SSE + 3] = sse.view(np.float64)
        self.rec[index] = r

    def collect(self, first, count):
        return np.stack([self.rec[first + i] for i in range(count)]) if count else np.zeros((0, N.RECORD_DOUBLES))

    def cancel(self):
        self.cancelled = True

    def close(self):
        pass
