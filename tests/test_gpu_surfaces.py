"""pqa_submit_surfaces: frames handed over as a hardware decoder writes them (NV12 / P010 / P012) must give exactly the
records the same samples give as planar frames -- the ingest kernels only move and shift bytes.  There is no decoder in
the image (rocDecode / VA-API absent), so the surfaces are built from the synthetic 4:2:0 clip on the device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _surfaces(planes, bpc, luma_pitch, chroma_pitch, torch):
    """planes: list of n frames, each [Y, U, V] numpy arrays -> (luma tensor [n, h, luma_pitch], chroma tensor
    [n, h/2, chroma_pitch]) in surface layout (samples MSB-aligned above 8 bit, U and V interleaved), on the device."""
    n = len(planes)
    h, w = planes[0][0].shape
    ch, cw = planes[0][1].shape
    dt = np.uint8 if bpc <= 8 else np.uint16
    sh = 0 if bpc <= 8 else 16 - bpc
    # the bytes the library must never read: a pattern, not zeros
    L = np.full((n, h, luma_pitch), 0xA5 if bpc <= 8 else 0xA5A5, dt)
    Cq = np.full((n, ch, chroma_pitch), 0x5A if bpc <= 8 else 0x5A5A, dt)
    for i, (y, u, v) in enumerate(planes):
        L[i, :, :w] = y.astype(dt) << sh
        Cq[i, :, 0:2 * cw:2] = u.astype(dt) << sh
        Cq[i, :, 1:2 * cw:2] = v.astype(dt) << sh
    as_t = (lambda a: torch.from_numpy(a.view(np.int16) if bpc > 8 else a).cuda())
    return as_t(L), as_t(Cq)


@pytest.mark.parametrize("w,h,bpc,pad", [(640, 360, 8, 0), (642, 362, 8, 6), (1920, 1080, 8, 64), (640, 360, 10, 0),
                                          (322, 182, 10, 5), (352, 288, 12, 16)])
def test_surfaces_give_the_planar_records(w, h, bpc, pad):
    import torch
    from pqa2_amd import _native as N, synth
    from pqa2_amd.engine import FeatureEngine
    n = 5
    refs, diss = synth.make_clip(w, h, n, bpc, chroma=True)
    es = 1 if bpc <= 8 else 2
    cw, chh = refs[0][1].shape[1], refs[0][1].shape[0]
    lp, cp = w + pad, 2 * cw + pad          # pitches in SAMPLES; odd paddings exercise the unaligned kernels
    fmt = N.SURFACE_NV12 if bpc <= 8 else N.SURFACE_P01X
    kw = dict(bit_depth=bpc, n_planes=3, features=N.FEAT_ALL, max_batch=2)   # 3 batches: halves alternate, halo carried
    with FeatureEngine(w, h, **kw) as eng:
        for i in range(n):
            eng.submit(i, refs[i], diss[i])
        planar = eng.collect(0, n)
    RL, RC = _surfaces(refs, bpc, lp, cp, torch)
    DL, DC = _surfaces(diss, bpc, lp, cp, torch)
    torch.cuda.synchronize()
    mk = lambda L, Cq, f0=0: FeatureEngine.surface_clip(fmt, L.data_ptr() + f0 * h * lp * es, lp * es, h * lp * es,
                                                        Cq.data_ptr() + f0 * chh * cp * es, cp * es, chh * cp * es)
    with FeatureEngine(w, h, **kw) as eng:
        eng.submit_surfaces(0, n, mk(RL, RC), mk(DL, DC))
        surf = eng.collect(0, n)
    assert np.array_equal(surf.view(np.uint64), planar.view(np.uint64))
    # a frame-sharded rank: frames 2.. with frame 1 of the reference as the motion halo
    with FeatureEngine(w, h, **kw) as eng:
        eng.submit_surfaces(2, n - 2, mk(RL, RC, 2), mk(DL, DC, 2), prev_ref=mk(RL, RC, 1))
        shard = eng.collect(2, n - 2)
    assert np.array_equal(shard.view(np.uint64), planar[2:].view(np.uint64))
    # one context, three ways in: host planes, then surfaces (motion continues across the switch), then resident planes
    with FeatureEngine(w, h, **kw) as eng:
        for i in range(2):
            eng.submit(i, refs[i], diss[i])
        eng.submit_surfaces(2, 2, mk(RL, RC, 2), mk(DL, DC, 2))
        eng.submit(4, refs[4], diss[4])
        mixed = eng.collect(0, n)
    assert np.array_equal(mixed.view(np.uint64), planar.view(np.uint64))
    # luma-only context (VMAF): an NV12 luma plane is scored in place, no chroma pointer needed
    with FeatureEngine(w, h, bit_depth=bpc, max_batch=4) as eng:
        s = lambda L: FeatureEngine.surface_clip(fmt, L.data_ptr(), lp * es, h * lp * es)
        eng.submit_surfaces(0, n, s(RL), s(DL))
        luma_only = eng.collect(0, n)
    assert np.array_equal(luma_only[:, :17].view(np.uint64), planar[:, :17].view(np.uint64))


def test_surface_argument_errors():
    import torch
    from pqa2_amd import _native as N
    from pqa2_amd.engine import FeatureEngine
    w, h = 320, 180
    buf = torch.zeros(4 * h * w, dtype=torch.uint8, device="cuda")
    ok = FeatureEngine.surface_clip(N.SURFACE_NV12, buf.data_ptr(), w, h * w, buf.data_ptr(), w, h * w // 2)
    with FeatureEngine(w, h, n_planes=3, features=N.FEAT_ALL) as eng:
        bad = FeatureEngine.surface_clip(N.SURFACE_P01X, buf.data_ptr(), w, h * w, buf.data_ptr(), w, h * w // 2)
        with pytest.raises(N.PqaError) as e:
            eng.submit_surfaces(0, 1, bad, ok)
        assert e.value.code == N.PQA_EINVAL and "format" in str(e.value)
        short = FeatureEngine.surface_clip(N.SURFACE_NV12, buf.data_ptr(), w - 1, h * w, buf.data_ptr(), w, h * w // 2)
        with pytest.raises(N.PqaError):
            eng.submit_surfaces(0, 1, short, ok)
        nochroma = FeatureEngine.surface_clip(N.SURFACE_NV12, buf.data_ptr(), w, h * w)
        with pytest.raises(N.PqaError):
            eng.submit_surfaces(0, 1, ok, nochroma)
        eng.submit_surfaces(0, 1, ok, ok)          # and the context is still usable
        with pytest.raises(N.PqaError) as e:       # frame 0's record has not been collected: a frame that maps to its
            eng.submit_surfaces(16384, 1, ok, ok)  # ring slot (index + result_capacity) must not overwrite it
        assert e.value.code == N.PQA_ESTATE
        assert np.all(np.isfinite(eng.collect(0, 1)))
    with FeatureEngine(w, h, n_planes=3, chroma_shift=(0, 0), features=N.FEAT_ALL) as eng:   # 4:4:4 context
        with pytest.raises(N.PqaError):
            eng.submit_surfaces(0, 1, ok, ok)
