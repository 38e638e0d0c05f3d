"""Synthetic clips generated directly in HBM with torch (bench inputs at full size: a 300-frame 4K
pair is 5 GB, too slow to synthesise with numpy on the host).  Same recipe as synth.py -- moving
sinusoids + drifting band-limited texture for the reference; blur mix + noise + 8x8 DC quantisation
for the distorted clip -- but not bit-identical to it; parity at these sizes is checked by pulling
sample frames back to the host and running the oracle on exactly those bytes."""
from __future__ import annotations

import math

import torch


def _texture(h: int, w: int, gen: torch.Generator, device) -> torch.Tensor:
    x = torch.randn((1, 1, h + 8, w + 8), generator=gen, device=device, dtype=torch.float32)
    k = torch.tensor([math.exp(-0.5 * (i / 1.5) ** 2) for i in range(-4, 5)], device=device)
    k = (k / k.sum()).view(1, 1, 1, 9)
    x = torch.nn.functional.conv2d(x, k)
    x = torch.nn.functional.conv2d(x, k.transpose(2, 3))
    x = x[0, 0, :h, :w]
    return x / x.std()


def make_clip_cuda(w: int, h: int, n: int, bit_depth: int = 8, seed: int = 20250418, device="cuda",
                   chroma: bool = False, t0: int = 0, progress=None):
    """Returns dict with 'ref' and 'dis': lists of per-plane tensors [n, ph, pw] (uint8 / int16-as-uint16 view)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    tex = _texture(h + 64, w + 64, gen, dev)
    yy, xx = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32),
                            torch.arange(w, device=dev, dtype=torch.float32), indexing="ij")
    u, v = xx / w, yy / h
    dtype = torch.uint8 if bit_depth <= 8 else torch.int16
    peak = float((1 << bit_depth) - 1)
    scale = float(1 << (bit_depth - 8))
    ref = torch.empty((n, h, w), dtype=dtype, device=dev)
    dis = torch.empty((n, h, w), dtype=dtype, device=dev)
    box = torch.full((1, 1, 5, 5), 1.0 / 25.0, device=dev)
    two_pi = 2 * math.pi
    for i in range(n):
        t = t0 + i
        # the noise of frame t is a function of (seed, t) alone: a rank that generates frames a .. b of a sharded job gets the
        # very frames the one-rank job has there (a generator that simply runs on would give every rank the noise of frames 0 ..)
        gen.manual_seed(seed * 1000003 + 1 + t)
        s = (torch.sin(two_pi * (3.0 * u + 0.011 * t)) * torch.cos(two_pi * (2.0 * v - 0.007 * t))
             + 0.6 * torch.sin(two_pi * (7.0 * u + 5.0 * v + 0.017 * t))
             + 0.4 * torch.cos(two_pi * (13.0 * u - 11.0 * v - 0.013 * t)))
        oy, ox = (t * 1) % 64, (t * 2) % 64
        r = (128.0 + 26.0 * s + 25.0 * tex[oy:oy + h, ox:ox + w]).clamp_(16.0, 235.0)
        rb = torch.nn.functional.conv2d(torch.nn.functional.pad(r[None, None], (2, 2, 2, 2), mode="reflect"), box)[0, 0]
        d = 0.5 * r + 0.5 * rb + 3.0 * torch.randn((h, w), generator=gen, device=dev)
        hb, wb = h // 8 * 8, w // 8 * 8
        blk = d[:hb, :wb].reshape(hb // 8, 8, wb // 8, 8)
        m = blk.mean(dim=(1, 3), keepdim=True)
        blk += torch.round(m / 6.0) * 6.0 - m
        d[:hb, :wb] = blk.reshape(hb, wb)
        if bit_depth <= 8:
            ref[i] = r.round().clamp_(0, 255).to(torch.uint8)
            dis[i] = d.round().clamp_(0, 255).to(torch.uint8)
        else:
            lsb = torch.randint(0, int(scale), (h, w), generator=gen, device=dev).float()
            ref[i] = (r * scale + lsb).round().clamp_(0, peak).to(torch.int16)
            dis[i] = (d * scale + lsb).round().clamp_(0, peak).to(torch.int16)
    out = {"ref": [ref], "dis": [dis]}
    if progress:
        torch.cuda.synchronize(dev)
        progress("luma planes generated")
    if chroma:
        cw, ch = (w + 1) // 2, (h + 1) // 2
        cy, cx = torch.meshgrid(torch.arange(ch, device=dev, dtype=torch.float32),
                                torch.arange(cw, device=dev, dtype=torch.float32), indexing="ij")
        for ph in (0.0, 0.37):
            cr = torch.empty((n, ch, cw), dtype=dtype, device=dev)
            cd = torch.empty((n, ch, cw), dtype=dtype, device=dev)
            for i in range(n):
                gen.manual_seed(seed * 1000003 + (500000 if ph == 0.0 else 750000) + t0 + i)
                c = 128.0 + 20.0 * torch.sin(two_pi * (1.5 * cx / cw + cy / ch + 0.005 * (t0 + i) + ph))
                e = c + 1.5 * torch.randn((ch, cw), generator=gen, device=dev)
                if bit_depth <= 8:
                    cr[i] = c.round().clamp_(0, 255).to(torch.uint8)
                    cd[i] = e.round().clamp_(0, 255).to(torch.uint8)
                else:
                    cr[i] = (c * scale).round().clamp_(0, peak).to(torch.int16)
                    cd[i] = (e * scale).round().clamp_(0, peak).to(torch.int16)
            out["ref"].append(cr)
            out["dis"].append(cd)
            if progress:
                torch.cuda.synchronize(dev)
                progress(f"chroma plane {len(out['ref']) - 1} generated")
    return out
