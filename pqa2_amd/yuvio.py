"""Planar YUV containers the scoring path reads: YUV4MPEG2 (.y4m) and headerless raw (.yuv).

The reference hands file paths to ffmpeg (app/vmaf_analyzer.py:411-419), which demuxes and
decodes them.  This engine has no decoder of its own; it consumes what a decoder emits --
planar Y, U, V at 8 or 10/12/16 bits (little-endian u16) -- from Y4M directly, from raw .yuv
given the geometry, or from an `ffmpeg ... -f yuv4mpegpipe -` child when a system ffmpeg exists.
"""
from __future__ import annotations

import io
import os
import re
from dataclasses import dataclass

import numpy as np

_Y4M_CHROMA = {
    # tag prefix -> (horizontal shift, vertical shift, bit depth, monochrome)
    "420jpeg": (1, 1, 8, False), "420mpeg2": (1, 1, 8, False), "420paldv": (1, 1, 8, False),
    "420": (1, 1, 8, False), "422": (1, 0, 8, False), "444": (0, 0, 8, False), "mono": (0, 0, 8, True),
    "420p10": (1, 1, 10, False), "422p10": (1, 0, 10, False), "444p10": (0, 0, 10, False),
    "420p12": (1, 1, 12, False), "422p12": (1, 0, 12, False), "444p12": (0, 0, 12, False),
    "420p16": (1, 1, 16, False), "422p16": (1, 0, 16, False), "444p16": (0, 0, 16, False),
    "mono10": (0, 0, 10, True), "mono12": (0, 0, 12, True), "mono16": (0, 0, 16, True),
}


@dataclass
class VideoInfo:
    width: int
    height: int
    bit_depth: int = 8
    hshift: int = 1            # chroma subsampling shifts (4:2:0 -> 1,1)
    vshift: int = 1
    mono: bool = False
    fps_num: int = 30
    fps_den: int = 1
    n_frames: int = 0
    chroma_tag: str = "420jpeg"
    color_range: str | None = None   # "limited" / "full" when the container says so (Y4M XCOLORRANGE=...), else None

    @property
    def dtype(self):
        return np.uint8 if self.bit_depth <= 8 else np.dtype("<u2")

    @property
    def bytes_per_sample(self) -> int:
        return 1 if self.bit_depth <= 8 else 2

    @property
    def chroma_w(self) -> int:
        return 0 if self.mono else -(-self.width >> self.hshift)

    @property
    def chroma_h(self) -> int:
        return 0 if self.mono else -(-self.height >> self.vshift)

    @property
    def plane_shapes(self):
        s = [(self.height, self.width)]
        if not self.mono:
            s += [(self.chroma_h, self.chroma_w)] * 2
        return s

    @property
    def frame_bytes(self) -> int:
        return sum(h * w for h, w in self.plane_shapes) * self.bytes_per_sample

    @property
    def pix_fmt(self) -> str:
        base = "gray" if self.mono else {(1, 1): "yuv420p", (1, 0): "yuv422p", (0, 0): "yuv444p"}[(self.hshift, self.vshift)]
        if self.bit_depth > 8:
            base += f"{self.bit_depth}le"
        return base

    @property
    def fps(self) -> float:
        return self.fps_num / self.fps_den if self.fps_den else 0.0


class Y4MReader:
    """Sequential/random-access reader of a YUV4MPEG2 file (memory-mapped, zero-copy planes)."""

    def __init__(self, path: str):
        self.path = path
        with open(path, "rb") as f:
            head = f.readline(4096)
        if not head.startswith(b"YUV4MPEG2"):
            raise ValueError(f"not a YUV4MPEG2 file: {path}")
        info = VideoInfo(0, 0)
        for tok in head.decode("ascii", "replace").strip().split(" ")[1:]:
            if not tok:
                continue
            k, v = tok[0], tok[1:]
            if k == "W":
                info.width = int(v)
            elif k == "H":
                info.height = int(v)
            elif k == "F":
                a, b = v.split(":")
                info.fps_num, info.fps_den = int(a), int(b)
            elif k == "C":
                if v not in _Y4M_CHROMA:
                    raise ValueError(f"unsupported Y4M chroma tag C{v}")
                info.hshift, info.vshift, info.bit_depth, info.mono = _Y4M_CHROMA[v]
                info.chroma_tag = v
            elif k == "X" and v.upper().startswith("COLORRANGE="):   # FFmpeg's Y4M extension
                info.color_range = {"FULL": "full", "LIMITED": "limited"}.get(v.split("=", 1)[1].upper())
        if info.width <= 0 or info.height <= 0:
            raise ValueError("Y4M header lacks W/H")
        self.info = info
        self._data_off = len(head)
        self._mm = np.memmap(path, dtype=np.uint8, mode="r")
        # frame markers: "FRAME" + optional params + "\n"; assume the common fixed 6-byte marker,
        # fall back to a scan when a frame header carries parameters
        fb = info.frame_bytes
        size = self._mm.shape[0]
        self._offsets = []
        off = self._data_off
        # the common case -- every frame header is the bare 6-byte "FRAME\n" -- is recognised from the first, the middle
        # and the last header of the frame count the file size implies: three page touches instead of one per frame
        n_plain = (size - off) // (fb + 6)
        if n_plain > 0 and all(bytes(self._mm[off + k * (fb + 6):off + k * (fb + 6) + 6]) == b"FRAME\n"
                               for k in {0, n_plain // 2, n_plain - 1}):
            self._offsets = [off + k * (fb + 6) + 6 for k in range(n_plain)]
            off = size   # nothing left to scan
        while off + 6 <= size:
            if bytes(self._mm[off:off + 5]) != b"FRAME":
                break
            nl = off + 5
            while nl < size and self._mm[nl] != 0x0A:
                nl += 1
            start = nl + 1
            if start + fb > size:
                break
            self._offsets.append(start)
            off = start + fb
        info.n_frames = len(self._offsets)

    def __len__(self):
        return self.info.n_frames

    def close(self):
        """Drops the mapping; removes the file too when it is a temporary decode (_decode_with_ffmpeg)."""
        self._mm = None
        fd, self._fd = getattr(self, "_fd", None), None
        if fd is not None:
            try:
                os.close(fd)
            except OSError:
                pass
        tmp, self._tempfile = getattr(self, "_tempfile", None), None
        if tmp:
            try:
                os.unlink(tmp)
            except OSError:
                pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fileno(self) -> int:
        """A read-only descriptor of the file (opened on first use, closed with the reader): with plane_offsets() it lets
        the library read frames straight into its staging (pqa_submit_fd) instead of copying them out of the mapping."""
        if getattr(self, "_fd", None) is None:
            self._fd = os.open(self.path, os.O_RDONLY)
        return self._fd

    def plane_offsets(self, i: int):
        """Byte offset of each plane of frame i in the file (planes are packed: rows width * sample-size bytes apart)."""
        off = self._offsets[i]
        out = []
        for (h, w) in self.info.plane_shapes:
            out.append(off)
            off += h * w * self.info.bytes_per_sample
        return out

    def run_stride(self, i: int, n: int):
        """Byte distance between consecutive frames i .. i + n - 1 when it is constant (it is unless FRAME headers carry
        parameters of varying length), else None: what pqa_submit_fd_run needs to take the run in one call."""
        if n <= 1:
            return 0
        d = self._offsets[i + 1] - self._offsets[i]
        return int(d) if all(self._offsets[k + 1] - self._offsets[k] == d for k in range(i, i + n - 1)) else None

    def frame(self, i: int):
        """Planes [Y, U, V] (or [Y]) of frame i as read-only arrays viewing the mapped file."""
        info = self.info
        off = self._offsets[i]
        planes = []
        for (h, w) in info.plane_shapes:
            nbytes = h * w * info.bytes_per_sample
            buf = self._mm[off:off + nbytes]
            planes.append(buf.view(info.dtype).reshape(h, w))
            off += nbytes
        return planes

    def __iter__(self):
        for i in range(len(self)):
            yield self.frame(i)


class RawYUVReader:
    """Headerless planar .yuv; geometry from arguments or a `_WxH` / `_WxH_10bit` name hint."""

    def __init__(self, path: str, width: int | None = None, height: int | None = None,
                 bit_depth: int | None = None, fps: int = 30):
        m = re.search(r"(\d{2,5})x(\d{2,5})", os.path.basename(path))
        if (width is None or height is None) and not m:
            raise ValueError(f"raw YUV needs a geometry (WxH in the file name or arguments): {path}")
        width = width or int(m.group(1))
        height = height or int(m.group(2))
        if bit_depth is None:
            mb = re.search(r"(\d{1,2})bit", os.path.basename(path))
            bit_depth = int(mb.group(1)) if mb else 8
        self.path = path
        self.info = VideoInfo(width, height, bit_depth, 1, 1, False, fps, 1)
        self._mm = np.memmap(path, dtype=np.uint8, mode="r")
        self.info.n_frames = self._mm.shape[0] // self.info.frame_bytes
        self._fd = None

    def __len__(self):
        return self.info.n_frames

    def fileno(self) -> int:
        if self._fd is None:
            self._fd = os.open(self.path, os.O_RDONLY)
        return self._fd

    def run_stride(self, i: int, n: int):
        return int(self.info.frame_bytes)

    def plane_offsets(self, i: int):
        off = i * self.info.frame_bytes
        out = []
        for (h, w) in self.info.plane_shapes:
            out.append(off)
            off += h * w * self.info.bytes_per_sample
        return out

    def __del__(self):
        try:
            if self._fd is not None:
                os.close(self._fd)
                self._fd = None
        except Exception:
            pass

    def frame(self, i: int):
        info = self.info
        off = i * info.frame_bytes
        planes = []
        for (h, w) in info.plane_shapes:
            nbytes = h * w * info.bytes_per_sample
            planes.append(self._mm[off:off + nbytes].view(info.dtype).reshape(h, w))
            off += nbytes
        return planes

    def __iter__(self):
        for i in range(len(self)):
            yield self.frame(i)


def write_y4m(path: str, frames, info: VideoInfo) -> None:
    """frames: iterable of [Y,U,V] plane lists matching `info`."""
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{info.width} H{info.height} F{info.fps_num}:{info.fps_den} Ip A1:1 C{info.chroma_tag}"
                f"{' XCOLORRANGE=' + info.color_range.upper() if info.color_range else ''}\n".encode())
        for planes in frames:
            f.write(b"FRAME\n")
            for p in planes:
                f.write(np.ascontiguousarray(p, dtype=info.dtype).tobytes())


def open_video(path: str, **raw_kwargs):
    """Reader for .y4m / .yuv; anything else is decoded by a system ffmpeg into a temp Y4M."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".y4m":
        return Y4MReader(path)
    if ext == ".yuv":
        return RawYUVReader(path, **raw_kwargs)
    with open(path, "rb") as f:
        if f.read(9) == b"YUV4MPEG2":
            return Y4MReader(path)
    return _decode_with_ffmpeg(path)


_DECODED: dict = {}   # (path, mtime, size) -> Y4MReader over the temporary decode; a run opens each clip several times


def _drop_decodes():
    while _DECODED:
        _DECODED.popitem()[1].close()


import atexit  # noqa: E402
atexit.register(_drop_decodes)


def _decode_with_ffmpeg(path: str):
    """Compressed containers need a decoder; the reference relies on ffmpeg for that too
    (app/vmaf_analyzer.py:411-419).  Only the *decode* is delegated -- never the metric filters.
    The decode is kept (at most 4 clips) because one analysis opens each clip for metadata and for scoring."""
    import shutil
    import subprocess
    import tempfile
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
    rd = _DECODED.get(key)
    if rd is not None and rd._mm is not None:
        return rd
    exe = shutil.which("ffmpeg")
    if not exe:
        raise RuntimeError(
            f"cannot decode {os.path.basename(path)}: no ffmpeg on PATH; supply .y4m or raw .yuv input")
    tmp = tempfile.NamedTemporaryFile(suffix=".y4m", delete=False)
    tmp.close()
    try:
        subprocess.run([exe, "-hide_banner", "-loglevel", "error", "-y", "-i", path, "-f", "yuv4mpegpipe",
                        "-strict", "-1", tmp.name], check=True)
        rd = Y4MReader(tmp.name)
    except Exception:
        os.unlink(tmp.name)
        raise
    rd._tempfile = tmp.name
    while len(_DECODED) >= 4:
        _DECODED.pop(next(iter(_DECODED))).close()
    _DECODED[key] = rd
    return rd
