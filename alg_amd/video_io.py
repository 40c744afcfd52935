"""Output writer (reference `run.py:121-133`: uint8 THWC frames -> h264 mp4 through
torchvision / PyAV).  Neither an h264 encoder nor a video I/O package exists in this environment, so the frames the HIP VAE
decoder produces ([T, H, W, 3] uint8, `output_type="uint8"`) are written in containers that need nothing but PIL:

    *.npy            the raw array (lossless; what `run.py` falls back to)
    *.avi            Motion-JPEG in a RIFF/AVI container (plays in ffmpeg / VLC / browsers' <video> via transcoding)
    *.gif            animated GIF (previews)
    a directory/     one PNG per frame, frame_00000.png ...

`*.mp4` raises with that explanation instead of silently writing something else.
"""
import io
import os
import struct

import numpy as np


def _as_frames(frames):
    a = frames.detach().cpu().numpy() if hasattr(frames, "detach") else np.asarray(frames)
    if a.ndim != 4 or a.shape[-1] != 3 or a.dtype != np.uint8:
        raise ValueError("frames must be uint8 [T, H, W, 3] (got %s %s)" % (a.dtype, a.shape))
    return a


def _chunk(tag, payload):
    return tag + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")


def write_mjpeg_avi(path, frames, fps=8, quality=95):
    """Minimal AVI 1.0 writer: one MJPG video stream, idx1 index."""
    from PIL import Image
    a = _as_frames(frames)
    T, H, W, _ = a.shape
    jpegs = []
    for f in a:
        buf = io.BytesIO()
        Image.fromarray(f).save(buf, format="JPEG", quality=quality)
        jpegs.append(buf.getvalue())
    usec = int(round(1e6 / fps))
    max_bytes = max(len(j) for j in jpegs)
    avih = struct.pack("<14I", usec, max_bytes * fps, 0, 0x10, T, 0, 1, max_bytes, W, H, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, fps, 0, T, max_bytes, 0xFFFFFFFF, 0) + \
        struct.pack("<4H", 0, 0, W, H)
    strf = struct.pack("<IiiHH4sIiiII", 40, W, H, 1, 24, b"MJPG", W * H * 3, 0, 0, 0, 0)
    hdrl = b"hdrl" + _chunk(b"avih", avih) + _chunk(b"LIST", b"strl" + _chunk(b"strh", strh) + _chunk(b"strf", strf))
    movi, index, off = b"movi", b"", 4
    for j in jpegs:
        c = _chunk(b"00dc", j)
        index += b"00dc" + struct.pack("<III", 0x10, off, len(j))
        movi += c
        off += len(c)
    body = b"AVI " + _chunk(b"LIST", hdrl) + _chunk(b"LIST", movi) + _chunk(b"idx1", index)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)
    return path


def read_mjpeg_avi(path):
    """Frames of a file written by `write_mjpeg_avi` (used by the tests; walks the movi list)."""
    from PIL import Image
    data = open(path, "rb").read()
    if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        raise ValueError("not a RIFF/AVI file")
    pos, frames = 12, []
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        if tag == b"LIST" and data[pos + 8:pos + 12] == b"movi":
            p, end = pos + 12, pos + 8 + size
            while p + 8 <= end:
                t, n = data[p:p + 4], struct.unpack("<I", data[p + 4:p + 8])[0]
                if t == b"00dc":
                    frames.append(np.asarray(Image.open(io.BytesIO(data[p + 8:p + 8 + n])).convert("RGB")))
                p += 8 + n + (n & 1)
        pos += 8 + size + (size & 1)
    return np.stack(frames)


def write_video(path, frames, fps=8):
    """Dispatch on the extension (see the module docstring)."""
    a = _as_frames(frames)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        np.save(path, a)
    elif ext == ".avi":
        write_mjpeg_avi(path, a, fps)
    elif ext == ".gif":
        from PIL import Image
        ims = [Image.fromarray(f) for f in a]
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=int(round(1000 / fps)), loop=0)
    elif ext == "":
        from PIL import Image
        os.makedirs(path, exist_ok=True)
        for i, f in enumerate(a):
            Image.fromarray(f).save(os.path.join(path, "frame_%05d.png" % i))
    elif ext == ".mp4":
        raise RuntimeError("no h264 encoder (torchvision / PyAV / ffmpeg) is available in this environment: write '.avi' "
                           "(Motion-JPEG), '.npy', '.gif' or a directory of PNG frames instead (reference run.py:126-132)")
    else:
        raise ValueError("unknown output extension %r" % ext)
    return path
