"""Output writer (reference `run.py:121-133`: uint8 THWC frames -> h264 mp4 through
torchvision / PyAV).  No video I/O package or encoder library exists in this environment, so the containers and the
bitstream are written here, from the frames the HIP VAE decoder produces ([T, H, W, 3] uint8, `output_type="uint8"`):

    *.mp4            ISO-BMFF container with an H.264 (AVC) video track, as the reference writes (`video_codec="h264"`):
                     a Constrained-Baseline stream of IDR pictures whose macroblocks are all I_PCM, i.e. the BT.601
                     limited-range YUV 4:2:0 samples stored verbatim -- every H.264 decoder plays it, nothing is lost beyond
                     the 4:2:0 conversion libx264 would also do (the reference's crf 18 is lossy on top); the price is size
                     (1.5 bytes per pixel per frame: 25 MB for a 49-frame 480x720 video).  `codec="mjpeg"` writes the much
                     smaller Motion-JPEG-in-mp4 form (`mp4v` sample entry, object type 0x6C) instead.
    *.npy            the raw array (lossless)
    *.avi            Motion-JPEG in a RIFF/AVI container
    *.gif            animated GIF (previews)
    a directory/     one PNG per frame, frame_00000.png ...
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


# ---------------------------------------------------------------------------------------------------------------------
# H.264 (I_PCM) / Motion-JPEG in an ISO base media (mp4) container
# ---------------------------------------------------------------------------------------------------------------------
class _Bits:
    """MSB-first bit writer for the H.264 headers (ITU-T H.264 section 7.2: u(n), ue(v), se(v))."""

    def __init__(self):
        self.bits = []

    def u(self, n, v):
        self.bits.extend((v >> (n - 1 - i)) & 1 for i in range(n))
        return self

    def ue(self, v):
        n = (v + 1).bit_length()
        return self.u(n - 1, 0).u(n, v + 1)

    def se(self, v):
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def align_zero(self):
        while len(self.bits) % 8:
            self.bits.append(0)
        return self

    def trailing(self):  # rbsp_trailing_bits(): a stop bit, then zeros to the byte boundary
        self.bits.append(1)
        return self.align_zero()

    def bytes(self):
        assert len(self.bits) % 8 == 0
        return np.packbits(np.array(self.bits, dtype=np.uint8)).tobytes()


def _escape_rbsp(payload):
    """Emulation prevention (H.264 7.4.1): 00 00 0x -> 00 00 03 0x for x <= 3."""
    import re
    return re.sub(b"\x00\x00(?=[\x00-\x03])", b"\x00\x00\x03", payload)


def rgb_to_yuv420(frames):
    """uint8 [T, H, W, 3] (H, W even) -> (Y [T, H, W], Cb [T, H/2, W/2], Cr [T, H/2, W/2]) uint8, BT.601 limited range
    (what swscale gives libx264 for rgb24 input), chroma = mean of each 2x2 block."""
    a = frames.astype(np.float32)
    r, g, b = a[..., 0], a[..., 1], a[..., 2]
    y = 16.0 + (65.481 * r + 128.553 * g + 24.966 * b) / 255.0
    cb = 128.0 + (-37.797 * r - 74.203 * g + 112.0 * b) / 255.0
    cr = 128.0 + (112.0 * r - 93.786 * g - 18.214 * b) / 255.0
    T, H, W = y.shape
    sub = lambda c: c.reshape(T, H // 2, 2, W // 2, 2).mean(axis=(2, 4))
    q = lambda c, hi: np.clip(np.rint(c), 16, hi).astype(np.uint8)
    return q(y, 235), q(sub(cb), 240), q(sub(cr), 240)


def yuv420_to_rgb(y, cb, cr):
    """Inverse of `rgb_to_yuv420` up to the 4:2:0 subsampling and rounding (nearest-neighbour chroma); tests / readers."""
    yf = (y.astype(np.float32) - 16.0) * (255.0 / 219.0)
    up = lambda c: np.repeat(np.repeat(c.astype(np.float32) - 128.0, 2, axis=-2), 2, axis=-1) * (255.0 / 224.0)
    u, v = up(cb), up(cr)
    r = yf + 1.402 * v
    g = yf - 0.344136 * u - 0.714136 * v
    b = yf + 1.772 * u
    return np.clip(np.rint(np.stack([r, g, b], axis=-1)), 0, 255).astype(np.uint8)


def _h264_parameter_sets(W, H):
    """SPS / PPS NAL units (with their one-byte NAL headers) of the I_PCM stream for a W x H picture."""
    mbw, mbh = (W + 15) // 16, (H + 15) // 16
    level = 51 if mbw * mbh <= 36864 else 62
    sps = _Bits().u(8, 66).u(8, 0xC0).u(8, level)      # Baseline, constraint_set0/1 (constrained baseline), level
    sps.ue(0)                                            # seq_parameter_set_id
    sps.ue(0)                                            # log2_max_frame_num_minus4
    sps.ue(2)                                            # pic_order_cnt_type 2: output order = decoding order
    sps.ue(1).u(1, 0)                                    # max_num_ref_frames, gaps_in_frame_num_value_allowed_flag
    sps.ue(mbw - 1).ue(mbh - 1)
    sps.u(1, 1).u(1, 1)                                  # frame_mbs_only_flag, direct_8x8_inference_flag
    cw, ch = mbw * 16 - W, mbh * 16 - H
    if cw or ch:
        sps.u(1, 1).ue(0).ue(cw // 2).ue(0).ue(ch // 2)  # frame cropping, in chroma sample units (4:2:0)
    else:
        sps.u(1, 0)
    sps.u(1, 0).trailing()                               # no VUI
    pps = _Bits().ue(0).ue(0).u(1, 0).u(1, 0)            # ids, CAVLC, no bottom_field_pic_order
    pps.ue(0).ue(0).ue(0).u(1, 0).u(2, 0)                # one slice group, ref idx defaults, no weighted prediction
    pps.se(0).se(0).se(0)                                # pic_init_qp / qs - 26, chroma_qp_index_offset
    pps.u(1, 1).u(1, 0).u(1, 0).trailing()               # deblocking control present, no constrained intra, no redundant pics
    return b"\x67" + _escape_rbsp(sps.bytes()), b"\x68" + _escape_rbsp(pps.bytes()), level


def _h264_idr_picture(y, cb, cr, idr_pic_id):
    """One IDR access unit = one slice NAL (header byte 0x65) whose macroblocks are all I_PCM (mb_type 25: 9 bits of
    ue(v), zero bits to the byte boundary, 256 luma + 64 Cb + 64 Cr samples).  y [16*mbh, 16*mbw], cb/cr half size."""
    mbh, mbw = y.shape[0] // 16, y.shape[1] // 16
    n = mbh * mbw
    blk = np.empty((n, 386), dtype=np.uint8)
    blk[:, 0], blk[:, 1] = 0x0D, 0x00                    # 0000 1101 0|000 0000: ue(25) + pcm_alignment_zero_bits
    blk[:, 2:258] = y.reshape(mbh, 16, mbw, 16).transpose(0, 2, 1, 3).reshape(n, 256)
    blk[:, 258:322] = cb.reshape(mbh, 8, mbw, 8).transpose(0, 2, 1, 3).reshape(n, 64)
    blk[:, 322:386] = cr.reshape(mbh, 8, mbw, 8).transpose(0, 2, 1, 3).reshape(n, 64)
    hdr = _Bits().ue(0).ue(7).ue(0).u(4, 0)             # first_mb_in_slice, slice_type I (all), pps id, frame_num
    hdr.ue(idr_pic_id).u(1, 0).u(1, 0)                   # idr_pic_id; no_output_of_prior_pics, long_term_reference
    hdr.se(0).ue(1)                                      # slice_qp_delta, disable_deblocking_filter_idc = 1
    hdr.ue(25).align_zero()                              # first macroblock's mb_type + alignment
    body = hdr.bytes() + blk.reshape(-1)[2:].tobytes() + b"\x80"   # ... + rbsp_slice_trailing_bits
    return b"\x65" + _escape_rbsp(body)


def _box(tag, *parts):
    payload = b"".join(parts)
    return struct.pack(">I", 8 + len(payload)) + tag + payload


def _full(tag, version, flags, *parts):
    return _box(tag, struct.pack(">I", (version << 24) | flags), *parts)


_MATRIX = struct.pack(">9I", 0x10000, 0, 0, 0, 0x10000, 0, 0, 0, 0x40000000)


def _mp4_file(path, samples, W, H, fps, sample_entry):
    """ftyp + mdat + moov for one video track whose samples (bytes each) sit in one chunk."""
    T = len(samples)
    scale, delta = int(round(fps * 1000)), 1000
    dur = T * delta
    ftyp = _box(b"ftyp", b"isom", struct.pack(">I", 0x200), b"isomiso2avc1mp41")
    mdat = _box(b"mdat", *samples)
    if len(mdat) >= 1 << 32:
        raise ValueError("video too large for a 32-bit mdat box; write fewer frames per file")
    first = len(ftyp) + 8
    stbl = _box(b"stbl",
                _full(b"stsd", 0, 0, struct.pack(">I", 1), sample_entry),
                _full(b"stts", 0, 0, struct.pack(">III", 1, T, delta)),
                _full(b"stsc", 0, 0, struct.pack(">IIII", 1, 1, T, 1)),
                _full(b"stsz", 0, 0, struct.pack(">II", 0, T), struct.pack(">%dI" % T, *[len(s_) for s_ in samples])),
                _full(b"stco", 0, 0, struct.pack(">II", 1, first)))
    minf = _box(b"minf", _full(b"vmhd", 0, 1, struct.pack(">4H", 0, 0, 0, 0)),
                _box(b"dinf", _full(b"dref", 0, 0, struct.pack(">I", 1), _full(b"url ", 0, 1))), stbl)
    mdia = _box(b"mdia", _full(b"mdhd", 0, 0, struct.pack(">IIIIHH", 0, 0, scale, dur, 0x55C4, 0)),
                _full(b"hdlr", 0, 0, struct.pack(">I", 0), b"vide", b"\x00" * 12, b"VideoHandler\x00"), minf)
    tkhd = _full(b"tkhd", 0, 3, struct.pack(">IIIII", 0, 0, 1, 0, dur), b"\x00" * 8, struct.pack(">4H", 0, 0, 0, 0), _MATRIX,
                 struct.pack(">II", W << 16, H << 16))
    mvhd = _full(b"mvhd", 0, 0, struct.pack(">IIIIIH", 0, 0, scale, dur, 0x10000, 0x0100), b"\x00" * 10, _MATRIX,
                 b"\x00" * 24, struct.pack(">I", 2))
    moov = _box(b"moov", mvhd, _box(b"trak", tkhd, mdia))
    with open(path, "wb") as fh:
        fh.write(ftyp + mdat + moov)
    return path


def _visual_entry(tag, W, H, name, *children):
    comp = bytes([len(name)]) + name + b"\x00" * (31 - len(name))
    return _box(tag, b"\x00" * 6, struct.pack(">H", 1), b"\x00" * 16, struct.pack(">HHII", W, H, 0x480000, 0x480000),
                struct.pack(">IH", 0, 1), comp, struct.pack(">Hh", 0x18, -1), *children)


def write_mp4(path, frames, fps=8, codec="h264", quality=95):
    """`run.py:127-133` (`write_video(..., video_codec="h264")`): the frames as an mp4 file with one video track."""
    a = _as_frames(frames)
    T, H, W, _ = a.shape
    if codec == "mjpeg":
        from PIL import Image
        samples = []
        for f in a:
            buf = io.BytesIO()
            Image.fromarray(f).save(buf, format="JPEG", quality=quality)
            samples.append(buf.getvalue())
        desc = bytes([0x04, 13, 0x6C, 0x11, 0, 0, 0]) + struct.pack(">II", 0, 0)    # DecoderConfig: JPEG, visual stream
        es = bytes([0x03, 3 + len(desc) + 3, 0, 1, 0]) + desc + bytes([0x06, 1, 2])   # ES_Descr + SLConfig
        return _mp4_file(path, samples, W, H, fps, _visual_entry(b"mp4v", W, H, b"alg_amd mjpeg", _full(b"esds", 0, 0, es)))
    if codec != "h264":
        raise ValueError("mp4 codec must be 'h264' or 'mjpeg'")
    if H % 2 or W % 2:
        raise ValueError("4:2:0 video needs even height and width")
    ph, pw = (H + 15) // 16 * 16, (W + 15) // 16 * 16
    sps, pps, level = _h264_parameter_sets(W, H)
    samples = []
    for t in range(T):
        f = a[t:t + 1]
        if (ph, pw) != (H, W):   # pad to whole macroblocks by edge replication; the SPS crops it away again
            f = np.pad(f, ((0, 0), (0, ph - H), (0, pw - W), (0, 0)), mode="edge")
        y, cb, cr = rgb_to_yuv420(f)
        nal = _h264_idr_picture(y[0], cb[0], cr[0], t & 1)
        samples.append(struct.pack(">I", len(nal)) + nal)
    avcc = _box(b"avcC", bytes([1, 66, 0xC0, level, 0xFF, 0xE1]), struct.pack(">H", len(sps)), sps, bytes([1]),
                struct.pack(">H", len(pps)), pps)
    return _mp4_file(path, samples, W, H, fps, _visual_entry(b"avc1", W, H, b"alg_amd h264 ipcm", avcc))


class _BitReader:
    def __init__(self, data):
        self.b, self.p = np.unpackbits(np.frombuffer(data, dtype=np.uint8)), 0

    def u(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | int(self.b[self.p])
            self.p += 1
        return v

    def ue(self):
        z = 0
        while self.b[self.p] == 0:
            z += 1
            self.p += 1
        return self.u(z + 1) - 1

    def se(self):
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)


def _unescape_rbsp(data):
    import re
    return re.sub(b"\x00\x00\x03", b"\x00\x00", data)


def read_mp4(path):
    """Decoder for the files `write_mp4` produces (an independent walk of the container and of the H.264 syntax, not a
    general player): box tree -> sample table -> per sample either the JPEG or the I_PCM slice -> RGB frames
    [T, H, W, 3] uint8 plus a dict with what was parsed (codec, fps, profile / level, picture size, YUV planes)."""
    data = open(path, "rb").read()

    def children(lo, hi):
        out, p = [], lo
        while p + 8 <= hi:
            n, tag = struct.unpack(">I", data[p:p + 4])[0], data[p + 4:p + 8]
            out.append((tag, p + 8, p + n))
            p += n
        return out

    def find(lo, hi, *tags):
        for tag in tags:
            hit = [c for c in children(lo, hi) if c[0] == tag]
            if not hit:
                raise ValueError("box %r not found" % tag)
            _, lo, hi = hit[0]
        return lo, hi

    top = children(0, len(data))
    if top[0][0] != b"ftyp":
        raise ValueError("not an ISO base media file")
    mlo, mhi = find(0, len(data), b"moov", b"trak", b"mdia")
    lo, hi = find(mlo, mhi, b"mdhd")
    scale = struct.unpack(">I", data[lo + 12:lo + 16])[0]
    slo, shi = find(mlo, mhi, b"minf", b"stbl")
    lo, hi = find(slo, shi, b"stts")
    _, count, delta = struct.unpack(">III", data[lo + 4:lo + 16])
    lo, hi = find(slo, shi, b"stsz")
    T = struct.unpack(">I", data[lo + 8:lo + 12])[0]
    sizes = struct.unpack(">%dI" % T, data[lo + 12:lo + 12 + 4 * T])
    lo, hi = find(slo, shi, b"stco")
    off = struct.unpack(">I", data[lo + 8:lo + 12])[0]
    lo, hi = find(slo, shi, b"stsd")
    entry = children(lo + 8, hi)[0]
    W, H = struct.unpack(">HH", data[entry[1] + 24:entry[1] + 28])
    info = dict(fps=scale / delta, frames=T, width=W, height=H, codec=entry[0].decode())
    samples = []
    for n in sizes:
        samples.append(data[off:off + n])
        off += n
    if entry[0] == b"mp4v":
        from PIL import Image
        return np.stack([np.asarray(Image.open(io.BytesIO(s_)).convert("RGB")) for s_ in samples]), info
    # ---- avc1: parameter sets from avcC, then every sample's slice ----
    alo, ahi = find(entry[1] + 78, entry[2], b"avcC")
    cfg = data[alo:ahi]
    info.update(profile=cfg[1], level=cfg[3], nal_length_size=(cfg[4] & 3) + 1)
    n_sps = struct.unpack(">H", cfg[6:8])[0]
    sps = _BitReader(_unescape_rbsp(cfg[9:8 + n_sps]))
    assert sps.u(8) == 66
    sps.u(8), sps.u(8), sps.ue()
    log2_fn = sps.ue() + 4
    assert sps.ue() == 2
    sps.ue(), sps.u(1)
    mbw, mbh = sps.ue() + 1, sps.ue() + 1
    assert sps.u(1) == 1
    sps.u(1)
    crop = [0, 0, 0, 0]
    if sps.u(1):
        crop = [sps.ue() for _ in range(4)]
    info["cropped"] = (mbw * 16 - 2 * (crop[0] + crop[1]), mbh * 16 - 2 * (crop[2] + crop[3]))
    ys, cbs, crs = [], [], []
    for s_ in samples:
        n = struct.unpack(">I", s_[:4])[0]
        nal = s_[4:4 + n]
        assert nal[0] == 0x65 and n + 4 == len(s_)
        rb = _unescape_rbsp(nal[1:])
        r = _BitReader(rb[:16])
        assert r.ue() == 0 and r.ue() == 7 and r.ue() == 0
        r.u(log2_fn), r.ue(), r.u(2), r.se()
        assert r.ue() == 1 and r.ue() == 25
        start = (r.p + 7) // 8
        n_mb = mbw * mbh
        body = np.frombuffer(rb, dtype=np.uint8, count=n_mb * 386 - 2, offset=start)
        assert rb[start + n_mb * 386 - 2] == 0x80 and len(rb) == start + n_mb * 386 - 1
        blk = np.concatenate([np.array([0x0D, 0], dtype=np.uint8), body]).reshape(n_mb, 386)
        assert (blk[:, 0] == 0x0D).all() and (blk[:, 1] == 0).all()
        ys.append(blk[:, 2:258].reshape(mbh, mbw, 16, 16).transpose(0, 2, 1, 3).reshape(mbh * 16, mbw * 16))
        cbs.append(blk[:, 258:322].reshape(mbh, mbw, 8, 8).transpose(0, 2, 1, 3).reshape(mbh * 8, mbw * 8))
        crs.append(blk[:, 322:386].reshape(mbh, mbw, 8, 8).transpose(0, 2, 1, 3).reshape(mbh * 8, mbw * 8))
    y, cb, cr = np.stack(ys), np.stack(cbs), np.stack(crs)
    cw, ch = info["cropped"]
    info["yuv"] = (y[:, :ch, :cw], cb[:, :ch // 2, :cw // 2], cr[:, :ch // 2, :cw // 2])
    return yuv420_to_rgb(*info["yuv"]), info


def write_video(path, frames, fps=8, codec="h264"):
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
        write_mp4(path, a, fps, codec=codec)
    else:
        raise ValueError("unknown output extension %r" % ext)
    return path
