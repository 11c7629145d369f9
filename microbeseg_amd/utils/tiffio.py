"""Minimal TIFF reader/writer for the on-disk formats of the hot path: uncompressed
grayscale uint8 / uint16 / int32 / float32 images, single page (H, W) or multi-page stacks (T, H, W).
``tifffile`` (what the reference uses: training_dataset.py:40, infer_script_local.py:82,165) is used when present;
the build / GPU image does not ship it, hence this dependency-free fallback for exactly those formats."""
import struct

import numpy as np

try:  # pragma: no cover
    import tifffile as _tf
except Exception:
    _tf = None

_TYPES = {1: ("B", 1), 2: ("c", 1), 3: ("H", 2), 4: ("I", 4), 5: ("II", 8), 16: ("Q", 8)}
_DTYPES = {(1, 8): np.uint8, (1, 16): np.uint16, (1, 32): np.uint32, (2, 8): np.int8, (2, 16): np.int16,
           (2, 32): np.int32, (3, 32): np.float32, (3, 64): np.float64}


def _read_ifd(buf, off, bo):
    n, = struct.unpack_from(bo + "H", buf, off)
    tags = {}
    for i in range(n):
        tag, typ, cnt = struct.unpack_from(bo + "HHI", buf, off + 2 + 12 * i)
        fmt, size = _TYPES.get(typ, ("B", 1))
        voff = off + 2 + 12 * i + 8
        if size * cnt > 4:
            voff, = struct.unpack_from(bo + "I", buf, voff)
        if typ == 5:
            vals = struct.unpack_from(bo + "II" * cnt, buf, voff)
        elif typ == 2:
            vals = (buf[voff:voff + cnt],)
        else:
            vals = struct.unpack_from(bo + fmt * cnt, buf, voff)
        tags[tag] = vals
    nxt, = struct.unpack_from(bo + "I", buf, off + 2 + 12 * n)
    return tags, nxt


def imread(path):
    if _tf is not None:  # pragma: no cover
        return _tf.imread(str(path))
    with open(path, "rb") as f:
        buf = f.read()
    bo = {b"II": "<", b"MM": ">"}[buf[:2]]
    magic, off = struct.unpack_from(bo + "HI", buf, 2)
    if magic != 42:
        raise ValueError(f"{path}: not a classic TIFF (BigTIFF/compressed files need tifffile)")
    pages = []
    while off:
        t, off = _read_ifd(buf, off, bo)
        w, h = t[256][0], t[257][0]
        bits = t.get(258, (1,))[0]
        comp = t.get(259, (1,))[0]
        spp = t.get(277, (1,))[0]
        fmt = t.get(339, (1,))[0]
        if comp != 1:
            raise ValueError(f"{path}: compressed TIFF (compression {comp}) needs tifffile")
        dt = np.dtype(_DTYPES[(fmt, bits)]).newbyteorder(bo)
        offs, cnts = t[273], t[279]
        data = b"".join(buf[o:o + c] for o, c in zip(offs, cnts))
        arr = np.frombuffer(data, dtype=dt, count=h * w * spp).astype(dt.newbyteorder("="))
        pages.append(arr.reshape((h, w, spp)) if spp > 1 else arr.reshape((h, w)))
    return pages[0] if len(pages) == 1 else np.stack(pages)


def imwrite(path, arr):
    arr = np.asarray(arr)
    if _tf is not None:  # pragma: no cover
        _tf.imwrite(str(path), arr)
        return
    if arr.ndim == 2:
        pages = [arr]
    elif arr.ndim == 3:
        pages = list(arr)
    else:
        raise ValueError("only (H, W) or (T, H, W) arrays")
    kind = {"u": 1, "i": 2, "f": 3}[arr.dtype.kind]
    bits = arr.dtype.itemsize * 8
    out = bytearray(b"II" + struct.pack("<HI", 42, 0))
    prev_next_ptr = 4
    for p in pages:
        p = np.ascontiguousarray(p, dtype=arr.dtype.newbyteorder("<"))
        if len(out) % 2:
            out += b"\0"
        data_off = len(out)
        out += p.tobytes()
        if len(out) % 2:
            out += b"\0"
        ifd_off = len(out)
        struct.pack_into("<I", out, prev_next_ptr, ifd_off)
        h, w = p.shape
        entries = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1),
                   (273, 4, 1, data_off), (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, p.nbytes), (339, 3, 1, kind)]
        out += struct.pack("<H", len(entries))
        for tag, typ, cnt, val in entries:
            out += struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
        prev_next_ptr = len(out)
        out += struct.pack("<I", 0)
    with open(path, "wb") as f:
        f.write(bytes(out))
