"""RecordIO containers as the reference reads them (`mx.recordio.MXIndexedRecordIO`, `mx.recordio.unpack_img`,
dataset/iterator.py:395, :556-557).  MXNet / dmlc-core are third-party and not vendored; the published format:

  .rec   sequence of  [uint32 0xced7230a][uint32 lrec][payload][pad to 4 bytes],  lrec = cflag << 29 | length.
         cflag 0: whole record; 1 / 2 / 3: first / middle / last part of a record that contained the magic word in
         its payload (the writer cuts there and drops the 4 magic bytes; the reader re-inserts them between parts).
  .idx   text, one "key\\toffset" line per record (offset of the record's first magic word in the .rec).
  item   IRHeader = struct 'IfQQ' (flag, label, id, id2); flag > 0: `flag` float32 labels follow and replace `label`;
         the rest is the encoded image (JPEG / PNG).

Decoding uses Pillow (OpenCV is not available here); both sit on libjpeg, and `unpack_img` returns the image in
OpenCV's BGR channel order like the reference's cv2.imdecode."""
import io
import struct
from collections import namedtuple

import numpy as np

IRHeader = namedtuple("HEADER", ["flag", "label", "id", "id2"])
_IR_FORMAT = "<IfQQ"
_IR_SIZE = struct.calcsize(_IR_FORMAT)
_MAGIC = 0xced7230a
_MAGIC_BYTES = struct.pack("<I", _MAGIC)


class MXRecordIO(object):
    """sequential reader / writer of a .rec file"""

    def __init__(self, uri, flag):
        assert flag in ("r", "w")
        self.uri, self.flag = uri, flag
        self.f = open(uri, "rb" if flag == "r" else "wb")

    def close(self):
        if self.f is not None:
            self.f.close()
            self.f = None

    def __del__(self):
        self.close()

    def reset(self):
        self.f.seek(0)

    def tell(self):
        return self.f.tell()

    def write(self, buf):
        """one record; the payload is cut at every 4-byte-aligned occurrence of the magic word (dmlc recordio.cc)"""
        assert self.flag == "w"
        buf = bytes(buf)
        cuts = [i for i in range(0, len(buf) - 3, 4) if buf[i:i + 4] == _MAGIC_BYTES]
        parts, start = [], 0
        for c in cuts:
            parts.append(buf[start:c])
            start = c + 4
        parts.append(buf[start:])
        for k, p in enumerate(parts):
            if len(parts) == 1:
                cflag = 0
            else:
                cflag = 1 if k == 0 else (3 if k == len(parts) - 1 else 2)
            assert len(p) < (1 << 29)
            self.f.write(struct.pack("<II", _MAGIC, (cflag << 29) | len(p)))
            self.f.write(p)
            self.f.write(b"\0" * ((4 - len(p) % 4) % 4))

    def read(self):
        """next record as bytes, None at the end of the file"""
        assert self.flag == "r"
        out = []
        while True:
            head = self.f.read(8)
            if len(head) < 8:
                if out:
                    raise IOError("%s: file ends inside a multi-part record" % self.uri)
                return None
            magic, lrec = struct.unpack("<II", head)
            if magic != _MAGIC:
                raise IOError("%s: bad record magic 0x%08x at offset %d" % (self.uri, magic, self.f.tell() - 8))
            cflag, length = lrec >> 29, lrec & ((1 << 29) - 1)
            payload = self.f.read(length)
            if len(payload) < length:
                raise IOError("%s: truncated record" % self.uri)
            self.f.seek((4 - length % 4) % 4, 1)
            out.append(payload)
            if cflag in (0, 3):
                return _MAGIC_BYTES.join(out)


class MXIndexedRecordIO(MXRecordIO):
    """random access through the .idx table (mx.recordio.MXIndexedRecordIO(idx_path, uri, flag))"""

    def __init__(self, idx_path, uri, flag, key_type=int):
        super(MXIndexedRecordIO, self).__init__(uri, flag)
        self.idx_path, self.key_type = idx_path, key_type
        self.idx, self.keys = {}, []
        if flag == "r":
            with open(idx_path, "r") as fin:
                for line in fin:
                    parts = line.strip().split("\t")
                    if len(parts) < 2:
                        continue
                    key = key_type(parts[0])
                    self.idx[key] = int(parts[1])
                    self.keys.append(key)
        else:
            self.fidx = open(idx_path, "w")

    def close(self):
        if getattr(self, "fidx", None) is not None:
            self.fidx.close()
            self.fidx = None
        super(MXIndexedRecordIO, self).close()

    def read_idx(self, idx):
        self.f.seek(self.idx[idx])
        return self.read()

    def write_idx(self, idx, buf):
        key = self.key_type(idx)
        pos = self.tell()
        self.write(buf)
        self.fidx.write("%s\t%d\n" % (str(key), pos))
        self.idx[key] = pos
        self.keys.append(key)


def pack(header, s):
    """mx.recordio.pack: header with a scalar label, or an array label (flag = its length)"""
    header = IRHeader(*header)
    label = header.label
    if isinstance(label, (int, float)) or np.ndim(label) == 0:
        header = header._replace(flag=0, label=float(label))
        tail = b""
    else:
        label = np.asarray(label, dtype=np.float32)
        header = header._replace(flag=label.size, label=0.0)
        tail = label.tobytes()
    return struct.pack(_IR_FORMAT, *header) + tail + bytes(s)


def unpack(s):
    """mx.recordio.unpack -> (IRHeader, payload bytes)"""
    header = IRHeader(*struct.unpack(_IR_FORMAT, s[:_IR_SIZE]))
    s = s[_IR_SIZE:]
    if header.flag > 0:
        header = header._replace(label=np.frombuffer(s, np.float32, header.flag).copy())
        s = s[header.flag * 4:]
    return header, s


def imdecode(buf, grey=False):
    """encoded bytes -> uint8 array in OpenCV's convention: (h, w, 3) BGR, or (h, w) for single-channel images read
    'unchanged' (cv2.imread(path, -1) of a label PNG)"""
    from PIL import Image
    im = Image.open(io.BytesIO(buf))
    if grey or im.mode in ("L", "P", "I;16", "I"):
        if im.mode == "P":
            return np.asarray(im).copy()                 # palette indices ARE the label ids
        return np.asarray(im.convert("L") if grey else im).copy()
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


def unpack_img(s):
    """mx.recordio.unpack_img -> (IRHeader, BGR uint8 image)"""
    header, s = unpack(s)
    return header, imdecode(s)


def pack_img(header, img_bgr, quality=95, img_fmt=".jpg"):
    """mx.recordio.pack_img for a BGR uint8 image"""
    from PIL import Image
    buf = io.BytesIO()
    im = Image.fromarray(np.ascontiguousarray(img_bgr[:, :, ::-1]))
    if img_fmt.lower() in (".jpg", ".jpeg"):
        im.save(buf, format="JPEG", quality=quality)
    else:
        im.save(buf, format="PNG")
    return pack(header, buf.getvalue())
