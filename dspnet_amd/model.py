"""Checkpoint files of the reference: `prefix-%04d.params` as written by MXNet's `mx.nd.save` /
`mx.callback.do_checkpoint` and read by `mx.model.load_checkpoint(prefix, epoch)` (multi_train.py:338,350,370;
detect/multitask_detector.py:105; multi_eval.py:216).

The byte format is MXNet's (third-party, not vendored in the reference, version unpinned -- SURVEY.md section 8c);
it is restated here from MXNet's published `NDArray::Save/Load` (src/ndarray/ndarray.cc) and `MXNDArraySave`
(src/c_api/c_api.cc).  All integers little-endian:

    uint64  0x112                      list magic (kMXAPINDArrayListMagic)
    uint64  0                          reserved
    uint64  n                          number of arrays
    n x NDArray:
        uint32  magic                  0xF993FAC9 (V2, MXNet >= 0.12) | 0xF993FACA (V3, numpy shapes) |
                                       0xF993FAC8 (V1) | anything else = LEGACY: the word is ndim itself
        int32   storage type           V2 / V3 only; 0 = dense (sparse arrays are rejected here)
        uint32  ndim                   (absent in LEGACY: the magic word was ndim)
        ndim x int64 dims              (LEGACY: ndim x uint32)
        int32   dev_type, int32 dev_id context the array was saved from (ignored on load; written as cpu(0))
        int32   type flag              0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64
        raw data, C order
    uint64  m                          number of names (0 or n)
    m x (uint64 length, bytes)         "arg:<name>" / "aux:<name>" in checkpoints

A zero-size shape (ndim 0 in V1/V2) marks a "none" array that carries no context / type / data.
PARITY STATUS: unpinned -- no MXNet here to produce or read a file; tests/test_params_io.py holds a byte-for-byte
hand-assembled file per format generation."""
import struct

import numpy as np

LIST_MAGIC = 0x112
V1_MAGIC, V2_MAGIC, V3_MAGIC = 0xF993FAC8, 0xF993FAC9, 0xF993FACA
_TYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _TYPES.items()}


class ParamsFormatError(ValueError):
    pass


class _Reader:
    def __init__(self, buf):
        self.buf, self.pos = memoryview(buf), 0

    def take(self, n):
        if self.pos + n > len(self.buf):
            raise ParamsFormatError("truncated file: need %d bytes at offset %d of %d" % (n, self.pos, len(self.buf)))
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out

    def unpack(self, fmt):
        return struct.unpack("<" + fmt, self.take(struct.calcsize("<" + fmt)))


def _read_ndarray(r):
    (magic,) = r.unpack("I")
    if magic in (V2_MAGIC, V3_MAGIC):
        (stype,) = r.unpack("i")
        if stype != 0:
            raise ParamsFormatError("sparse storage type %d is not supported" % stype)
    if magic in (V1_MAGIC, V2_MAGIC, V3_MAGIC):
        (ndim,) = r.unpack("I") if magic != V3_MAGIC else r.unpack("i")
        if magic == V3_MAGIC and ndim < 0:
            return None                                     # unknown shape: a "none" array
        shape = r.unpack("%dq" % ndim) if ndim else ()
        if magic != V3_MAGIC and ndim == 0:
            return None
    else:                                                   # legacy: the word just read is ndim, dims are uint32
        ndim = magic
        if ndim > 32:
            raise ParamsFormatError("not an NDArray record (first word 0x%08x)" % magic)
        shape = r.unpack("%dI" % ndim) if ndim else ()
        if ndim == 0:
            return None
    r.unpack("ii")                                          # context
    (flag,) = r.unpack("i")
    if flag not in _TYPES:
        raise ParamsFormatError("unknown type flag %d" % flag)
    dt = np.dtype(_TYPES[flag]).newbyteorder("<")
    count = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    data = np.frombuffer(r.take(count * dt.itemsize), dtype=dt, count=count)
    return data.reshape(shape).astype(_TYPES[flag], copy=True)


def nd_load(fname):
    """mx.nd.load: -> dict name -> numpy array when the file carries names, else a list"""
    with open(fname, "rb") as f:
        r = _Reader(f.read())
    magic, _reserved = r.unpack("QQ")
    if magic != LIST_MAGIC:
        raise ParamsFormatError("%s: not an NDArray list file (header 0x%x)" % (fname, magic))
    (n,) = r.unpack("Q")
    arrays = [_read_ndarray(r) for _ in range(n)]
    (m,) = r.unpack("Q")
    if m not in (0, n):
        raise ParamsFormatError("%s: %d names for %d arrays" % (fname, m, n))
    names = []
    for _ in range(m):
        (ln,) = r.unpack("Q")
        names.append(bytes(r.take(ln)).decode("utf-8"))
    if r.pos != len(r.buf):
        raise ParamsFormatError("%s: %d trailing bytes" % (fname, len(r.buf) - r.pos))
    return dict(zip(names, arrays)) if m else arrays


def nd_save(fname, data):
    """mx.nd.save in the V2 layout: data = dict name -> array, or a list of arrays"""
    names = list(data.keys()) if isinstance(data, dict) else []
    arrays = list(data.values()) if isinstance(data, dict) else list(data)
    out = [struct.pack("<QQQ", LIST_MAGIC, 0, len(arrays))]
    for a in arrays:
        a = np.ascontiguousarray(a)
        if a.dtype not in _FLAGS:
            raise ParamsFormatError("dtype %s has no MXNet type flag" % a.dtype)
        if a.ndim == 0:
            a = a.reshape(1)
        out.append(struct.pack("<IiI", V2_MAGIC, 0, a.ndim))
        out.append(struct.pack("<%dq" % a.ndim, *a.shape))
        out.append(struct.pack("<iii", 1, 0, _FLAGS[a.dtype]))              # cpu(0)
        out.append(a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes())
    out.append(struct.pack("<Q", len(names)))
    for k in names:
        b = k.encode("utf-8")
        out.append(struct.pack("<Q", len(b)) + b)
    with open(fname, "wb") as f:
        f.write(b"".join(out))


def load_params(prefix, epoch):
    """mx.model.load_params: 'prefix-%04d.params' -> (arg_params, aux_params) split on the 'arg:' / 'aux:' tags"""
    saved = nd_load("%s-%04d.params" % (prefix, epoch))
    if not isinstance(saved, dict):
        raise ParamsFormatError("checkpoint without names")
    arg_params, aux_params = {}, {}
    for k, v in saved.items():
        tag, _, name = k.partition(":")
        if tag == "arg":
            arg_params[name] = v
        elif tag == "aux":
            aux_params[name] = v
    return arg_params, aux_params


def load_checkpoint(prefix, epoch):
    """mx.model.load_checkpoint -> (symbol, arg_params, aux_params).  The symbol json is MXNet's graph description;
    this build constructs its graph from symbol/multitask_symbol_factory.py, so the first item is always None."""
    arg_params, aux_params = load_params(prefix, epoch)
    return None, arg_params, aux_params


def save_checkpoint(prefix, epoch, net, aux_params=None):
    """mx.model.save_checkpoint for this build's graph (`net`: a MultiTaskNet or its Graph): every parameter in the
    reference's shapes under 'arg:', plus what MXNet lists for the same layers and this graph does not own:
    `gamma` = 1 for fix_gamma BatchNorms and the moving statistics under 'aux:'.  The solver of the reference zeroes
    the aux states and always runs with batch statistics (multi_solver.py:212,284), so they never influence an
    output; they are written from `aux_params` when given (a loaded checkpoint carried through) else as 0 / 1."""
    g = getattr(net, "g", net)
    blob = {}
    args = g.get_params()
    for name, channels, fix_gamma in g.bn_names:
        if fix_gamma:
            args.setdefault(name + "_gamma", np.ones(channels, np.float32))
    for k, v in args.items():
        blob["arg:" + k] = v
    aux_params = aux_params or {}
    for name, channels, _ in g.bn_names:
        blob["aux:%s_moving_mean" % name] = np.asarray(aux_params.get(name + "_moving_mean", np.zeros(channels)), np.float32)
        blob["aux:%s_moving_var" % name] = np.asarray(aux_params.get(name + "_moving_var", np.ones(channels)), np.float32)
    nd_save("%s-%04d.params" % (prefix, epoch), blob)
