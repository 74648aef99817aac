"""MultiTaskRecordIter (dataset/iterator.py:301-603): batches of (data, label_det, seg_out_label) from an indexed
RecordIO file plus per-image label-map PNGs, with the reference's augmentation (flip, rotation, anisotropic scale,
translation) or its plain resize.

Division of labour (MI355X-first):
  host   record lookup, JPEG / PNG decode (Pillow, a small thread pool: the decoders release the GIL), the box
         arithmetic of `_get_augmented` / `_get_resized` (a few dozen boxes per image), the random draws;
  device ONE upload of the decoded uint8 pixels of the whole batch and ONE launch pair (dspn_augment_batch_u8,
         include/dspn_augment.h) that does warpAffine (bilinear, OpenCV's fixed point) + flip + BGR->RGB planes + mean
         subtraction for the images and nearest warp + flip + /4 resize + LUT for the label maps.  The reference runs
         four OpenCV passes per sample on the host and uploads float32 planes.

The shuffle and the augmentation parameters come from the same MT19937 stream as the reference
(`np.random.seed(233)`, then shuffle / rand in the reference's call order, :381-383, :406-419), so for a given
record file the sample order and every (flip, theta, sx, sy, tx, ty) are identical.

Kept from the reference on purpose: an image without any box returns before the flip (:498-499) -- it is never
flipped even when its draw says so; exactly one surviving box is written to the first SIX label rows (np.squeeze
turns the (1, 6) selection into (6,), :542-544); `iter_next` drops the last partial batch (:421-423)."""
import ctypes as _c
import math
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import _lib
from . import recordio

# dspn_warp_sample (include/dspn_augment.h)
_SAMPLE = np.dtype([("img_offset", "<i8"), ("seg_offset", "<i8"), ("src_h", "<i4"), ("src_w", "<i4"), ("flip", "<i4"),
                    ("img_border", "<i4"), ("seg_border", "<i4"), ("reserved", "<i4"), ("minv", "<f8", (6,))])
assert _SAMPLE.itemsize == 88
_CMAP_BGR = (_c.c_int * 3)(2, 1, 0)                             # BGR source -> R, G, B planes (:568-569)
_CMAP_RGB = (_c.c_int * 3)(0, 1, 2)                             # Pillow decodes to RGB: identity


_lib.register({"dspn_augment_batch_u8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                                     _c.POINTER(_c.c_int), _c.POINTER(_c.c_double), _c.c_void_p,
                                                     _c.c_void_p, _c.c_void_p, _c.c_void_p])})


def _entry():
    """dspn_augment_batch_u8 (fails loudly when the HIP library is missing)"""
    return _lib.lib().dspn_augment_batch_u8


def seg_lut():
    """:361-366 over dataset/cs_labels.py: ids whose trainId is >= 0 (0..34) map to themselves, the rest to 255"""
    lut = np.full(256, 255, np.uint8)
    lut[:35] = np.arange(35, dtype=np.uint8)
    return lut


def invert_affine(M):
    """the inverse map cv2.warpAffine derives from M (double arithmetic, same statement order)"""
    m = np.array(M, np.float64).reshape(-1).copy()
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11; m[1] *= -D; m[3] *= -D; m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2] = b1; m[5] = b2
    return m


def _pack_to_top(rows):
    """surviving rows (xmax > -0.5) first, -1 below; a single survivor fills six rows (module note)"""
    keep = rows[rows[:, 3] > -.5]
    rows[:] = -1
    if keep.shape[0] == 1:
        rows[:6] = keep[0]
    else:
        rows[:keep.shape[0]] = keep


def resized_boxes(rows, data_shape):
    """box bookkeeping of `_get_resized` (:447-469) on the (n, 6) label rows, in place"""
    if not (rows[:, 0] >= 0).any():
        return
    area = (rows[:, 3] - rows[:, 1]) * data_shape[2] * (rows[:, 4] - rows[:, 2]) * data_shape[1]
    rows[area < 100] = -1
    _pack_to_top(rows)


def augmented_boxes(rows, data_shape, aug):
    """box bookkeeping of `_get_augmented` (:486-544) on the (n, 6) label rows, in place.
    -> False when the sample has no box (the caller then skips the flip, like the reference's early return)"""
    flip, theta, sx, sy, tx, ty = (float(a) for a in aug)
    W, H = data_shape[2], data_shape[1]
    sel = np.where(rows[:, 0] >= 0)[0]
    if sel.size < 1:
        return False
    dist = rows[sel, 5].copy()
    box = rows[sel, 1:5] * np.array([W, H, W, H], np.float64)
    A = np.array([[sx * math.cos(theta), -sy * math.sin(theta)], [sx * math.sin(theta), sy * math.cos(theta)]])
    t = np.array([tx, ty])
    p0 = box[:, :2] @ A.T + t                                   # cv2.transform of the (xmin, ymin) corners
    p1 = box[:, 2:] @ A.T + t                                   # ... and of the (xmax, ymax) corners
    if flip > .5:
        p0[:, 0] = W - p0[:, 0]
        p1[:, 0] = W - p1[:, 0]
    scale = np.array([1.0 / W, 1.0 / H])
    p0 = p0 * scale
    p1 = p1 * scale
    new = np.hstack((p0, p1))
    if flip > .5:
        new[:, [0, 2]] = new[:, [2, 0]]
    vis = new[:, 2] > -.5
    new[vis] = np.clip(new[vis], 0, 1)
    rows[sel, 1:5] = new
    rows[sel, 5] = dist / math.sqrt(sx * sy)
    area = (rows[:, 3] - rows[:, 1]) * W * (rows[:, 4] - rows[:, 2]) * H
    rows[area < 100] = -1
    rows[rows[:, 3] < .01] = -1
    rows[rows[:, 1] > .99] = -1
    rows[rows[:, 4] < .01] = -1
    rows[rows[:, 2] > .99] = -1
    _pack_to_top(rows)
    return True


class _Decoded(object):
    """decoded pixels of a cached record, with the two attributes _prepare reads from a PIL image"""

    def __init__(self, a):
        self.a = a
        self.size = (a.shape[1], a.shape[0])


class DataBatch(object):
    def __init__(self, data, label):
        self.data, self.label = data, label


class MultiTaskRecordIter(object):
    """see the module docstring; constructor arguments as in the reference (unused ones accepted and ignored)"""

    def __init__(self, path_imgrec, batch_size, data_shape, path_imglist="", label_width=-1, label_pad_width=-1,
                 label_pad_value=-1, resize_mode="force", mean_pixels=(123.68, 116.779, 103.939), enable_aug=True,
                 device=None, decode_threads=8, prefetch=True, cache_decoded=False, **kwargs):
        path_imgidx = path_imgrec.replace(".rec", ".idx")
        path_imglst = path_imgrec.replace(".rec", ".lst")
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.enable_aug = enable_aug
        self.batch_size = batch_size
        self.data_shape = tuple(data_shape)
        assert self.data_shape[0] == 3 and self.data_shape[1] % 4 == 0 and self.data_shape[2] % 4 == 0
        self.mean_pixels = list(mean_pixels)
        self.angle_range = (-5, 5)
        self.scale_range = (.5, 2.)
        self.ratio_range = (.8, 1.2)
        self.lut = seg_lut()
        self._lut_dev = torch.from_numpy(self.lut).to(self.device)
        with open(path_imgidx, "r") as f:
            self.num_samples = sum(1 for _ in f)
        self.index_table = np.arange(self.num_samples)
        self._rs = np.random.RandomState(233)                  # np.random.seed(233) (:381)
        self._rs.shuffle(self.index_table)
        self._reset_aug_params()
        self.curr_index = 0
        self.imglst = {}
        dirname = os.path.dirname(path_imglst)
        with open(path_imglst, "r") as fp:
            for line in fp:
                patch = line.rstrip("\n").split()
                if not patch:
                    continue
                segfile = patch[-1].replace("leftImg8bit.jpg", "gtFine_labelTrainIds.png")
                segfile = segfile.replace("JPEGImages", "SegmentationClass")
                self.imglst[patch[0]] = os.path.join(dirname, "cityscapes", segfile)
        self.rec = recordio.MXIndexedRecordIO(path_imgidx, path_imgrec, "r")
        self.provide_data = [("data", [self.batch_size] + list(self.data_shape))]
        self.provide_label = None
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(decode_threads)))
        self._producer = ThreadPoolExecutor(max_workers=1)
        self.prefetch, self._ahead = bool(prefetch), None
        # cache_decoded: keep the decoded uint8 pixels of every record in host memory (Cityscapes train: 2975 x 8.4 MB =
        # 25 GB); from the second epoch on the host phase is a memcpy instead of a JPEG + PNG decode
        self._cache = {} if cache_decoded else None
        self._get_batch()
        if not self.provide_label:
            raise RuntimeError("Invalid ImageDetRecordIter: " + path_imgrec)
        self.reset()

    # ---- epoch control (:401-434) -------------------------------------------------------------------------------
    def reset(self):
        self._drop_ahead()                                     # a batch prepared ahead belongs to the old order
        self._rs.shuffle(self.index_table)
        self.curr_index = 0
        self._reset_aug_params()

    def _reset_aug_params(self):
        n, r = self.num_samples, self._rs
        p = np.zeros((n, 6))                                   # flip, theta, sx, sy, tx, ty
        p[:, 0] = r.rand(n) > .5
        p[:, 1] = np.radians(self.angle_range[0] + r.rand(n) * (self.angle_range[1] - self.angle_range[0]))
        p[:, 2] = self.scale_range[0] + r.rand(n) * (self.scale_range[1] - self.scale_range[0])
        p[:, 3] = p[:, 2] * (self.ratio_range[0] + r.rand(n) * (self.ratio_range[1] - self.ratio_range[0]))
        p[:, 4] = -r.rand(n) * self.data_shape[2] * (p[:, 2] - 1.)
        p[:, 5] = -r.rand(n) * self.data_shape[1] * (p[:, 3] - 1.)
        self.aug_params = p

    def iter_next(self):
        return (self.curr_index + self.batch_size) <= self.num_samples

    def next(self):
        if self.iter_next():
            self._get_batch()
            return self._batch, self._fnames
        raise StopIteration

    __next__ = next

    def __iter__(self):
        return self

    # ---- one batch (:550-603) -----------------------------------------------------------------------------------
    # host phase (_prepare: records, decode straight into a pinned pool, boxes, descriptors; may run ahead of the
    # consumer on the producer thread) and device phase (_finish: one upload, one launch pair, on the caller's stream)
    def _peek(self, item):
        """header, label row and the lazily opened image(s) of one record: sizes are known before any pixel is decoded.
        With cache_decoded the images come back as arrays from the second time a record is seen."""
        import io
        from PIL import Image
        header, payload = recordio.unpack(item)
        hdr = np.array([header.label.shape[0]] + header.label.tolist())
        seg_path = self.imglst[str(header.id)]
        hit = self._cache.get(header.id) if self._cache is not None else None
        if hit is not None:
            return hdr, _Decoded(hit[0]), (None if hit[1] is None else _Decoded(hit[1])), seg_path, header.id
        im = Image.open(io.BytesIO(payload))
        seg_im = Image.open(seg_path) if os.path.exists(seg_path) else None
        if self.enable_aug:
            assert seg_im is not None, seg_path + " not found."
        if seg_im is not None:
            assert seg_im.size == im.size, "label map and image differ in size: " + seg_path
        return hdr, im, seg_im, seg_path, header.id

    def _decode_into(self, args):
        """decode one image / label map into its slice of the pinned pools (worker thread; Pillow's decoders and
        numpy's copy loops release the GIL)"""
        im, seg_im, img_dst, seg_dst, rec_id = args
        if isinstance(im, _Decoded):                           # cached pixels: one memcpy each
            img_dst[...] = im.a
            if seg_im is not None:
                seg_dst[...] = seg_im.a
            return True
        img_dst[...] = np.asarray(im if im.mode == "RGB" else im.convert("RGB"))   # the kernel's channel map is the identity
        if seg_im is not None:
            a = np.asarray(seg_im if seg_im.mode in ("L", "P") else seg_im.convert("L"))
            seg_dst[...] = a
        if self._cache is not None:
            self._cache[rec_id] = (img_dst.copy(), None if seg_im is None else seg_dst.copy())
        return True

    def _pool_set(self, k, img_bytes, seg_bytes):
        """pinned staging buffers, two sets used alternately, grown on demand; a set is rewritten only after the
        upload that last read it has completed"""
        if not hasattr(self, "_pools"):
            self._pools = [None, None]
        cur = self._pools[k]
        if cur is not None and cur[2] is not None:
            cur[2].synchronize()
        if cur is None or cur[0].numel() < img_bytes or cur[1].numel() < seg_bytes:
            cur = [torch.empty(max(img_bytes, 1), dtype=torch.uint8).pin_memory(),
                   torch.empty(max(seg_bytes, 1), dtype=torch.uint8).pin_memory(), None]
            self._pools[k] = cur
        return cur

    def _prepare(self, start, serial):
        B, (C, H, W) = self.batch_size, self.data_shape
        items = [self.rec.read_idx(self.rec.key_type(self.index_table[s])) for s in range(start, start + B)]
        peeked = [self._peek(item) for item in items]
        img_sizes = [im.size[0] * im.size[1] * 3 for _, im, _, _, _ in peeked]
        seg_sizes = [(sm.size[0] * sm.size[1] if sm is not None else 0) for _, _, sm, _, _ in peeked]
        pool = self._pool_set(serial % 2, sum(img_sizes), sum(seg_sizes))
        ip, sp = pool[0].numpy(), pool[1].numpy()
        label = np.ones((B, 1206)) * -1
        label[:, :3] = np.array(list(self.data_shape))
        samples = np.zeros(B, _SAMPLE)
        jobs, fnames = [], []
        io_ = so = 0
        for b, (hdr, im, seg_im, seg_path, rec_id) in enumerate(peeked):
            ww, hh = im.size
            slot = start + b
            rows = hdr[3:].reshape((-1, 6))                     # view: the edits land in hdr
            if self.enable_aug:
                flip, theta, sx, sy, tx, ty = self.aug_params[slot]
                sx2, sy2 = sx * (W / float(ww)), sy * (H / float(hh))
                M = [sx2 * math.cos(theta), -sy2 * math.sin(theta), tx, sx2 * math.sin(theta), sy2 * math.cos(theta), ty]
                has_box = augmented_boxes(rows, self.data_shape, self.aug_params[slot])
                flip_img, ib, sb = bool(flip > .5) and has_box, 128, 255
            else:
                M = [1. * (W / float(ww)), -0.0, 0, 0.0, 1. * (H / float(hh)), 0]
                resized_boxes(rows, self.data_shape)
                flip_img, ib, sb = False, 0, 0
            s = samples[b]
            s["img_offset"], s["src_h"], s["src_w"] = io_, hh, ww
            s["seg_offset"] = so if seg_im is not None else -1
            s["flip"], s["img_border"], s["seg_border"] = int(flip_img), ib, sb
            s["minv"] = invert_affine(M)
            jobs.append((im, seg_im, ip[io_:io_ + img_sizes[b]].reshape(hh, ww, 3),
                         sp[so:so + seg_sizes[b]].reshape(hh, ww) if seg_im is not None else None, rec_id))
            io_ += img_sizes[b]
            so += seg_sizes[b]
            label[b, 3:3 + hdr.shape[0]] = hdr
            fnames.append(seg_path)
        list(self._pool.map(self._decode_into, jobs))
        return {"pool": pool, "img_bytes": io_, "seg_bytes": so, "samples": samples, "label": label, "fnames": fnames,
                "start": start}

    def _finish(self, prep):
        B, (C, H, W) = self.batch_size, self.data_shape
        dev = self.device
        pool = prep["pool"]
        img_dev = pool[0][:max(prep["img_bytes"], 1)].to(dev, non_blocking=True)
        seg_dev = pool[1][:max(prep["seg_bytes"], 1)].to(dev, non_blocking=True)
        pool[2] = torch.cuda.Event()
        pool[2].record(torch.cuda.current_stream(dev))
        desc_dev = torch.from_numpy(prep["samples"].view(np.uint8).reshape(-1)).to(dev)
        data = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
        seg_out = torch.empty(B, H // 4, W // 4, dtype=torch.float32, device=dev)
        _lib.check(_entry()(img_dev.data_ptr(), seg_dev.data_ptr(), desc_dev.data_ptr(), B, H, W, _CMAP_RGB,
                            (_c.c_double * 3)(*[float(m) for m in self.mean_pixels]), self._lut_dev.data_ptr(),
                            data.data_ptr(), seg_out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                   "augment_batch")
        label = prep["label"]
        if self.provide_label is None:
            first = label[0]
            self.label_header_width = int(first[4])
            self.label_object_width = int(first[5])
            assert self.label_object_width >= 5, "object width must >=5"
            self.label_start = 4 + self.label_header_width
            self.max_objects = (first.size - self.label_start) // self.label_object_width
            self.label_shape = (B, self.max_objects, self.label_object_width)
            self.label_end = self.label_start + self.max_objects * self.label_object_width
            self.provide_label = [("label_det", self.label_shape), ("seg_out_label", tuple(seg_out.shape))]
        det = label[:, self.label_start:self.label_end].reshape((B, self.max_objects, self.label_object_width))
        det_dev = torch.from_numpy(det.astype(np.float32)).to(dev)
        self._batch = DataBatch(data=[data], label=[det_dev, seg_out])
        self._fnames = prep["fnames"]
        self._keepalive = (img_dev, seg_dev, desc_dev)          # until the next batch replaces them

    def _drop_ahead(self):
        fut, self._ahead = getattr(self, "_ahead", None), None
        if fut is not None:
            fut.result()                                        # let it finish: it writes into a pinned pool

    def _get_batch(self):
        fut, self._ahead = getattr(self, "_ahead", None), None
        prep = fut.result() if fut is not None else None
        if prep is None or prep["start"] != self.curr_index:    # nothing prepared ahead (or the order changed)
            self._serial = getattr(self, "_serial", 0) + 1
            prep = self._prepare(self.curr_index, self._serial)
        self.curr_index += self.batch_size
        self._finish(prep)
        if self.prefetch and self.iter_next():                  # the next batch's host phase overlaps the consumer
            self._serial += 1
            self._ahead = self._producer.submit(self._prepare, self.curr_index, self._serial)
