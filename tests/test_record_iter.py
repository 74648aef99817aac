"""Data path: RecordIO container, the box bookkeeping of MultiTaskRecordIter, and the fused
device pipeline (warpAffine + flip + planes + mean | nearest warp + /4 + LUT) against the numpy restatement."""
import copy
import os
import struct

import numpy as np
import pytest

from dspnet_amd.dataset import iterator as it
from dspnet_amd.dataset import recordio
from oracle import augment as oa

MAGIC = struct.pack("<I", 0xced7230a)


# ---------------------------------------------------------------------------------------------- RecordIO container
def test_recordio_layout_and_magic_split(tmp_path):
    rec, idx = str(tmp_path / "a.rec"), str(tmp_path / "a.idx")
    payloads = [b"abcdefg", b"", b"1234" + MAGIC + b"5678" + MAGIC + b"9", MAGIC, b"xy" + MAGIC + b"zz"]
    w = recordio.MXIndexedRecordIO(idx, rec, "w")
    for i, p in enumerate(payloads):
        w.write_idx(i, p)
    w.close()
    raw = open(rec, "rb").read()
    # first record: magic, lrec = (0 << 29) | 7, payload, one pad byte
    assert raw[:16] == MAGIC + struct.pack("<I", 7) + b"abcdefg\0"
    # third record is cut at the two aligned magic words: parts "1234" (cflag 1), "5678" (2), "9" (3)
    off = int(open(idx).read().split("\n")[2].split("\t")[1])
    assert raw[off:off + 12] == MAGIC + struct.pack("<I", (1 << 29) | 4) + b"1234"
    assert raw[off + 12:off + 24] == MAGIC + struct.pack("<I", (2 << 29) | 4) + b"5678"
    assert raw[off + 24:off + 36] == MAGIC + struct.pack("<I", (3 << 29) | 1) + b"9\0\0\0"
    r = recordio.MXIndexedRecordIO(idx, rec, "r")
    assert r.keys == [0, 1, 2, 3, 4]
    for i in (3, 0, 4, 2, 1):                               # random access
        assert r.read_idx(i) == payloads[i]
    r.reset()
    assert [r.read() for _ in range(5)] == payloads and r.read() is None     # (the unaligned magic in #4 is not cut)


def test_pack_unpack_header_and_image():
    label = np.array([2, 6, 1, .1, .2, .3, .4, .5], np.float32)
    s = recordio.pack(recordio.IRHeader(0, label, 7, 0), b"payload")
    assert s[:24] == struct.pack("<IfQQ", 8, 0.0, 7, 0)
    h, body = recordio.unpack(s)
    assert h.flag == 8 and h.id == 7 and body == b"payload"
    np.testing.assert_array_equal(h.label, label)
    h2, _ = recordio.unpack(recordio.pack(recordio.IRHeader(0, 3.0, 1, 0), b""))
    assert h2.flag == 0 and h2.label == 3.0
    img = np.zeros((8, 12, 3), np.uint8)
    img[..., 0] = 250                                        # blue in BGR
    h3, back = recordio.unpack_img(recordio.pack_img(recordio.IRHeader(0, label, 1, 0), img, img_fmt=".png"))
    np.testing.assert_array_equal(back, img)


# ---------------------------------------------------------------------------------------------- warp restatement
def _smooth_image(g, h, w, c=3):
    yy, xx = np.mgrid[:h, :w]
    chans = [127 + 100 * np.sin(xx / g.uniform(5, 15) + g.uniform(0, 3)) * np.cos(yy / g.uniform(5, 15)) for _ in range(c)]
    return np.clip(np.stack(chans, -1) + g.normal(0, 6, (h, w, c)), 0, 255).astype(np.uint8)


def test_warp_restatement_properties():
    g = np.random.Generator(np.random.PCG64(1))
    img = _smooth_image(g, 40, 56)
    ident = [[1, 0, 0], [0, 1, 0]]
    np.testing.assert_array_equal(oa.warp_affine(img, ident, (56, 40), True, 128), img)
    np.testing.assert_array_equal(oa.warp_affine(img[..., 0], ident, (56, 40), False, 255), img[..., 0])
    shifted = oa.warp_affine(img, [[1, 0, 5], [0, 1, -3]], (56, 40), True, 128)          # dst(x, y) = src(x - 5, y + 3)
    np.testing.assert_array_equal(shifted[:37, 5:], img[3:, :51])
    assert (shifted[:, :5] == 128).all() and (shifted[37:] == 128).all()
    # against a float bilinear interpolation: the 1/32-pixel coordinate grid and 15-bit weights stay within 2 grey levels
    M = np.array([[1.3 * np.cos(.07), -1.1 * np.sin(.07), -6.5], [1.3 * np.sin(.07), 1.1 * np.cos(.07), 2.25]])
    got = oa.warp_affine(img, M, (64, 48), True, 128).astype(np.float64)
    minv = oa.invert_affine(M)
    yy, xx = np.mgrid[:48, :64].astype(np.float64)
    sx = minv[0] * xx + minv[1] * yy + minv[2]
    sy = minv[3] * xx + minv[4] * yy + minv[5]
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    inside = (x0 >= 0) & (x0 < 55) & (y0 >= 0) & (y0 < 39)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    xs, ys = np.clip(x0, 0, 54), np.clip(y0, 0, 38)
    f = img.astype(np.float64)
    ref = f[ys, xs] * (1 - fy) * (1 - fx) + f[ys, xs + 1] * (1 - fy) * fx + f[ys + 1, xs] * fy * (1 - fx) + f[ys + 1, xs + 1] * fy * fx
    # the 1/32-pixel coordinate grid moves a sample by up to 1/64 pixel: bound the error by the local gradient
    assert np.abs(got - ref)[inside].max() <= 6.0 and np.abs(got - ref)[inside].mean() < 0.6


# ---------------------------------------------------------------------------------------------- box bookkeeping
def _random_header(g, n_boxes, n_rows=8):
    rows = np.full((n_rows, 6), -1.0)
    for i in range(n_boxes):
        x0, y0 = g.uniform(0, .8), g.uniform(0, .8)
        rows[i] = [g.integers(0, 8), x0, y0, x0 + g.uniform(.01, .3), y0 + g.uniform(.01, .3), g.uniform(0, 1)]
    return np.array([2 + 6 * n_rows, 2, 6] + rows.reshape(-1).tolist())


@pytest.mark.parametrize("seed", range(12))
def test_box_bookkeeping_matches_restatement(seed):
    g = np.random.Generator(np.random.PCG64(seed))
    data_shape = (3, 64, 128)
    n = [0, 1, 1, 2, 5, 7, 3, 1, 4, 6, 2, 8][seed]
    hdr = _random_header(g, n)
    aug = [float(g.random() > .5), np.radians(g.uniform(-5, 5)), g.uniform(.5, 2.), 0, 0, 0]
    aug[3] = aug[2] * g.uniform(.8, 1.2)
    aug[4] = -g.random() * 128 * (aug[2] - 1.)
    aug[5] = -g.random() * 64 * (aug[3] - 1.)
    img = np.zeros((32, 64, 3), np.uint8); seg = np.zeros((32, 64), np.uint8)
    h_ref = hdr.copy()
    oa.get_augmented(img, h_ref, seg, data_shape, aug)
    h_got = hdr.copy()
    has = it.augmented_boxes(h_got[3:].reshape(-1, 6), data_shape, aug)
    assert has == (n > 0)
    np.testing.assert_allclose(h_got, h_ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(h_got == -1, h_ref == -1)
    h_ref2, h_got2 = hdr.copy(), hdr.copy()
    oa.get_resized(img, h_ref2, seg, data_shape)
    it.resized_boxes(h_got2[3:].reshape(-1, 6), data_shape)
    np.testing.assert_array_equal(h_got2, h_ref2)


def test_single_survivor_fills_six_rows():
    hdr = np.array([2 + 6 * 8, 2, 6] + [3, .2, .2, .6, .7, .4] + [-1.0] * 42)
    aug = [0.0, 0.0, 1.0, 1.0, 0.0, 0.0]
    rows = hdr[3:].reshape(-1, 6)
    it.augmented_boxes(rows, (3, 64, 128), aug)
    assert (rows[:6] == [3, .2, .2, .6, .7, .4]).all() and (rows[6:] == -1).all()


# ---------------------------------------------------------------------------------------------- device pipeline
def _write_dataset(root, n, g, hw=(96, 160), boxes=(0, 1, 3, 5, 2, 4, 1, 6)):
    os.makedirs(os.path.join(root, "cityscapes", "SegmentationClass"), exist_ok=True)
    from PIL import Image
    rec = recordio.MXIndexedRecordIO(os.path.join(root, "train.idx"), os.path.join(root, "train.rec"), "w")
    lines = []
    for i in range(n):
        img = _smooth_image(g, *hw)
        hdr = _random_header(g, boxes[i % len(boxes)], 10)
        name = "JPEGImages/city_%06d_leftImg8bit.jpg" % i
        rec.write_idx(i, recordio.pack_img(recordio.IRHeader(0, hdr[1:].astype(np.float32), i, 0), img, quality=92))
        seg = g.integers(0, 19, hw).astype(np.uint8)
        seg[g.random(hw) < .1] = 255
        seg[:4, :4] = 40                                        # an id outside the LUT's identity range -> 255
        Image.fromarray(seg).save(os.path.join(root, "cityscapes", "SegmentationClass",
                                               "city_%06d_gtFine_labelTrainIds.png" % i))
        lines.append("%d\t%s\n" % (i, name))
    rec.close()
    with open(os.path.join(root, "train.lst"), "w") as f:
        f.writelines(lines)
    return os.path.join(root, "train.rec")


def _reference_batch(itr, rd, positions, enable_aug):
    """the restatement's batch for epoch positions `positions` of iterator `itr` (same decode, same parameters);
    rd: a reader of its own (the iterator's handle is busy preparing the next batch)"""
    data_shape = itr.data_shape
    datas, segs, labels = [], [], []
    for pos in positions:
        header, img = recordio.unpack_img(rd.read_idx(int(itr.index_table[pos])))
        hdr = np.array([header.label.shape[0]] + header.label.tolist())
        seg = recordio.imdecode(open(itr.imglst[str(header.id)], "rb").read())
        if enable_aug:
            img2, hdr, seg2 = oa.get_augmented(img, hdr, seg, data_shape, itr.aug_params[pos])
        else:
            img2, hdr, seg2 = oa.get_resized(img, hdr, seg, data_shape)
        d, s = oa.finish_sample(img2, seg2, data_shape, itr.mean_pixels, oa.seg_lut())
        row = np.ones(1206) * -1
        row[:3] = data_shape
        row[3:3 + hdr.shape[0]] = hdr
        datas.append(d); segs.append(s); labels.append(row[6:1206].reshape(200, 6).astype(np.float32))
    return np.stack(datas), np.stack(segs), np.stack(labels)


@pytest.mark.gpu
@pytest.mark.parametrize("enable_aug", [True, False])
def test_record_iter_matches_restatement(gpu_device, tmp_path, enable_aug):
    g = np.random.Generator(np.random.PCG64(5))
    n, B = 11, 4
    path = _write_dataset(str(tmp_path), n, g)
    itr = it.MultiTaskRecordIter(path, B, (3, 64, 128), enable_aug=enable_aug, device=gpu_device)
    rd = recordio.MXIndexedRecordIO(path.replace(".rec", ".idx"), path, "r")
    assert itr.provide_label == [("label_det", (B, 200, 6)), ("seg_out_label", (B, 16, 32))]
    assert itr.provide_data == [("data", [B, 3, 64, 128])]
    # the reference's stream: seed, shuffle, draws, first batch, then reset() = shuffle + draws
    rs = np.random.RandomState(233)
    table = np.arange(n); rs.shuffle(table)
    for _ in range(6):
        rs.rand(n)
    rs.shuffle(table)
    np.testing.assert_array_equal(itr.index_table, table)
    flips = 0
    for epoch in range(2):
        seen = 0
        while itr.iter_next():
            pos = list(range(itr.curr_index, itr.curr_index + B))
            state = copy.deepcopy(itr.aug_params)
            batch, fnames = itr.next()
            assert len(fnames) == B and fnames[0].endswith("_gtFine_labelTrainIds.png")
            np.testing.assert_array_equal(state, itr.aug_params)
            d_ref, s_ref, l_ref = _reference_batch(itr, rd, pos, enable_aug)
            np.testing.assert_array_equal(batch.data[0].cpu().numpy(), d_ref)
            np.testing.assert_array_equal(batch.label[1].cpu().numpy(), s_ref)
            np.testing.assert_allclose(batch.label[0].cpu().numpy(), l_ref, rtol=1e-6, atol=1e-7)
            np.testing.assert_array_equal(batch.label[0].cpu().numpy() == -1, l_ref == -1)
            flips += int((itr.aug_params[pos, 0] > .5).sum())
            seen += B
        assert seen == (n // B) * B                             # the partial last batch is dropped
        with pytest.raises(StopIteration):
            itr.next()
        itr.reset()
    if enable_aug:
        assert flips > 0
        assert set(np.unique(batch.label[1].cpu().numpy())).issubset(set(range(19)) | {255.0})


@pytest.mark.gpu
def test_augment_kernel_random_affines(gpu_device):
    """the C entry directly: random source sizes, affines that leave the image on every side, both flips"""
    import ctypes as c
    import torch
    from dspnet_amd import _lib
    g = np.random.Generator(np.random.PCG64(9))
    B, H, W = 6, 48, 80
    imgs = [_smooth_image(g, int(g.integers(20, 70)), int(g.integers(20, 90))) for _ in range(B)]
    segs = [g.integers(0, 256, im.shape[:2]).astype(np.uint8) for im in imgs]
    samples = np.zeros(B, it._SAMPLE)
    Ms, io, so = [], 0, 0
    for b, im in enumerate(imgs):
        th = g.uniform(-.5, .5)
        sx, sy = g.uniform(.4, 2.5), g.uniform(.4, 2.5)
        M = [sx * np.cos(th), -sy * np.sin(th), g.uniform(-30, 30), sx * np.sin(th), sy * np.cos(th), g.uniform(-20, 20)]
        Ms.append(M)
        samples[b] = (io, so if b != 2 else -1, im.shape[0], im.shape[1], b % 2, 128 if b % 3 else 7, 255 if b % 3 else 3, 0,
                      it.invert_affine(M))
        io += im.size
        so += segs[b].size
    dev = gpu_device
    ipool = torch.from_numpy(np.concatenate([i.reshape(-1) for i in imgs])).to(dev)
    spool = torch.from_numpy(np.concatenate([s.reshape(-1) for s in segs])).to(dev)
    desc = torch.from_numpy(samples.view(np.uint8).reshape(-1).copy()).to(dev)
    lut = np.arange(256, dtype=np.uint8)[::-1].copy()
    data = torch.empty(B, 3, H, W, device=dev)
    seg_out = torch.empty(B, H // 4, W // 4, device=dev)
    mean = [123.68, 116.779, 103.939]
    _lib.check(it._entry()(ipool.data_ptr(), spool.data_ptr(), desc.data_ptr(), B, H, W, it._CMAP_BGR, (c.c_double * 3)(*mean),
                           torch.from_numpy(lut).to(dev).data_ptr(), data.data_ptr(), seg_out.data_ptr(),
                           torch.cuda.current_stream(dev).cuda_stream), "augment")
    got, got_seg = data.cpu().numpy(), seg_out.cpu().numpy()
    for b in range(B):
        s = samples[b]
        w = oa.warp_affine(imgs[b], np.array(Ms[b]).reshape(2, 3), (W, H), True, int(s["img_border"]))
        ws = oa.warp_affine(segs[b], np.array(Ms[b]).reshape(2, 3), (W, H), False, int(s["seg_border"]))
        if s["flip"]:
            w, ws = w[:, ::-1], ws[:, ::-1]
        d, q = oa.finish_sample(w, ws, (3, H, W), mean, lut.astype(np.float64))
        np.testing.assert_array_equal(got[b], d)
        np.testing.assert_array_equal(got_seg[b], q if b != 2 else np.zeros_like(q))
    # argument checks
    assert it._entry()(ipool.data_ptr(), spool.data_ptr(), desc.data_ptr(), B, 50, W, it._CMAP_BGR, (c.c_double * 3)(*mean),
                       0, data.data_ptr(), seg_out.data_ptr(), 0) != 0


@pytest.mark.gpu
def test_fit_from_records_and_checkpoint(gpu_device, tmp_path):
    """iterator -> solver epoch loop -> checkpoint -> evaluation, the chain multi_train.py / multi_solver.fit drive"""
    import math
    from dspnet_amd import model
    from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
    from dspnet_amd.train.solver import MultiTaskSolver, do_checkpoint, fit
    g = np.random.Generator(np.random.PCG64(21))
    path = _write_dataset(str(tmp_path), 6, g, hw=(128, 128))
    B, S = 2, 128
    train_iter = it.MultiTaskRecordIter(path, B, (3, S, S), enable_aug=True, device=gpu_device)
    eval_iter = it.MultiTaskRecordIter(path, B, (3, S, S), enable_aug=False, device=gpu_device)
    net = get_multi_symbol_train("resnet-50", S, num_classes=8, batch_size=B, device=gpu_device)
    solver = MultiTaskSolver(net, learning_rate=0.0005)
    seen = []
    hist = fit(solver, train_iter, num_epoch=2, batch_end_callback=lambda p: seen.append((p.epoch, p.nbatch)),
               epoch_end_callback=do_checkpoint(str(tmp_path / "dspnet")), eval_data=eval_iter,
               class_names=["c%d" % i for i in range(8)], seg_class_names=["s%d" % i for i in range(19)])
    assert seen == [(0, 1), (0, 2), (0, 3), (1, 1), (1, 2), (1, 3)] and len(hist) == 2
    for h in hist:
        assert all(math.isfinite(h[k]) for k in ("CrossEntropy", "SmoothL1", "SegCrossEntropy", "accuracy"))
        assert math.isfinite(h["validation"]["mIoU"]) and "mAP" in h["validation"]
    _, args, _ = model.load_checkpoint(str(tmp_path / "dspnet"), 2)
    now = net.g.get_params()
    for k, v in now.items():
        np.testing.assert_array_equal(args[k], v)


@pytest.mark.gpu
def test_record_iter_missing_label_map_and_grey_record(gpu_device, tmp_path):
    """enable_aug=False tolerates a missing label PNG (cv2.imread returns None there: the label map stays 0,
    dataset/iterator.py:571-575); a single-channel JPEG record decodes to three equal planes like cv2.imdecode(..., 1)"""
    from PIL import Image
    import io
    g = np.random.Generator(np.random.PCG64(8))
    path = _write_dataset(str(tmp_path), 4, g, hw=(64, 96))
    os.remove(os.path.join(str(tmp_path), "cityscapes", "SegmentationClass", "city_000001_gtFine_labelTrainIds.png"))
    # rewrite record 2 as a greyscale JPEG
    rd = recordio.MXIndexedRecordIO(path.replace(".rec", ".idx"), path, "r")
    items = [rd.read_idx(i) for i in range(4)]
    rd.close()
    h2, img2 = recordio.unpack_img(items[2])
    buf = io.BytesIO()
    Image.fromarray(img2[:, :, 1]).save(buf, format="JPEG", quality=90)
    items[2] = recordio.pack(recordio.IRHeader(0, h2.label, h2.id, 0), buf.getvalue())
    w = recordio.MXIndexedRecordIO(path.replace(".rec", ".idx"), path, "w")
    for i, it_ in enumerate(items):
        w.write_idx(i, it_)
    w.close()
    itr = it.MultiTaskRecordIter(path, 4, (3, 32, 64), enable_aug=False, device=gpu_device, prefetch=False)
    batch, fnames = itr.next()
    order = [int(k) for k in itr.index_table[:4]]
    seg = batch.label[1].cpu().numpy()
    data = batch.data[0].cpu().numpy()
    assert (seg[order.index(1)] == 0).all() and (seg[order.index(0)] != 0).any()
    grey = np.rint(data[order.index(2)].astype(np.float64) + np.asarray(itr.mean_pixels)[:, None, None])
    np.testing.assert_array_equal(grey[0], grey[1]); np.testing.assert_array_equal(grey[1], grey[2])
    with pytest.raises(AssertionError, match="not found"):
        it.MultiTaskRecordIter(path, 4, (3, 32, 64), enable_aug=True, device=gpu_device, prefetch=False)


def test_augment_restatement_against_committed_vectors():
    """regression pin of oracle/augment.py itself: tests/golden/augment_warp.npz (generator: make_augment_golden.py)"""
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment_warp.npz"))
    np.testing.assert_array_equal(oa.warp_affine(z["img"], z["M"], (32, 16), True, 128), z["linear_border128"])
    np.testing.assert_array_equal(oa.warp_affine(z["seg"], z["M"], (32, 16), False, 255), z["nearest_border255"])
    h = z["hdr_in"].copy()
    oa.get_augmented(np.zeros((24, 40, 3), np.uint8), h, np.zeros((24, 40), np.uint8), (3, 16, 32), list(z["aug"]))
    np.testing.assert_array_equal(h, z["hdr_out"])
    rows = z["hdr_in"].copy()
    it.augmented_boxes(rows[3:].reshape(-1, 6), (3, 16, 32), list(z["aug"]))
    np.testing.assert_allclose(rows, z["hdr_out"], rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_record_iter_decoded_cache_is_transparent(gpu_device, tmp_path):
    """cache_decoded=True (pixels kept in host memory after the first decode) yields the same batches"""
    import torch
    g = np.random.Generator(np.random.PCG64(13))
    path = _write_dataset(str(tmp_path), 8, g, hw=(64, 96))
    a = it.MultiTaskRecordIter(path, 4, (3, 32, 64), device=gpu_device)
    b = it.MultiTaskRecordIter(path, 4, (3, 32, 64), device=gpu_device, cache_decoded=True)
    for epoch in range(3):
        while a.iter_next():
            (ba, fa), (bb, fb) = a.next(), b.next()
            assert fa == fb
            assert torch.equal(ba.data[0], bb.data[0]) and torch.equal(ba.label[0], bb.label[0])
            assert torch.equal(ba.label[1], bb.label[1])
        a.reset(); b.reset()
    assert len(b._cache) == 8
