"""Timing of the (f)-row device kernels: dspn_augment_batch_u8 (record iterator) and dspn_seg_upsample_argmax_f32
(full-resolution class map), plus the iterator end to end on a synthetic Cityscapes-sized record file.
usage: python scratch/aux_bench.py [--iter]"""
import ctypes as c
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dspnet_amd import _lib, functional as fn          # noqa: E402
from dspnet_amd.dataset import iterator as it, recordio   # noqa: E402


def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = torch.device("cuda", 0)
    g = np.random.Generator(np.random.PCG64(0))
    B, H, W, sh, sw = 32, 512, 512, 1024, 2048
    img = torch.from_numpy(g.integers(0, 256, (B, sh, sw, 3), dtype=np.uint8)).to(dev)
    seg = torch.from_numpy(g.integers(0, 19, (B, sh, sw), dtype=np.uint8)).to(dev)
    samples = np.zeros(B, it._SAMPLE)
    for b in range(B):
        th, sx = np.radians(g.uniform(-5, 5)), g.uniform(.5, 2.)
        sy = sx * g.uniform(.8, 1.2)
        M = [sx * W / sw * np.cos(th), -sy * H / sh * np.sin(th), -g.random() * W * (sx - 1), sx * W / sw * np.sin(th),
             sy * H / sh * np.cos(th), -g.random() * H * (sy - 1)]
        samples[b] = (b * sh * sw * 3, b * sh * sw, sh, sw, b % 2, 128, 255, 0, it.invert_affine(M))
    desc = torch.from_numpy(samples.view(np.uint8).reshape(-1).copy()).to(dev)
    lut = torch.from_numpy(it.seg_lut()).to(dev)
    data = torch.empty(B, 3, H, W, device=dev)
    so = torch.empty(B, H // 4, W // 4, device=dev)
    mean = (c.c_double * 3)(123.68, 116.779, 103.939)
    st = torch.cuda.current_stream(dev).cuda_stream
    ms = timed(lambda: _lib.check(it._entry()(img.data_ptr(), seg.data_ptr(), desc.data_ptr(), B, H, W, it._CMAP_BGR, mean,
                                              lut.data_ptr(), data.data_ptr(), so.data_ptr(), st)))
    out_bytes = B * (3 * H * W * 4 + H * W // 16 * 4)
    print("augment_batch  B=%d %dx%d -> 3x%dx%d: %.3f ms  (%.0f img/s; writes %.1f MB -> %.0f GB/s on the output alone)"
          % (B, sh, sw, H, W, ms, B / ms * 1e3, out_bytes / 1e6, out_bytes / ms / 1e6))
    prob = torch.softmax(torch.randn(1, 128, 256, 20, device=dev), -1).contiguous()
    ms2 = timed(lambda: fn.seg_upsample_argmax(prob, 19, 1024, 2048))
    print("seg_upsample_argmax 1x128x256x19 -> 1024x2048: %.3f ms  (%.1f MB algorithmic -> %.0f GB/s)"
          % (ms2, (prob.numel() * 4 + 1024 * 2048) / 1e6, (prob.numel() * 4 + 1024 * 2048) / ms2 / 1e6))
    ref = lambda: torch.nn.functional.interpolate(prob.permute(0, 3, 1, 2)[:, :19], size=(1024, 2048), mode="bilinear",  # noqa: E731
                                                  align_corners=True).argmax(1)
    print("   (torch interpolate + argmax on the same input: %.3f ms)" % timed(ref, 5))
    if "--iter" in sys.argv:
        from PIL import Image
        root = tempfile.mkdtemp()
        os.makedirs(os.path.join(root, "cityscapes", "SegmentationClass"))
        rec = recordio.MXIndexedRecordIO(os.path.join(root, "t.idx"), os.path.join(root, "t.rec"), "w")
        n = 64
        yy, xx = np.mgrid[:sh, :sw]
        base = (127 + 90 * np.sin(xx / 37.) * np.cos(yy / 23.))
        with open(os.path.join(root, "t.lst"), "w") as f:
            for i in range(n):
                im = np.clip(base[..., None] + g.normal(0, 12, (sh, sw, 3)), 0, 255).astype(np.uint8)
                rows = np.full((20, 6), -1.0)
                for k in range(8):
                    x0, y0 = g.uniform(0, .7), g.uniform(0, .7)
                    rows[k] = [k % 8, x0, y0, x0 + .2, y0 + .2, .3]
                lab = np.array([2, 6] + rows.reshape(-1).tolist(), np.float32)
                rec.write_idx(i, recordio.pack_img(recordio.IRHeader(0, lab, i, 0), im, quality=90))
                Image.fromarray(g.integers(0, 19, (sh, sw)).astype(np.uint8)).save(
                    os.path.join(root, "cityscapes", "SegmentationClass", "c_%04d_gtFine_labelTrainIds.png" % i))
                f.write("%d\tJPEGImages/c_%04d_leftImg8bit.jpg\n" % (i, i))
        rec.close()
        for threads in (1, 8, 32):
            itr = it.MultiTaskRecordIter(os.path.join(root, "t.rec"), 32, (3, 512, 512), device=dev, decode_threads=threads)
            t0 = time.time()
            cnt = 0
            while itr.iter_next():
                itr.next(); cnt += 32
            torch.cuda.synchronize()
            print("iterator end to end (decode 1024x2048 JPEG + PNG on %d host threads, %d cores): %.0f img/s"
                  % (threads, os.cpu_count(), cnt / (time.time() - t0)))
        itr = it.MultiTaskRecordIter(os.path.join(root, "t.rec"), 32, (3, 512, 512), device=dev, decode_threads=8, cache_decoded=True)
        for epoch in range(3):
            t0 = time.time()
            cnt = 0
            while itr.iter_next():
                itr.next(); cnt += 32
            torch.cuda.synchronize()
            print("   cache_decoded, 8 threads, epoch %d: %.0f img/s" % (epoch, cnt / (time.time() - t0)))
            itr.reset()


if __name__ == "__main__":
    main()
