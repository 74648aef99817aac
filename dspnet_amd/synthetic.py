"""Synthetic Cityscapes-shaped batches.

Shapes and value ranges follow what the reference iterator emits
(dataset/iterator.py:524-571): data (B,3,H,W) = uint8 image minus the RGB mean
(123,117,104) (multi_train.py:62-67); label_det (B,200,6) rows
[cls, xmin, ymin, xmax, ymax, dist] padded with -1; seg label (B,H/4,W/4) in
{0..18} with 255 = ignore.  numpy only; seeds via PCG64 (seed 233 echoes
dataset/iterator.py:381)."""
import numpy as np

MEAN_RGB = (123.0, 117.0, 104.0)


def rng(seed=233):
    return np.random.Generator(np.random.PCG64(seed))


def det_labels(batch, num_labels=200, num_classes=8, max_gt=40, height=512, width=512, gen=None,
               first_empty=True):
    gen = gen or rng()
    lab = -np.ones((batch, num_labels, 6), np.float32)
    for b in range(batch):
        g = 0 if (first_empty and b == batch - 1 and batch > 1) else int(gen.integers(1, max_gt + 1))
        for k in range(min(g, num_labels)):
            # boxes of at least 10 px a side (area >= 100 px^2, dataset/iterator.py:524-526)
            w = gen.uniform(10.0 / width, 0.6)
            h = gen.uniform(10.0 / height, 0.6)
            x0 = gen.uniform(0.0, 1.0 - w)
            y0 = gen.uniform(0.0, 1.0 - h)
            lab[b, k] = [float(gen.integers(0, num_classes)), x0, y0, x0 + w, y0 + h,
                         gen.uniform(0.0, 1.0)]
    return lab


def images(batch, height=512, width=512, gen=None):
    gen = gen or rng()
    img = gen.integers(0, 256, size=(batch, 3, height, width)).astype(np.float32)
    img -= np.asarray(MEAN_RGB, np.float32).reshape(1, 3, 1, 1)
    return img


def seg_labels(batch, height=512, width=512, seg_classes=19, ignore_frac=0.1, gen=None):
    gen = gen or rng()
    lab = gen.integers(0, seg_classes, size=(batch, height // 4, width // 4)).astype(np.float32)
    lab[gen.random(lab.shape) < ignore_frac] = 255.0
    return lab
