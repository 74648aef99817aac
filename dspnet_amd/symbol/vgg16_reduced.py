"""VGG16 with fc6/fc7 as (dilated) convolutions (counterpart of symbol/vgg16_reduced.py:3-86).

conv3x3+bias -> ReLU (fused epilogue) x13, max-pool 2x2/2 (pool3 with pooling_convention='full'),
pool5 3x3/1 pad 1, fc6 = 3x3 dilate 6 pad 6 (1024), fc7 = 1x1 (1024).  The classifier tail (:76-85) is
not built.  Returns {internal name + '_output': Tensor}."""
from .. import engine as E


def get_symbol(g, data, **kwargs):
    internals = {}
    x = g.add(E.InputNCHW(g, data)).out

    def conv(x, name, nf, relu_name, kernel=3, pad=1, dilate=1, cin_logical=None):
        y = g.add(E.Conv(g, x, name, nf, kernel, 1, pad, dilate, no_bias=False, relu=True, init="xavier",
                         cin_logical=cin_logical)).out
        internals[relu_name + "_output"] = y
        return y

    x = conv(x, "conv1_1", 64, "relu1_1", cin_logical=3)
    x = conv(x, "conv1_2", 64, "relu1_2")
    x = g.add(E.MaxPool(g, x, "pool1", 2, 2, 0)).out
    x = conv(x, "conv2_1", 128, "relu2_1")
    x = conv(x, "conv2_2", 128, "relu2_2")
    x = g.add(E.MaxPool(g, x, "pool2", 2, 2, 0)).out
    x = conv(x, "conv3_1", 256, "relu3_1")
    x = conv(x, "conv3_2", 256, "relu3_2")
    x = conv(x, "conv3_3", 256, "relu3_3")
    x = g.add(E.MaxPool(g, x, "pool3", 2, 2, 0, full=True)).out
    x = conv(x, "conv4_1", 512, "relu4_1")
    x = conv(x, "conv4_2", 512, "relu4_2")
    x = conv(x, "conv4_3", 512, "relu4_3")
    x = g.add(E.MaxPool(g, x, "pool4", 2, 2, 0)).out
    x = conv(x, "conv5_1", 512, "relu5_1")
    x = conv(x, "conv5_2", 512, "relu5_2")
    x = conv(x, "conv5_3", 512, "relu5_3")
    x = g.add(E.MaxPool(g, x, "pool5", 3, 1, 1)).out
    x = conv(x, "fc6", 1024, "relu6", kernel=3, pad=6, dilate=6)
    x = conv(x, "fc7", 1024, "relu7", kernel=1, pad=0)
    internals["_output"] = x
    return internals
