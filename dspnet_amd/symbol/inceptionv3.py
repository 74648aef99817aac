"""Inception-v3 backbone (counterpart of symbol/inceptionv3.py:10-168 of the reference).

Every `Conv` is Convolution(no bias) -> BatchNorm(fix_gamma=True, batch statistics) -> ReLU (:10-14), with
the reference's layer names; the towers use 1x1, 3x3 (stride 1 pad 1, stride 1 pad 0, stride 2 pad 0), 5x5
pad 2, 1x7 / 7x1 pad 3 and 1x3 / 3x1 pad 1 kernels, 3x3 average pooling (stride 1 pad 1, padding counted)
and 3x3 max pooling (stride 2 pad 0; stride 1 pad 1 in mixed_10).  The classifier tail (:162-167) is not
built.  Returns {internal name + '_output': Tensor}; the SSD preset reads `ch_concat_mixed_7_chconcat` and
`ch_concat_mixed_10_chconcat` (symbol/multitask_symbol_factory.py:43-53)."""
from .. import engine as E


def get_symbol(g, data, **kwargs):
    internals = {}

    def Conv(x, num_filter, kernel=(1, 1), stride=1, pad=(0, 0), name=None, suffix='', cin_logical=None):
        c = g.add(E.Conv(g, x, '%s%s_conv2d' % (name, suffix), num_filter, kernel, stride, pad,
                         cin_logical=cin_logical)).out
        # "auto": the BN-apply + ReLU runs inside the next convolution's loader when convolutions are the only readers
        # (tower interiors); outputs that feed a concat or a pooling layer are materialised
        return g.add(E.BatchNorm(g, c, '%s%s_batchnorm' % (name, suffix), fix_gamma=True, eps=0.001, relu=True,
                                 defer_apply="auto")).out

    def pool(x, kind, kernel, stride, pad, name):
        if kind == "max":
            return g.add(E.MaxPool(g, x, name, kernel, stride, pad)).out
        return g.add(E.AvgPool2d(g, x, name, kernel, stride, pad)).out

    def concat(parts, name):
        y = g.add(E.Concat(g, parts, 'ch_concat_%s_chconcat' % name)).out
        internals['ch_concat_%s_chconcat_output' % name] = y
        return y

    def Inception7A(x, n1, n3r, n3_1, n3_2, n5r, n5, kind, proj, name):          # :17-34
        t1 = Conv(x, n1, name='%s_conv' % name)
        t5 = Conv(x, n5r, name='%s_tower' % name, suffix='_conv')
        t5 = Conv(t5, n5, (5, 5), pad=(2, 2), name='%s_tower' % name, suffix='_conv_1')
        t3 = Conv(x, n3r, name='%s_tower_1' % name, suffix='_conv')
        t3 = Conv(t3, n3_1, (3, 3), pad=(1, 1), name='%s_tower_1' % name, suffix='_conv_1')
        t3 = Conv(t3, n3_2, (3, 3), pad=(1, 1), name='%s_tower_1' % name, suffix='_conv_2')
        p = pool(x, kind, 3, 1, 1, '%s_pool_%s_pool' % (kind, name))
        cp = Conv(p, proj, name='%s_tower_2' % name, suffix='_conv')
        return concat([t1, t5, t3, cp], name)

    def Inception7B(x, n3, nd3r, nd3_1, nd3_2, kind, name):                      # :37-49
        t3 = Conv(x, n3, (3, 3), stride=2, name='%s_conv' % name)
        td = Conv(x, nd3r, name='%s_tower' % name, suffix='_conv')
        td = Conv(td, nd3_1, (3, 3), pad=(1, 1), name='%s_tower' % name, suffix='_conv_1')
        td = Conv(td, nd3_2, (3, 3), stride=2, name='%s_tower' % name, suffix='_conv_2')
        p = pool(x, "max", 3, 2, 0, 'max_pool_%s_pool' % name)
        return concat([t3, td, p], name)

    def Inception7C(x, n1, nd7r, nd7_1, nd7_2, nq7r, nq7_1, nq7_2, nq7_3, nq7_4, kind, proj, name):   # :51-70
        t1 = Conv(x, n1, name='%s_conv' % name)
        td = Conv(x, nd7r, name='%s_tower' % name, suffix='_conv')
        td = Conv(td, nd7_1, (1, 7), pad=(0, 3), name='%s_tower' % name, suffix='_conv_1')
        td = Conv(td, nd7_2, (7, 1), pad=(3, 0), name='%s_tower' % name, suffix='_conv_2')
        tq = Conv(x, nq7r, name='%s_tower_1' % name, suffix='_conv')
        tq = Conv(tq, nq7_1, (7, 1), pad=(3, 0), name='%s_tower_1' % name, suffix='_conv_1')
        tq = Conv(tq, nq7_2, (1, 7), pad=(0, 3), name='%s_tower_1' % name, suffix='_conv_2')
        tq = Conv(tq, nq7_3, (7, 1), pad=(3, 0), name='%s_tower_1' % name, suffix='_conv_3')
        tq = Conv(tq, nq7_4, (1, 7), pad=(0, 3), name='%s_tower_1' % name, suffix='_conv_4')
        p = pool(x, kind, 3, 1, 1, '%s_pool_%s_pool' % (kind, name))
        cp = Conv(p, proj, name='%s_tower_2' % name, suffix='_conv')
        return concat([t1, td, tq, cp], name)

    def Inception7D(x, n3r, n3, nd7r, nd7_1, nd7_2, nd7_3x3, kind, name):         # :72-87
        t3 = Conv(x, n3r, name='%s_tower' % name, suffix='_conv')
        t3 = Conv(t3, n3, (3, 3), stride=2, name='%s_tower' % name, suffix='_conv_1')
        td = Conv(x, nd7r, name='%s_tower_1' % name, suffix='_conv')
        td = Conv(td, nd7_1, (1, 7), pad=(0, 3), name='%s_tower_1' % name, suffix='_conv_1')
        td = Conv(td, nd7_2, (7, 1), pad=(3, 0), name='%s_tower_1' % name, suffix='_conv_2')
        td = Conv(td, nd7_3x3, (3, 3), stride=2, name='%s_tower_1' % name, suffix='_conv_3')
        p = pool(x, kind, 3, 2, 0, '%s_pool_%s_pool' % (kind, name))
        return concat([t3, td, p], name)

    def Inception7E(x, n1, nd3r, nd3_1, nd3_2, n33r, n33, n33_1, n33_2, kind, proj, name):   # :89-108
        t1 = Conv(x, n1, name='%s_conv' % name)
        td = Conv(x, nd3r, name='%s_tower' % name, suffix='_conv')
        ta = Conv(td, nd3_1, (1, 3), pad=(0, 1), name='%s_tower' % name, suffix='_mixed_conv')
        tb = Conv(td, nd3_2, (3, 1), pad=(1, 0), name='%s_tower' % name, suffix='_mixed_conv_1')
        t3 = Conv(x, n33r, name='%s_tower_1' % name, suffix='_conv')
        t3 = Conv(t3, n33, (3, 3), pad=(1, 1), name='%s_tower_1' % name, suffix='_conv_1')
        t3a = Conv(t3, n33_1, (1, 3), pad=(0, 1), name='%s_tower_1' % name, suffix='_mixed_conv')
        t3b = Conv(t3, n33_2, (3, 1), pad=(1, 0), name='%s_tower_1' % name, suffix='_mixed_conv_1')
        p = pool(x, kind, 3, 1, 1, '%s_pool_%s_pool' % (kind, name))
        cp = Conv(p, proj, name='%s_tower_2' % name, suffix='_conv')
        return concat([t1, ta, tb, t3a, t3b, cp], name)

    x = g.add(E.InputNCHW(g, data)).out
    # stage 1 (:114-118)
    x = Conv(x, 32, (3, 3), stride=2, name="conv", cin_logical=3)
    x = Conv(x, 32, (3, 3), name="conv_1")
    x = Conv(x, 64, (3, 3), pad=(1, 1), name="conv_2")
    x = pool(x, "max", 3, 2, 0, "pool")
    # stage 2 (:120-122)
    x = Conv(x, 80, (1, 1), name="conv_3")
    x = Conv(x, 192, (3, 3), name="conv_4")
    x = pool(x, "max", 3, 2, 0, "pool1")
    # stage 3 (:124-139)
    x = Inception7A(x, 64, 64, 96, 96, 48, 64, "avg", 32, "mixed")
    x = Inception7A(x, 64, 64, 96, 96, 48, 64, "avg", 64, "mixed_1")
    x = Inception7A(x, 64, 64, 96, 96, 48, 64, "avg", 64, "mixed_2")
    x = Inception7B(x, 384, 64, 96, 96, "max", "mixed_3")
    # stage 4 (:141-160)
    x = Inception7C(x, 192, 128, 128, 192, 128, 128, 128, 128, 192, "avg", 192, "mixed_4")
    x = Inception7C(x, 192, 160, 160, 192, 160, 160, 160, 160, 192, "avg", 192, "mixed_5")
    x = Inception7C(x, 192, 160, 160, 192, 160, 160, 160, 160, 192, "avg", 192, "mixed_6")
    x = Inception7C(x, 192, 192, 192, 192, 192, 192, 192, 192, 192, "avg", 192, "mixed_7")
    x = Inception7D(x, 192, 320, 192, 192, 192, 192, "max", "mixed_8")
    # stage 5 (:162-169)
    x = Inception7E(x, 320, 384, 384, 384, 448, 384, 384, 384, "avg", 192, "mixed_9")
    x = Inception7E(x, 320, 384, 384, 384, 448, 384, 384, 384, "max", 192, "mixed_10")
    internals["_output"] = x
    return internals
