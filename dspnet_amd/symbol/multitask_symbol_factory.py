"""Network presets (counterpart of symbol/multitask_symbol_factory.py)."""
from . import multitask_symbol_builder as builder


def get_config(network, data_shape, **kwargs):
    """symbol/multitask_symbol_factory.py:5-98.  Only the presets that build in the reference are
    offered (resnet-50 :68-81) plus vgg16_reduced (:17-42) and inceptionv3 (:43-53), whose multi-task wiring
    is this build's (their presets do not build in the reference)."""
    if isinstance(data_shape, (tuple, list)):
        data_shape = data_shape[1]
    if network == 'vgg16_reduced':   # symbol/multitask_symbol_factory.py:17-42
        if data_shape >= 448:
            from_layers = ['relu4_3', 'relu7', '', '', '', '', '']
            num_filters = [512, -1, 512, 256, 256, 256, 256]
            strides = [-1, -1, 2, 2, 2, 2, 1]
            pads = [-1, -1, 1, 1, 1, 1, 1]
            sizes = [[.07, .1025], [.15, .2121], [.3, .3674], [.45, .5196], [.6, .6708], [.75, .8216], [.9, .9721]]
            ratios = [[1, 2, .5], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3],
                      [1, 2, .5, 3, 1. / 3], [1, 2, .5], [1, 2, .5]]
            normalizations = [20, -1, -1, -1, -1, -1, -1]
            steps = [] if data_shape != 512 else [x / 512.0 for x in [8, 16, 32, 64, 128, 256, 512]]
        else:
            from_layers = ['relu4_3', 'relu7', '', '', '', '']
            num_filters = [512, -1, 512, 256, 256, 256]
            strides = [-1, -1, 2, 2, 1, 1]
            pads = [-1, -1, 1, 1, 0, 0]
            sizes = [[.1, .141], [.2, .272], [.37, .447], [.54, .619], [.71, .79], [.88, .961]]
            ratios = [[1, 2, .5], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3], [1, 2, .5],
                      [1, 2, .5]]
            normalizations = [20, -1, -1, -1, -1, -1]
            steps = [] if data_shape != 300 else [x / 300.0 for x in [8, 16, 32, 64, 100, 300]]
        return locals()
    if network == 'inceptionv3':     # symbol/multitask_symbol_factory.py:43-53
        from_layers = ['ch_concat_mixed_7_chconcat', 'ch_concat_mixed_10_chconcat', '', '', '', '']
        num_filters = [-1, -1, 512, 256, 256, 128]
        strides = [-1, -1, 2, 2, 2, 2]
        pads = [-1, -1, 1, 1, 1, 1]
        sizes = [[.1, .141], [.2, .272], [.37, .447], [.54, .619], [.71, .79], [.88, .961]]
        ratios = [[1, 2, .5], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3],
                  [1, 2, .5], [1, 2, .5]]
        normalizations = -1
        steps = []
        return locals()
    if network == 'resnet-50':
        num_layers = 50
        image_shape = '3,224,224'  # (the reference hands it to symbol/resnet.py as a shape check; unused here)
        network = 'resnet'
        from_layers = ['_plus6', '_plus12', '_plus15', '', '', '', '']
        num_filters = [-1, -1, -1, 512, 256, 256, 128]
        strides = [-1, -1, -1, 2, 2, 2, 2]
        pads = [-1, -1, -1, 1, 1, 1, 1]
        sizes = [[.5, .705], [.1, .141], [.2, .272], [.37, .447], [.54, .619], [.71, .79], [.88, .961]]
        ratios = [[1, 2, .5], [1, 2, .5], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3],
                  [1, 2, .5], [1, 2, .5]]
        normalizations = -1
        steps = []
        return locals()
    if network == 'resnet101':       # symbol/multitask_symbol_factory.py:82-95 (table only: the multi-task builder reads
        num_layers = 101             # from_layers[2], which this six-entry preset leaves empty)
        image_shape = '3,224,224'
        network = 'resnet'
        from_layers = ['_plus12', '_plus15', '', '', '', '']
        num_filters = [-1, -1, 512, 256, 256, 128]
        strides = [-1, -1, 2, 2, 2, 2]
        pads = [-1, -1, 1, 1, 1, 1]
        sizes = [[.1, .141], [.2, .272], [.37, .447], [.54, .619], [.71, .79], [.88, .961]]
        ratios = [[1, 2, .5], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3], [1, 2, .5, 3, 1. / 3],
                  [1, 2, .5], [1, 2, .5]]
        normalizations = -1
        steps = []
        return locals()
    msg = 'No configuration found for %s with data_shape %s' % (network, str(data_shape))
    raise NotImplementedError(msg)


def get_multi_symbol_train(network, data_shape, **kwargs):
    """symbol/multitask_symbol_factory.py:188-205.  data_shape: int (square) or (C, H, W)."""
    if isinstance(data_shape, int):
        data_shape = (3, data_shape, data_shape)
    config = get_config(network, data_shape, **kwargs)
    config.pop('kwargs', None)
    config.pop('data_shape', None)
    kwargs = dict(kwargs)
    config.update(kwargs)
    return builder.get_multi_symbol_train(data_shape=data_shape, **config)


def get_det_symbol_train(network, data_shape, **kwargs):
    """symbol/multitask_symbol_factory.py:104-121 (detection + depth only)"""
    if isinstance(data_shape, int):
        data_shape = (3, data_shape, data_shape)
    config = get_config(network, data_shape, **kwargs)
    config.pop('kwargs', None)
    config.pop('data_shape', None)
    config.update(dict(kwargs))
    return builder.get_det_symbol_train(data_shape=data_shape, **config)


def get_multi_symbol(network, data_shape, **kwargs):
    """symbol/multitask_symbol_factory.py:207-224 (test graph: outputs [det, seg_out])"""
    if isinstance(data_shape, int):
        data_shape = (3, data_shape, data_shape)
    config = get_config(network, data_shape, **kwargs)
    config.pop('kwargs', None)
    config.pop('data_shape', None)
    config.update(dict(kwargs))
    return builder.get_multi_symbol(data_shape=data_shape, **config)


def _configured(build, network, data_shape, kwargs):
    if isinstance(data_shape, int):
        data_shape = (3, data_shape, data_shape)
    config = get_config(network, data_shape, **kwargs)
    config.pop('kwargs', None)
    config.pop('data_shape', None)
    config.update(dict(kwargs))
    return build(data_shape=data_shape, **config)


def get_det_symbol(network, data_shape, **kwargs):
    """symbol/multitask_symbol_factory.py:123-144 (detection + depth test graph: outputs [det])"""
    return _configured(builder.get_det_symbol, network, data_shape, kwargs)


def get_seg_symbol_train(network, data_shape, **kwargs):
    """symbol/multitask_symbol_factory.py:146-163 (segmentation only: outputs [seg_out])"""
    return _configured(builder.get_seg_symbol_train, network, data_shape, kwargs)


def get_seg_symbol(network, data_shape, **kwargs):
    """symbol/multitask_symbol_factory.py:165-186 (segmentation test graph: outputs [seg_out])"""
    return _configured(builder.get_seg_symbol, network, data_shape, kwargs)
