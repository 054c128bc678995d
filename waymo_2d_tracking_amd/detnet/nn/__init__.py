"""Model factory - mirrors create() / save() / load() of /root/reference/detnet/nn/__init__.py:9-63 for the one model family
on the Waymo hot path (``detectron2:<yaml>`` arch strings).

File format (reference :47-54): ``torch.save({'args': args, 'kwargs': kwargs, 'state_dict': net.state_dict()})`` where the
state dict carries detectron2's parameter names under ``model.`` (Detectron2Det.model is a detectron2 GeneralizedRCNN there).
``load`` reads such a file - or this package's own files - into the MI355X-native graph (weights.py folds FrozenBN, permutes
fc1); ``save`` writes the reference's format so that files travel both ways.
"""
import functools
import os


def add_save_and_load(create_func):
    """reference nn/__init__.py:9-15: every created net gets .save(filename) / .load."""
    @functools.wraps(create_func)
    def extended_create_func(*args, **kwargs):
        net = create_func(*args, **kwargs)
        net.save = functools.partial(save, net=net, args=args, kwargs=kwargs)
        net.load = load
        return net
    return extended_create_func


@add_save_and_load
def create(arch, classnames=None, basenet=None, pretrained='imagenet', freeze_pretrained=0, frozen_bn=False, **kw):
    """nn/__init__.py:18-44: only ``detectron2:...X_152...`` architectures are built (the solution's model); every other
    arch of the reference (SSD, pointdet, torchvision, mmdet) is outside the hot path."""
    from .detectron2_det import Detectron2Det, WAYMO_CLASSNAMES
    if classnames and classnames[0] == 'background':           # :22-24 compatibility for old models
        print('Warning: removing "background" from classnames')
        classnames = classnames[1:]
    if arch.split(':')[0] != 'detectron2':
        raise NotImplementedError('arch %r is outside the Cascade R-CNN hot path' % arch)
    return Detectron2Det(arch.split(':')[1], classnames or WAYMO_CLASSNAMES, freeze_pretrained, frozen_bn,
                         pretrained=pretrained, **kw)


def save(filename, net, args, kwargs):
    """nn/__init__.py:47-54 (same argument order).  The state dict is written in detectron2 naming."""
    import torch
    from torch import nn
    from . import weights
    if isinstance(net, nn.DataParallel):
        net = net.module
    data = dict(args=args, kwargs=kwargs, state_dict=weights.export_state_dict_detectron2(net.model))
    torch.save(data, filename)


def load(filename):
    """nn/__init__.py:57-68: ``detectron2:<yaml>`` builds the model with its COCO weights (pretrained='coco'); otherwise a
    file written by save() here or by the reference."""
    import torch
    from . import weights
    print('load {}'.format(filename))
    if isinstance(filename, str) and filename.startswith('detectron2:'):
        return create(filename, classnames=None, pretrained='coco')
    data = torch.load(os.fspath(filename), map_location='cpu', weights_only=False)
    if not isinstance(data, dict) or 'state_dict' not in data or 'args' not in data:
        raise ValueError('%s is not a {args, kwargs, state_dict} model file (detnet/nn/__init__.py:47-54)' % filename)
    kwargs = dict(data['kwargs'])
    kwargs['pretrained'] = None                               # :64 the weights come from the file
    net = create(*data['args'], **kwargs)
    sd = data['state_dict']
    if weights.is_detectron2_state_dict(sd):
        weights.load_state_dict_detectron2(net.model, sd, strict=True)
    else:
        net.load_state_dict(sd)                               # files of this package's round 1 (native parameter names)
    return net
