"""Model factory - mirrors create()/load() of /root/reference/detnet/nn/__init__.py:19-63 for the one model family
on the Waymo hot path (``detectron2:<yaml>`` arch strings)."""


def create(arch, classnames=None, basenet=None, pretrained=False, freeze_pretrained=2, frozen_bn=True, **kw):
    """nn/__init__.py:19: only ``detectron2:...X_152...`` architectures are built (the solution's model)."""
    from .detectron2_det import Detectron2Det, WAYMO_CLASSNAMES
    if not arch.startswith('detectron2:'):
        raise NotImplementedError('arch %r is outside the Cascade R-CNN hot path' % arch)
    return Detectron2Det(arch[len('detectron2:'):], classnames or WAYMO_CLASSNAMES, freeze_pretrained, frozen_bn, pretrained, **kw)


def load(filename):
    """nn/__init__.py:57-63: a ``detectron2:<yaml>`` string builds a fresh model, otherwise a file saved by save()."""
    import torch
    if str(filename).startswith('detectron2:'):
        return create(str(filename))
    data = torch.load(filename, map_location='cpu')
    model = create(*data['args'], **data['kwargs'])
    model.load_state_dict(data['state_dict'])
    return model


def save(model, filename, *args, **kwargs):
    """nn/__init__.py:47-54 file format: {args, kwargs, state_dict}."""
    import torch
    torch.save({'args': args or ('detectron2:' + model.arch,), 'kwargs': kwargs, 'state_dict': model.state_dict()}, filename)
