"""Small nn building blocks, API- and state_dict-compatible with the reference's
`pointnet2/pytorch_utils.py`: `SharedMLP` (:11-36), `BatchNorm1d/2d/3d` (:39-64),
`Conv1d/2d/3d` (:67-222), `FC` (:225-262), `BNMomentumScheduler` (:264-298).

Parameter names are the checkpoint contract (e.g. `layer0.conv.weight`, `layer0.bn.bn.weight`,
`layer0.bn.bn.running_mean`): every wrapper below registers its children under exactly the
names the reference uses, so released `.pth` files load unchanged.
"""
import torch
import torch.nn as nn


class _BNBase(nn.Sequential):
    """A BatchNorm wrapped in a one-element Sequential under the child name `<name>bn`."""

    def __init__(self, in_size, batch_norm=None, name=""):
        super().__init__()
        self.add_module(name + "bn", batch_norm(in_size))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0)


class BatchNorm1d(_BNBase):
    def __init__(self, in_size, *, name=""):
        super().__init__(in_size, batch_norm=nn.BatchNorm1d, name=name)


class BatchNorm2d(_BNBase):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, batch_norm=nn.BatchNorm2d, name=name)


class BatchNorm3d(_BNBase):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, batch_norm=nn.BatchNorm3d, name=name)


def _norm_act(seq, name, bn, norm_cls, width, activation):
    if bn:
        seq.add_module(name + "bn", norm_cls(width))
    if activation is not None:
        seq.add_module(name + "activation", activation)


class _ConvBase(nn.Sequential):
    """conv -> [bn] -> [activation]   (or [bn] -> [activation] -> conv with preact=True).
    The conv carries a bias only when no BN follows."""

    def __init__(self, in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                 conv=None, batch_norm=None, bias=True, preact=False, name=""):
        super().__init__()
        use_bias = bias and (not bn)
        unit = conv(in_size, out_size, kernel_size=kernel_size, stride=stride, padding=padding,
                    bias=use_bias)
        init(unit.weight)
        if use_bias:
            nn.init.constant_(unit.bias, 0)
        if preact:
            _norm_act(self, name, bn, batch_norm, in_size, activation)
        self.add_module(name + "conv", unit)
        if not preact:
            _norm_act(self, name, bn, batch_norm, out_size, activation)


class Conv1d(_ConvBase):
    def __init__(self, in_size, out_size, *, kernel_size=1, stride=1, padding=0,
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name=""):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                         conv=nn.Conv1d, batch_norm=BatchNorm1d, bias=bias, preact=preact, name=name)


class Conv2d(_ConvBase):
    def __init__(self, in_size, out_size, *, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_, bias=True,
                 preact=False, name=""):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                         conv=nn.Conv2d, batch_norm=BatchNorm2d, bias=bias, preact=preact, name=name)


class Conv3d(_ConvBase):
    def __init__(self, in_size, out_size, *, kernel_size=(1, 1, 1), stride=(1, 1, 1),
                 padding=(0, 0, 0), activation=nn.ReLU(inplace=True), bn=False,
                 init=nn.init.kaiming_normal_, bias=True, preact=False, name=""):
        super().__init__(in_size, out_size, kernel_size, stride, padding, activation, bn, init,
                         conv=nn.Conv3d, batch_norm=BatchNorm3d, bias=bias, preact=preact, name=name)


class SharedMLP(nn.Sequential):
    """Per-point MLP: a stack of 1x1 Conv2d(+BN+ReLU) named layer0, layer1, ..."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False, first=False,
                 name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0      # a pre-activated first layer stays bare
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(args[i], args[i + 1], bn=bn and not plain,
                       activation=None if plain else activation, preact=preact))


class FC(nn.Sequential):
    def __init__(self, in_size, out_size, *, activation=nn.ReLU(inplace=True), bn=False, init=None,
                 preact=False, name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        if preact:
            _norm_act(self, name, bn, BatchNorm1d, in_size, activation)
        self.add_module(name + "fc", fc)
        if not preact:
            _norm_act(self, name, bn, BatchNorm1d, out_size, activation)


def feature_dropout_no_scaling(x, p, train, inplace=False):
    """Zero whole feature channels with probability p, without the 1/(1-p) rescale."""
    training = train() if callable(train) else bool(train)
    if not training or float(p) <= 0.0:
        return x
    keep = (torch.rand(x.shape[:2], device=x.device) >= float(p)).to(x.dtype)
    keep = keep.reshape(*x.shape[:2], *([1] * (x.dim() - 2)))
    return x.mul_(keep) if inplace else x * keep


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """Sets every BatchNorm's momentum to bn_lambda(epoch) on each step()."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model = model
        self.setter = setter
        self.lmbd = bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))
