"""Layer library with the reference's class / attribute / parameter names (models/layers.py).

Only what the search hot path touches is provided:
  MBInvertedResBlock  (models/layers.py:431-561)  -- parameter container with the reference's sub-module names
                      (``inverted_bottleneck.conv``, ``depth_conv.conv``, ``squeeze_excite.conv_reduce/conv_expand``,
                      ``point_linear.conv``) so that train_search.py:164-193's ``exec`` weight slicing and the
                      state_dict keys (:244-258) keep working.  Its arithmetic runs in the HIP library.
  ConvLayer / LinearLayer (models/layers.py:190-271, :322-428) -- parameter containers of the stems and the head; inside
                      Network they run as HIP cells (TFNAS_MODE_STEM / _HEAD), the classifier is an nn.Linear.
  Swish               (models/layers.py:26-35)
BatchNorm of the search net has no affine and no running statistics (layers.py:101-103,469,498,533): it is
a pure function of the batch and therefore has no module/state here.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .functions import CellPlan, MixedOpFn, BN_EPS


class Swish(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return x.mul_(x.sigmoid()) if self.inplace else x * x.sigmoid()


def get_same_padding(kernel_size):
    assert kernel_size % 2 > 0, 'kernel size should be odd number'
    return kernel_size // 2


class ConvLayer(nn.Module):
    """conv -> BN -> act of the stems / head: the parameter container (the arithmetic runs in Network's stem / head HIP cells)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, affine=False, act_func='relu'):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.act_func = kernel_size, stride, act_func
        self.affine = affine
        if affine:           # derived network (models/model_eval.py): BatchNorm2d(affine, running stats); registered before
            self.bn = nn.BatchNorm2d(out_channels, affine=True, track_running_stats=True)      # conv, like BasicLayer
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, get_same_padding(kernel_size),
                              bias=False)

    @property
    def name(self):
        return 'ConvLayer'

    def forward(self, x):
        # Inside Network (search and derived) the stems and the feature-mix head run as HIP cells (TFNAS_MODE_STEM / _HEAD:
        # model_search.Network._stem / _head, model_eval._DerivedBase); this module is their parameter container.  There is
        # deliberately NO stock-torch body here: a caller invoking the layer on its own would silently get MIOpen / rocBLAS
        # arithmetic instead of the HIP path.
        raise RuntimeError('tfnas_amd: ConvLayer runs only as part of Network (stem / head HIP cells: Network._stem, '
                           'Network._head); it has no standalone forward')


class LinearLayer(nn.Module):
    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.linear = nn.Linear(in_features, out_features, bias)

    @property
    def name(self):
        return 'LinearLayer'

    def forward(self, x):
        return self.linear(x)


def _seq(**mods):
    return nn.Sequential(OrderedDict(mods))


class MBInvertedResBlock(nn.Module):
    """MBConv block: 1x1 expand -> BN -> act -> depthwise kxk -> BN -> act -> [SE] -> 1x1 project -> BN [-> +x].

    With an expand convolution (every search candidate) ``forward`` runs the fused HIP path (sampled mode of
    ``tfnas_mixedop_fwd``); without one (only ``second_stem``: mid == in) it is a stem and uses torch ops."""

    def __init__(self, in_channels, mid_channels, se_channels, out_channels, kernel_size=3, stride=1,
                 affine=False, act_func='relu'):
        super().__init__()
        self.in_channels, self.mid_channels = in_channels, mid_channels
        self.se_channels, self.out_channels = se_channels, out_channels
        self.kernel_size, self.stride, self.act_func = kernel_size, stride, act_func
        self.affine = affine
        self.drop_connect_rate = 0.0
        def bn(ch):          # affine=True: the derived network's BatchNorm2d(affine, running statistics), layers.py:468,497,533
            return dict(bn=nn.BatchNorm2d(ch, affine=True, track_running_stats=True)) if affine else {}
        if mid_channels > in_channels:
            self.inverted_bottleneck = _seq(conv=nn.Conv2d(in_channels, mid_channels, 1, 1, 0, bias=False), **bn(mid_channels))
        else:
            self.inverted_bottleneck = None
            self.mid_channels = mid_channels = in_channels
        self.depth_conv = _seq(conv=nn.Conv2d(mid_channels, mid_channels, kernel_size, stride,
                                              get_same_padding(kernel_size), groups=mid_channels, bias=False),
                               **bn(mid_channels))
        if se_channels > 0:
            self.squeeze_excite = _seq(conv_reduce=nn.Conv2d(mid_channels, se_channels, 1, 1, 0, bias=True),
                                       conv_expand=nn.Conv2d(se_channels, mid_channels, 1, 1, 0, bias=True))
        else:
            self.squeeze_excite = None
            self.se_channels = 0
        self.point_linear = _seq(conv=nn.Conv2d(mid_channels, out_channels, 1, 1, 0, bias=False), **bn(out_channels))
        self.has_residual = (in_channels == out_channels) and (stride == 1)
        self._plan = None

    @property
    def name(self):
        return 'MBInvertedResBlock'

    def hip_params(self):
        """Weights in the order of TfnasGroup's pointer fields."""
        ps = [self.inverted_bottleneck.conv.weight, self.depth_conv.conv.weight, self.point_linear.conv.weight]
        if self.squeeze_excite is not None:
            se = self.squeeze_excite
            ps += [se.conv_reduce.weight, se.conv_reduce.bias, se.conv_expand.weight, se.conv_expand.bias]
        return ps

    def _stem_forward(self, x):
        # second_stem (mid == in: no expand convolution) runs fused with first_stem as ONE stem cell of the HIP library
        # (Network._stem); no stock-torch body on purpose (see ConvLayer.forward)
        raise RuntimeError('tfnas_amd: an MBInvertedResBlock without expand convolution (second_stem) runs only as part of '
                           'Network._stem; it has no standalone forward')

    def forward(self, x):
        if self.inverted_bottleneck is None:
            return self._stem_forward(x)
        if self._plan is None:
            self._plan = CellPlan(self.in_channels, self.out_channels, self.stride, self.act_func, [self])
        if self.affine:
            return self._affine_forward(x)
        return MixedOpFn.apply(self._plan, x, None, *self.hip_params())

    def bn_modules(self):
        return [self.inverted_bottleneck.bn, self.depth_conv.bn, self.point_linear.bn]

    def _affine_forward(self, x):
        """Derived-network block (layers.py:539-561 with affine BatchNorm; drop_connect of tools/utils.py:77-86 on the residual
        branch in training) -- tfnas_mbconv_fwd/bwd."""
        from .functions import MBConvAffineFn
        ds = None
        if self.training and self.has_residual and self.drop_connect_rate > 0.0:
            keep = 1.0 - self.drop_connect_rate
            u = getattr(self, 'drop_u', None)                   # (tests inject the uniform draws)
            u = torch.rand(x.size(0), dtype=x.dtype, device=x.device) if u is None else u.to(x.device)
            ds = torch.floor(keep + u) / keep
        bns = self.bn_modules()
        conv = self.hip_params()
        bnp = [t for m in bns for t in (m.weight, m.bias)]
        return MBConvAffineFn.apply(self._plan, x, ds, bns, self.training, len(conv), *conv, *bnp)
