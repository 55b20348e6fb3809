"""Architecture parsing / export (reference: parsing_model.py:20-134, models/model_eval.py:214-228,
tools/flops_benchmark.py, tools/utils.py:114-115).

``parse_searched_model`` turns a search checkpoint (or model) into the discrete architecture, its ``model.config`` JSON in
exactly the reference's format (so ``models/model_eval.NetworkCfg`` / train_eval.py consume it unchanged), the looked-up
latency, parameter count and multiply-accumulate count -- without instantiating the derived network."""
import json
from collections import OrderedDict

import torch

from . import geometry
from .epoch import get_op_and_depth_weights, parse_architecture          # noqa: F401  (re-exported: parsing_model.py API)
from .geometry import get_mc_num_dddict                                  # noqa: F401
from .latency import get_lookup_latency


def _conv_layer_config(ic, oc, k, stride, act):
    """models/layers.py ConvLayer.config (:251-261 + BasicLayer.config :169-177) for an affine conv-bn-act layer."""
    return {'name': 'ConvLayer', 'kernel_size': k, 'stride': stride, 'groups': 1, 'has_shuffle': False, 'bias': False,
            'in_channels': ic, 'out_channels': oc, 'use_bn': True, 'affine': True, 'act_func': act,
            'ops_order': 'weight_bn_act'}


def _mbconv_config(ic, mc, se, oc, k, stride, act):
    """models/layers.py MBInvertedResBlock.config (:581-596)."""
    return {'name': 'MBInvertedResBlock', 'in_channels': ic, 'mid_channels': mc, 'se_channels': se, 'out_channels': oc,
            'kernel_size': k, 'stride': stride, 'groups': 1, 'has_shuffle': False, 'bias': False, 'use_bn': True,
            'affine': True, 'act_func': act}


def derived_config(parsed_arch, mc_num_dddict, num_classes=1000):
    """``models/model_eval.Network(num_classes, parsed_arch, mc_num_dddict).config`` (model_eval.py:214-228)."""
    cfg = OrderedDict()
    cfg['first_stem'] = _conv_layer_config(3, 32, 3, 2, 'relu')
    cfg['second_stem'] = _mbconv_config(32, 32, 8, 16, 3, 1, 'relu')
    for stage, st in geometry.STAGES.items():
        blocks = []
        for b, (ic, oc, s) in enumerate(zip(st['ics'], st['ocs'], st['ss']), start=1):
            name = 'block%d' % b
            if name not in parsed_arch[stage]:
                continue
            op = parsed_arch[stage][name]
            blocks.append(_mbconv_config(ic, mc_num_dddict[stage][name][op], geometry.se_channels(ic, op), oc,
                                         geometry.OP_KERNEL[op], s, st['act']))
        cfg[stage] = blocks
    cfg['feature_mix_layer'] = _conv_layer_config(320, 1280, 1, 1, 'swish')
    cfg['classifier'] = {'name': 'LinearLayer', 'in_features': 1280, 'out_features': num_classes, 'bias': True,
                         'use_bn': False, 'affine': False, 'act_func': None, 'ops_order': 'weight_bn_act'}
    return dict(cfg)


def _layers(config):
    yield config['first_stem']
    yield config['second_stem']
    for k in sorted(k for k in config if k.startswith('stage')):
        for blk in config[k]:
            yield blk
    yield config['feature_mix_layer']


def count_macs_in_M(config, input_size=224):
    """What tools/flops_benchmark.calculate_FLOPs_in_M reports for the derived network (per image, in millions): conv /
    linear multiply-accumulates incl. bias adds, and the network's global average pool (SE pooling is functional in the
    reference, layers.py:549, hence not hooked and not counted)."""
    size = input_size
    total = 0.0
    for c in _layers(config):
        if c['name'] == 'ConvLayer':
            size = (size - 1) // c['stride'] + 1
            total += c['kernel_size'] ** 2 * c['in_channels'] * c['out_channels'] * size * size
        else:
            ic, mc, se, oc, k, s = (c[n] for n in ('in_channels', 'mid_channels', 'se_channels', 'out_channels',
                                                    'kernel_size', 'stride'))
            if mc > ic:                                        # expand conv only when mid > in (layers.py:462)
                total += ic * mc * size * size
            else:
                mc = ic
            size = (size - 1) // s + 1
            total += k * k * mc * size * size                  # depthwise
            if se > 0:
                total += (mc * se + se) + (se * mc + mc)        # 1x1 convs on the pooled vector, with bias
            total += mc * oc * size * size
    last = config['feature_mix_layer']['out_channels']
    total += last * size * size                                # AdaptiveAvgPool2d(1): one op per input element
    cl = config['classifier']
    total += cl['in_features'] * cl['out_features'] + (cl['out_features'] if cl['bias'] else 0)
    return total / 1e6


def count_params_in_MB(config):
    """tools/utils.count_parameters_in_MB of the derived network (every parameter incl. affine BN, /1e6)."""
    n = 0
    for c in _layers(config):
        if c['name'] == 'ConvLayer':
            n += c['kernel_size'] ** 2 * c['in_channels'] * c['out_channels'] + 2 * c['out_channels']
        else:
            ic, mc, se, oc, k = (c[x] for x in ('in_channels', 'mid_channels', 'se_channels', 'out_channels', 'kernel_size'))
            if mc > ic:
                n += ic * mc + 2 * mc
            else:
                mc = ic
            n += k * k * mc + 2 * mc
            if se > 0:
                n += mc * se + se + se * mc + mc
            n += mc * oc + 2 * oc
    cl = config['classifier']
    n += cl['in_features'] * cl['out_features'] + (cl['out_features'] if cl['bias'] else 0)
    return n / 1e6


def parse_searched_model(model_or_path, lat_lookup=None, num_classes=1000, save_path=None):
    """parsing_model.py's __main__ (:91-134) as a function: checkpoint / model -> dict(parsed_arch, mc_num_dddict, config,
    lat_lut, params_MB, macs_M); optionally writes the config JSON like ``--save_path``."""
    if isinstance(model_or_path, str):
        ck = torch.load(model_or_path, map_location='cpu', weights_only=False)
        sd, masks = ck['state_dict'], ck['mc_mask_dddict']
        mc_num = get_mc_num_dddict(masks)
    else:
        net = getattr(model_or_path, 'module', model_or_path)
        sd, mc_num = net.state_dict(), net.mc_num_dddict
    parsed = parse_architecture(*get_op_and_depth_weights(sd))
    config = derived_config(parsed, mc_num, num_classes)
    out = dict(parsed_arch=parsed, mc_num_dddict=mc_num, config=config, params_MB=count_params_in_MB(config),
               macs_M=count_macs_in_M(config))
    if lat_lookup is not None:
        out['lat_lut'] = get_lookup_latency(parsed, mc_num, geometry.make_lat_lookup_key_dddict(), lat_lookup)
    if save_path:
        with open(save_path, 'w') as f:
            json.dump(config, f, indent=4)
    return out
