"""Epoch boundary of the TF-NAS search: everything ``train_search.py`` does between two calls of the inner loop.

SURVEY.md 8(f) row 1.  Host logic (index_select / scatter on the weight store, L1-norm channel ranking, the latency-driven
width search of ``elasticity.py``), written so that the reference's data survives unchanged:

  * the *max-width store* is a plain ``state_dict`` with ``nn.DataParallel``'s ``module.`` key prefix
    (train_search.py:100-103, 163, 235) -- a checkpoint written by the reference loads here and vice versa;
  * ``mc_mask_dddict`` is the reference's nested OrderedDict of 0/1 float masks (tools/config.py:4-197);
  * checkpoint files are ``{'state_dict', 'mc_mask_dddict'}`` named ``searched_model_XX.pth.tar`` (train_search.py:99-103,
    310-315).

Functions (reference lines they restate):
  slice_weights_from_max    train_search.py:161-194   current-width model <- rows/columns of the store picked by the masks
  scatter_weights_to_max    train_search.py:234-259   store <- trained current-width weights
  get_op_and_depth_weights  parsing_model.py:20-41
  parse_architecture        parsing_model.py:44-74    argmax op per cell, argmax depth per stage
  shrink_or_expand          train_search.py:261-291   elasticity scaling towards the latency target
  remask_by_l1              train_search.py:293-305   keep the channels whose depthwise filters have the largest L1 norm
  save/load_search_checkpoint  train_search.py:99-103, 162-163, 310-315
  cosine_lr_list            train_search.py:104-117
  search_epoch / run_search train_search.py:155-315   one epoch / the whole schedule on the HIP model

The kernels are re-specialised for the new (ragged) widths simply by building the next epoch's Network from the new
``mc_num_dddict`` -- descriptors are planned per launch (tfnas_cell_plan), nothing is compiled per width.
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import geometry
from .elasticity import fit_mc_num_by_latency
from .latency import get_lookup_latency

# (tensor name inside an op, dimension the mid-channel mask applies to; None: copied whole)   train_search.py:170-193
_OP_TENSORS = (
    ('inverted_bottleneck.conv.weight', 0),
    ('depth_conv.conv.weight', 0),
    ('point_linear.conv.weight', 1),
)
_SE_TENSORS = (
    ('squeeze_excite.conv_reduce.weight', 1),
    ('squeeze_excite.conv_reduce.bias', None),
    ('squeeze_excite.conv_expand.weight', 0),
    ('squeeze_excite.conv_expand.bias', 0),
)


def _op_tensors(op_idx):
    return _OP_TENSORS + (_SE_TENSORS if op_idx >= 4 else ())


def _mask_index(mask, device):
    return torch.nonzero(mask).view(-1).to(device)


def _resolve(module, dotted):
    obj = module
    for part in dotted.split('.'):
        obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
    return obj


def slice_weights_from_max(model, state_dict, mc_mask_dddict, prefix='module.'):
    """Load a current-width ``model`` (built from ``get_mc_num_dddict(mc_mask_dddict)``) from the max-width store:
    tensors outside ``m_ops`` are taken as they are, candidate weights are ``index_select``-ed with the channel masks.
    ``model`` is the bare Network (what the reference reaches as ``model.module``)."""
    net = getattr(model, 'module', model)
    dev = next(net.parameters()).device
    for key, val in state_dict.items():
        if 'm_ops' not in key:
            name = key[len(prefix):] if key.startswith(prefix) else key
            _resolve(net, name).data = val.data.to(dev)
    for stage, blocks in mc_mask_dddict.items():
        for block, ops in blocks.items():
            for op_idx, mask in ops.items():
                op = getattr(getattr(net, stage), block).m_ops[op_idx]
                for name, dim in _op_tensors(op_idx):
                    src = state_dict['{}{}.{}.m_ops.{}.{}'.format(prefix, stage, block, op_idx, name)]
                    index = _mask_index(mask, src.device)
                    val = src.data if dim is None else torch.index_select(src, dim, index).data
                    _resolve(op, name).data = val.to(dev)
    return model


def scatter_weights_to_max(state_dict, model, mc_mask_dddict, prefix='module.'):
    """Write the trained current-width weights back into the rows / columns of the max-width store they came from."""
    net = getattr(model, 'module', model)
    cur = {prefix + k: v for k, v in net.state_dict().items()}
    for key in state_dict:
        if 'm_ops' not in key:
            # clone: `cur` holds views into the model's flat WeightArena buffer; keeping a view would alias the store to the
            # (dead) epoch model's arena and torch.save would serialise the whole arena storage per tensor
            state_dict[key].data = cur[key].data.detach().clone().to(state_dict[key].device)
    for stage, blocks in mc_mask_dddict.items():
        for block, ops in blocks.items():
            for op_idx, mask in ops.items():
                for name, dim in _op_tensors(op_idx):
                    key = '{}{}.{}.m_ops.{}.{}'.format(prefix, stage, block, op_idx, name)
                    dst = state_dict[key].data
                    src = cur[key].data.to(dst.device)
                    index = _mask_index(mask, dst.device)
                    if dim is None:
                        dst[:] = src
                    elif dim == 0:
                        dst[index] = src
                    else:
                        dst[:, index] = src
    return state_dict


def get_op_and_depth_weights(model_or_state_dict):
    """exp(log_alphas) per cell and softmax(betas) per stage, in state_dict order (parsing_model.py:20-41); accepts a model, a
    state_dict or a checkpoint path."""
    if isinstance(model_or_state_dict, str):
        sd = torch.load(model_or_state_dict, map_location='cpu')['state_dict']
    elif hasattr(model_or_state_dict, 'state_dict'):
        sd = model_or_state_dict.state_dict()
    else:
        sd = model_or_state_dict
    op_weights, depth_weights = [], []
    for key, val in sd.items():
        if key.endswith('log_alphas'):
            op_weights.append(np.exp(val.detach().cpu().numpy()))
        elif key.endswith('betas'):
            depth_weights.append(F.softmax(val.detach().cpu(), dim=-1).numpy())
    return op_weights, depth_weights


def parse_architecture(op_weights, depth_weights):
    """Discrete architecture: the strongest candidate of every cell, cut at the strongest depth of every stage
    (parsing_model.py:44-74).  Returns OrderedDict[stage][block] -> op index."""
    parsed = OrderedDict((stage, OrderedDict(('block%d' % b, -1) for b in range(1, len(cfg['ics']) + 1)))
                         for stage, cfg in geometry.STAGES.items())
    cells = [(st, blk) for st in parsed for blk in parsed[st]]
    for (st, blk), w in zip(cells, op_weights):
        parsed[st][blk] = int(np.argmax(w))
    for stage_index, w in enumerate(depth_weights, start=1):
        depth = int(np.argmax(w)) + 1
        stage = 'stage%d' % stage_index
        for b in range(depth + 1, 6):
            parsed[stage].pop('block%d' % b, None)
    return parsed


def shrink_or_expand(parsed_arch, mc_mask_dddict, mc_maxnum_dddict, lat_lookup_key_dddict, lat_lookup, target_lat):
    """Elasticity scaling of the chosen candidates' widths towards ``target_lat`` (train_search.py:261-291): one pass over
    all stages in the needed direction, then expanding passes over stages 2..6, 3..6, ..., 6.
    Returns (mc_num_dddict, before_lat, after_lat)."""
    mc_num = geometry.get_mc_num_dddict(mc_mask_dddict)
    before = get_lookup_latency(parsed_arch, mc_num, lat_lookup_key_dddict, lat_lookup)
    after = before
    if before != target_lat:
        stages = ['stage%d' % x for x in range(1, 7)]
        mc_num, after = fit_mc_num_by_latency(parsed_arch, mc_num, mc_maxnum_dddict, lat_lookup_key_dddict, lat_lookup,
                                              target_lat, stages, sign=-1 if before > target_lat else 1)
        for start in range(2, 7):
            stages = ['stage%d' % x for x in range(start, 7)]
            mc_num, after = fit_mc_num_by_latency(parsed_arch, mc_num, mc_maxnum_dddict, lat_lookup_key_dddict, lat_lookup,
                                                  target_lat, stages, sign=1)
    return mc_num, before, after


def remask_by_l1(parsed_arch, mc_num_dddict, mc_mask_dddict, state_dict, prefix='module.'):
    """Where elasticity scaling changed a chosen candidate's width, re-select its active channels: the ``mc_num`` output
    channels of the max-width depthwise filter bank with the largest L1 norm (train_search.py:293-305).  In place."""
    changed = []
    for stage, blocks in parsed_arch.items():
        for block, op_idx in blocks.items():
            mask = mc_mask_dddict[stage][block][op_idx]
            mc_num = mc_num_dddict[stage][block][op_idx]
            if mc_num == int(mask.sum().item()):
                continue
            key = '{}{}.{}.m_ops.{}.depth_conv.conv.weight'.format(prefix, stage, block, op_idx)
            l1 = np.sum(state_dict[key].detach().clone().abs().cpu().numpy(), axis=(1, 2, 3))
            keep = np.argsort(l1)[::-1][:mc_num]
            mask.data[:] = 0.0
            mask.data[keep.tolist()] = 1.0
            changed.append((stage, block, op_idx))
    return changed


def save_search_checkpoint(save_dir, epoch, state_dict, mc_mask_dddict):
    path = os.path.join(save_dir, 'searched_model_{:02}.pth.tar'.format(epoch))
    torch.save({'state_dict': state_dict, 'mc_mask_dddict': mc_mask_dddict}, path)
    return path


def load_search_checkpoint(save_dir, epoch, map_location='cpu'):
    ck = torch.load(os.path.join(save_dir, 'searched_model_{:02}.pth.tar'.format(epoch)), map_location=map_location,
                    weights_only=False)
    return ck['state_dict'], ck['mc_mask_dddict']


def cosine_lr_list(w_lr, epochs):
    """Per-epoch learning rates of CosineAnnealingLR(optimizer_w, epochs) read before each scheduler.step()
    (train_search.py:104-117): lr_e = w_lr * (1 + cos(pi * e / epochs)) / 2."""
    return [0.5 * w_lr * (1.0 + math.cos(math.pi * e / float(epochs))) for e in range(epochs)]


def search_epoch(epoch, state_dict, mc_mask_dddict, lat_lookup, train_queue, val_queue, *, num_classes=100, epochs=100,
                 lr=0.025, T=5.0, w_mom=0.9, w_wd=1e-5, a_lr=0.01, a_wd=5e-4, a_betas=(0.5, 0.999), grad_clip=5.0,
                 target_lat=15.0, lambda_lat=0.1, warmup_epochs=10, noise=None, device='cuda', group=None, log=None):
    """One epoch of the reference's main loop (train_search.py:155-307) on the HIP model.  ``state_dict`` / ``mc_mask_dddict``
    are updated in place and returned together with the epoch's statistics.  ``train_queue`` / ``val_queue``: iterables of
    (x, target) batches (the weight-sharing / arch-step queues of train_w_arch)."""
    from . import search
    from .model_search import Network
    log = log or (lambda *a: None)
    lat_keys = geometry.make_lat_lookup_key_dddict()
    mc_max = geometry.get_mc_num_dddict(mc_mask_dddict, is_max=True)
    mc_num = geometry.get_mc_num_dddict(mc_mask_dddict)
    model = Network(num_classes, mc_num, lat_lookup).to(device)
    model.set_temperature(T)
    slice_weights_from_max(model, state_dict, mc_mask_dddict)
    state = search.SearchState(model)
    opt_w, opt_a = search.make_optimizers(model, lr, w_mom, w_wd, a_lr, a_wd, a_betas)
    noise = noise or search.NoiseSource(1000 + epoch)
    dev = torch.device(device)
    stats = dict(epoch=epoch, lr=lr, T=T, steps=0)
    val_iter = iter(val_queue) if val_queue is not None else None
    try:
        for step, (x_w, t_w) in enumerate(train_queue):
            x_w, t_w = x_w.to(dev, non_blocking=True), t_w.to(dev, non_blocking=True)
            if epoch < warmup_epochs:                              # train_wo_arch (train_search.py:318-354)
                search.w_step(state, x_w, t_w, opt_w, grad_clip, noise.exp(dev), bi_sampling=False, group=group)
            else:                                                  # train_w_arch (train_search.py:357-432)
                search.w_step(state, x_w, t_w, opt_w, grad_clip, noise.exp(dev), noise.rand_pos(), group=group)
                if step % 2 == 0:
                    try:
                        x_a, t_a = next(val_iter)
                    except StopIteration:
                        val_iter = iter(val_queue)
                        x_a, t_a = next(val_iter)
                    la, ll, lat, _ = search.a_step(state, x_a.to(dev, non_blocking=True), t_a.to(dev, non_blocking=True), opt_a,
                                                   target_lat, lambda_lat, grad_clip, noise.exp(dev), group=group)
                    stats['last_lat'] = lat
            stats['steps'] = step + 1
        if 'last_lat' in stats:
            stats['last_lat'] = float(stats['last_lat'])
        if epochs - epoch < 5 and val_queue is not None:           # validation for the last 5 epochs (train_search.py:229-231)
            stats['val_top1'], stats['val_top5'], stats['val_loss'] = search.validate(state, val_queue, noise=noise)
        torch.cuda.synchronize(dev) if dev.type == 'cuda' else None
    except BaseException:
        # a failed step must not leak the path contexts (streams + ~130 events each) or leave the data-parallel segment hook
        # armed with this epoch's state
        if state.runner is not None:
            state.runner.segment_hook = None
            state.runner.close()
        raise
    scatter_weights_to_max(state_dict, model, mc_mask_dddict)
    if epoch >= warmup_epochs:
        op_w, depth_w = get_op_and_depth_weights(model)
        parsed = parse_architecture(op_w, depth_w)
        mc_new, before, after = shrink_or_expand(parsed, mc_mask_dddict, mc_max, lat_keys, lat_lookup, target_lat)
        changed = remask_by_l1(parsed, mc_new, mc_mask_dddict, state_dict)
        stats.update(parsed_arch=parsed, before_lat=before, after_lat=after, remasked=changed)
        log('epoch %d: lat %.4f -> %.4f (target %.4f), %d candidates re-masked' % (epoch, before, after, target_lat,
                                                                                  len(changed)))
    if state.runner is not None:
        state.runner.segment_hook = None
        state.runner.close()
    return state_dict, mc_mask_dddict, stats


def run_search(save_dir, lat_lookup, make_train_queue, make_val_queue, *, num_classes=100, epochs=100, w_lr=0.025, T=5.0,
               T_decay=0.96, warmup_epochs=10, seed=2, start_epoch=0, device='cuda', log=print, **kw):
    """The whole schedule of train_search.py:84-315: initial max-width checkpoint, per-epoch model at the current widths,
    cosine learning-rate list, temperature decay after every arch epoch, per-epoch checkpoints ``searched_model_XX.pth.tar``."""
    from .model_search import Network
    from .search import TfnasDataParallel
    os.makedirs(save_dir, exist_ok=True)
    torch.manual_seed(seed)
    if start_epoch == 0:
        mc_mask = geometry.make_mc_mask_dddict()
        full = TfnasDataParallel(Network(num_classes, geometry.get_mc_num_dddict(mc_mask, is_max=True), lat_lookup),
                                 device=torch.device(device))
        save_search_checkpoint(save_dir, 0, full.state_dict(), mc_mask)
        del full
    lrs = cosine_lr_list(w_lr, epochs)
    for e in range(start_epoch):
        if e >= warmup_epochs:
            T *= T_decay
    history = []
    for epoch in range(start_epoch, epochs):
        state_dict, mc_mask = load_search_checkpoint(save_dir, epoch, map_location=device)
        state_dict, mc_mask, stats = search_epoch(epoch, state_dict, mc_mask, lat_lookup, make_train_queue(epoch),
                                                  make_val_queue(epoch), num_classes=num_classes, epochs=epochs,
                                                  lr=lrs[epoch], T=T, warmup_epochs=warmup_epochs, device=device, log=log, **kw)
        if epoch >= warmup_epochs:
            T *= T_decay
        save_search_checkpoint(save_dir, epoch + 1, state_dict, mc_mask)
        history.append(stats)
    return history
