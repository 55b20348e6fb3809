"""Derived ("searched") network for the retrain path -- drop-in for the reference's ``models/model_eval.py``
(BASELINE configs[4]).

``Network(num_classes, parsed_arch, mc_num_dddict, lat_lookup, dropout_rate, drop_connect_rate)`` (model_eval.py:31-244) and
``NetworkCfg(num_classes, model_config, ...)`` (:247-430) with the reference's module / parameter / buffer names (``state_dict``
keys identical, incl. BatchNorm running statistics), ``forward(x) -> logits``, ``get_lookup_latency(x)`` and ``config``.  Every
block -- stems, MBConv blocks with AFFINE BatchNorm, running statistics in train / eval mode and drop-connect, the feature-mix
head -- runs on the HIP kernels of the search path (``tfnas_mbconv_fwd/bwd``, ``tfnas_head_affine_fwd/bwd``: the affine part is
folded into the per-channel statistics tables, csrc/bn_affine.hip); dropout, the classifier GEMM and the loss are torch ops.
``CrossEntropyLabelSmooth`` and ``train_step`` / ``validate`` restate train_eval.py:72-85, 228-293.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, geometry
from .functions import CellPlan, HeadAffineFn, MBConvAffineFn
from .layers import ConvLayer, LinearLayer, MBInvertedResBlock
from .parsing import derived_config


class _StemBlock:
    """first_stem.conv + second_stem presented as one candidate block of a TFNAS_MODE_STEM cell."""

    def __init__(self, first, second):
        self.first, self.second = first, second
        self.mid_channels, self.kernel_size, self.se_channels = second.mid_channels, second.kernel_size, second.se_channels

    def hip_params(self):
        se = self.second.squeeze_excite
        return [self.first.conv.weight, self.second.depth_conv.conv.weight, self.second.point_linear.conv.weight,
                se.conv_reduce.weight, se.conv_reduce.bias, se.conv_expand.weight, se.conv_expand.bias]


class _HeadBlock:
    def __init__(self, layer):
        self.layer = layer
        self.mid_channels, self.kernel_size, self.se_channels = layer.out_channels, 3, 0

    def hip_params(self):
        return [self.layer.conv.weight]


class _DerivedBase(nn.Module):
    def _finish(self, num_classes):
        self.feature_mix_layer = ConvLayer(320, 1280, kernel_size=1, stride=1, affine=True, act_func='swish')
        self.global_avg_pooling = nn.AdaptiveAvgPool2d(1)
        self.classifier = LinearLayer(1280, num_classes)
        self._initialization()
        self._stem_plan = self._head_plan = None
        from .functions import adopt_modes
        adopt_modes(self)                      # self.hip_modes (functions.HipModes), shared by every sub-module's plans

    def set_hip_modes(self, **kw):
        """e.g. set_hip_modes(gemm='bf16'): this model's launches only (TfnasCellDesc.gemm_mode; functions.HipModes)."""
        for k, v in kw.items():
            if not hasattr(self.hip_modes, k):
                raise AttributeError(k)
            setattr(self.hip_modes, k, v)

    def _stages(self):
        return [getattr(self, 'stage%d' % i) for i in range(1, 7)]

    def _stem(self, x):
        fs, ss = self.first_stem, self.second_stem
        if self._stem_plan is None:
            self._stem_plan = CellPlan(27, ss.out_channels, 1, 'relu', [_StemBlock(fs, ss)], mode=_lib.MODE_STEM, modes=self.hip_modes)
        plan = self._stem_plan
        bns = [fs.bn, ss.depth_conv.bn, ss.point_linear.bn]
        conv = plan.params()
        bnp = [t for m in bns for t in (m.weight, m.bias)]
        return MBConvAffineFn.apply(plan, x, None, bns, self.training, len(conv), *conv, *bnp)

    def _head(self, x):
        fm = self.feature_mix_layer
        if self._head_plan is None:
            self._head_plan = CellPlan(fm.in_channels, 4, 1, fm.act_func, [_HeadBlock(fm)], mode=_lib.MODE_HEAD, modes=self.hip_modes)
        return HeadAffineFn.apply(self._head_plan, x, fm.bn, self.training, fm.conv.weight, fm.bn.weight, fm.bn.bias)

    def forward(self, x):
        x = self._stem(x)
        for stage in self._stages():
            for block in stage:
                x = block(x)
        x = self._head(x)                                   # feature_mix_layer + global_avg_pooling, [N, 1280]
        if self.dropout_rate > 0.0:
            x = F.dropout(x, p=self.dropout_rate, training=self.training)
        return self.classifier(x)

    def get_lookup_latency(self, x):
        """Sum of the looked-up block latencies (model_eval.py:133-212); ``x`` only provides the input resolution."""
        if not self.lat_lookup:
            return 0.0
        lat = self.lat_lookup['base']
        size = (x.size(-1) - 1) // 2 + 1                    # after first_stem (stride 2); second_stem keeps it
        for stage in self._stages():
            for b in stage:
                key = '{}_{}_{}_{}_{}_k{}_s{}_{}'.format(b.name, size, b.in_channels, b.se_channels, b.out_channels,
                                                         b.kernel_size, b.stride, b.act_func)
                lat += self.lat_lookup[key][b.mid_channels]
                size = (size - 1) // b.stride + 1
        return lat

    @staticmethod
    def _block_config(b):
        return {'name': 'MBInvertedResBlock', 'in_channels': b.in_channels, 'mid_channels': b.mid_channels,
                'se_channels': b.se_channels, 'out_channels': b.out_channels, 'kernel_size': b.kernel_size, 'stride': b.stride,
                'groups': 1, 'has_shuffle': False, 'bias': False, 'use_bn': True, 'affine': True, 'act_func': b.act_func}

    @property
    def config(self):
        from .parsing import _conv_layer_config
        cfg = {'first_stem': _conv_layer_config(3, 32, 3, 2, 'relu'), 'second_stem': self._block_config(self.second_stem)}
        for i, stage in enumerate(self._stages(), start=1):
            cfg['stage%d' % i] = [self._block_config(b) for b in stage]
        cfg['feature_mix_layer'] = _conv_layer_config(320, 1280, 1, 1, 'swish')
        cl = self.classifier
        cfg['classifier'] = {'name': 'LinearLayer', 'in_features': cl.in_features, 'out_features': cl.out_features,
                             'bias': True, 'use_bn': False, 'affine': False, 'act_func': None, 'ops_order': 'weight_bn_act'}
        return cfg

    def _initialization(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)) and m.bias is not None:
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)


class Network(_DerivedBase):
    def __init__(self, num_classes, parsed_arch, mc_num_dddict, lat_lookup=None, dropout_rate=0.0, drop_connect_rate=0.0):
        super().__init__()
        self.lat_lookup, self.mc_num_dddict, self.parsed_arch = lat_lookup, mc_num_dddict, parsed_arch
        self.dropout_rate, self.drop_connect_rate = dropout_rate, drop_connect_rate
        self.block_count = 1 + sum(len(parsed_arch[st]) for st in parsed_arch)
        self.block_idx = 0
        self.first_stem = ConvLayer(3, 32, kernel_size=3, stride=2, affine=True, act_func='relu')
        self.second_stem = MBInvertedResBlock(32, 32, 8, 16, kernel_size=3, stride=1, affine=True, act_func='relu')
        self.block_idx += 1
        self.second_stem.drop_connect_rate = self.drop_connect_rate * self.block_idx / self.block_count
        for name, cfg in geometry.STAGES.items():
            stage = nn.ModuleList()
            for i, block_name in enumerate(parsed_arch[name]):
                self.block_idx += 1
                op = parsed_arch[name][block_name]
                ic, oc, s = cfg['ics'][i], cfg['ocs'][i], cfg['ss'][i]
                blk = MBInvertedResBlock(ic, mc_num_dddict[name][block_name][op], geometry.se_channels(ic, op), oc,
                                         geometry.OP_KERNEL[op], s, affine=True, act_func=cfg['act'])
                blk.drop_connect_rate = self.drop_connect_rate * self.block_idx / self.block_count
                stage.append(blk)
            setattr(self, name, stage)
        self._finish(num_classes)


class NetworkCfg(_DerivedBase):
    """Built from the exported ``model.config`` JSON (model_eval.py:247-300; parsing.derived_config writes it)."""

    def __init__(self, num_classes, model_config, lat_lookup=None, dropout_rate=0.0, drop_connect_rate=0.0):
        super().__init__()
        self.lat_lookup, self.model_config = lat_lookup, model_config
        self.dropout_rate, self.drop_connect_rate = dropout_rate, drop_connect_rate
        self.block_count = 1 + sum(len(v) for k, v in model_config.items() if k.startswith('stage'))
        self.block_idx = 0

        def mb(c):
            if c['name'] != 'MBInvertedResBlock' or c.get('groups', 1) != 1 or c.get('has_shuffle') or c.get('bias'):
                raise NotImplementedError('only plain MBInvertedResBlock configs are supported: %r' % (c,))
            return MBInvertedResBlock(c['in_channels'], c['mid_channels'], c['se_channels'], c['out_channels'],
                                      c['kernel_size'], c['stride'], affine=c.get('affine', True), act_func=c['act_func'])
        fs = model_config['first_stem']
        self.first_stem = ConvLayer(fs['in_channels'], fs['out_channels'], fs['kernel_size'], fs['stride'], affine=True,
                                    act_func=fs['act_func'])
        self.second_stem = mb(model_config['second_stem'])
        self.block_idx += 1
        self.second_stem.drop_connect_rate = self.drop_connect_rate * self.block_idx / self.block_count
        for i in range(1, 7):
            stage = nn.ModuleList()
            for c in model_config['stage%d' % i]:
                self.block_idx += 1
                blk = mb(c)
                blk.drop_connect_rate = self.drop_connect_rate * self.block_idx / self.block_count
                stage.append(blk)
            setattr(self, 'stage%d' % i, stage)
        self._finish(num_classes)


class CrossEntropyLabelSmooth(nn.Module):
    """train_eval.py:72-85: mean over the batch of -sum_k ((1-eps) onehot + eps/K) log_softmax."""

    def __init__(self, num_classes, epsilon):
        super().__init__()
        self.num_classes, self.epsilon = num_classes, epsilon

    def forward(self, xs, targets):
        return F.cross_entropy(xs, targets, label_smoothing=self.epsilon)


# TFNAS_RETRAIN_DIRECT=0: the blocks return gradient temporaries and autograd adds them into the arena (one launch per parameter);
# TFNAS_RETRAIN_LAZY=0: every block joins its weight-gradient kernels before it returns.
DIRECT_GRADS = True          # (tests flip these two module flags to compare with the plain route)
LAZY_JOIN = True


class RetrainState:
    """Flat weight / gradient / momentum arenas of the derived network (path.WeightArena) + the fused step tail of the search
    path: ``.grad`` of every parameter is a view into ONE gradient buffer (autograd accumulates in place), so a data-parallel
    step is ONE all-reduce of that buffer (no ``torch.cat``, no copy back -- what apex DDP's ``delay_allreduce`` flattening does,
    train_eval_amp.py:188) and clip_grad_norm_ + SGD(momentum, weight decay) are the two launches of ``tfnas_sgd_clip_step``
    (1 / world folded in), sharing the momentum buffers with the caller's ``torch.optim.SGD`` (checkpoints keep working)."""

    def __init__(self, model):
        from .path import WeightArena
        self.model = model
        self.arena = WeightArena(model)
        self._bound = None
        self._scratch = self._gnorm = None

    def intact(self):
        return self.arena.intact()

    @staticmethod
    def fusable(opt, model=None):
        """The fused tail updates EVERY parameter of the arena with one set of hyper-parameters, so it stands in for
        ``optimizer.step()`` only when that is what the optimizer would do: a plain SGD with ONE param group that holds exactly
        the model's parameters, all of them trainable, and no step hooks registered (they would never run).  Anything else --
        frozen parameters, a subset, per-group rates, hooks -- takes torch's clip_grad_norm_ + optimizer.step()."""
        g = opt.param_groups
        if not (isinstance(opt, torch.optim.SGD) and len(g) == 1 and not g[0].get('nesterov') and not g[0].get('dampening')
                and not g[0].get('maximize')):
            return False
        if getattr(opt, '_optimizer_step_pre_hooks', None) or getattr(opt, '_optimizer_step_post_hooks', None):
            return False
        if model is not None:
            mine = list(model.parameters())
            if not all(p.requires_grad for p in mine) or {id(p) for p in g[0]['params']} != {id(p) for p in mine}:
                return False
        return True

    def _bind_momentum(self, opt):
        a = self.arena
        probe = a.params[0]
        buf = opt.state.get(probe, {}).get('momentum_buffer')
        if self._bound is opt and buf is not None and buf.data_ptr() == a.m.data_ptr() + 4 * a.slot[id(probe)][0]:
            return
        with torch.no_grad():
            for p in a.params:
                o, n = a.slot[id(p)]
                view = a.m[o:o + n].view(p.shape)
                old = opt.state.get(p, {}).get('momentum_buffer')
                if old is not None and old.data_ptr() != view.data_ptr():
                    view.copy_(old)
                elif old is None:
                    view.zero_()
                opt.state[p]['momentum_buffer'] = view
        self._bound = opt

    def begin(self):
        """Zero the gradient arena (one memset), make every .grad a view of it, and switch the blocks' backward to writing the
        gradients in place with a lazily joined weight-gradient stream (functions.retrain_context)."""
        from . import functions
        a = self.arena
        a.g.zero_()
        for p in a.params:
            if p.grad is None or p.grad.data_ptr() != a.grad_ptr(p):
                p.grad = a.grad_view(p)
        functions.retrain_context(DIRECT_GRADS, LAZY_JOIN, self._modes_owner())

    def _modes_owner(self):
        """The module whose ``hip_modes`` the blocks of this model consult: the model itself, or -- when ``self.model`` is a
        wrapper without one (nn.DataParallel / DDP / nn.Sequential / a user wrapper) -- the first sub-module that has it."""
        m = self.model
        if getattr(m, 'hip_modes', None) is not None:
            return m
        for sub in m.modules():
            if getattr(sub, 'hip_modes', None) is not None:
                return sub
        return None                                      # (no HIP module inside: functions.retrain_context falls back to DEFAULT_MODES)

    def end(self):
        """Join the weight-gradient stream and leave the training-step context (also on an error path)."""
        from . import functions
        owner = self._modes_owner()
        modes = owner.hip_modes if owner is not None else functions.DEFAULT_MODES
        if modes.lazy_join:
            functions.retrain_join(self.arena.device)
        functions.retrain_context(False, False, owner)

    def step(self, opt, grad_clip, group=None):
        import ctypes as C
        import torch.distributed as dist
        a = self.arena
        self.end()                                       # the gradients are complete on the current stream from here on
        self._bind_momentum(opt)
        hp = opt.param_groups[0]
        scale = 1.0
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(a.g, group=group)            # ONE message: the whole gradient arena
            scale = 1.0 / dist.get_world_size(group)
        dev = a.device
        nblk = (a.total + 8191) // 8192
        if self._scratch is None or self._scratch.numel() < nblk + 8:
            self._scratch = torch.empty(max(4096, 2 * nblk), device=dev, dtype=torch.float64)
            self._gnorm = torch.zeros(2, device=dev, dtype=torch.float32)
        off, ln = (C.c_uint64 * 1)(0), (C.c_uint64 * 1)(a.total)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().tfnas_sgd_clip_step(_lib.ptr(a.w), _lib.ptr(a.g), _lib.ptr(a.m), 1, off, None, ln, float(grad_clip),
                                                  float(hp['lr']), float(hp['momentum']), float(hp['weight_decay']), float(scale),
                                                  _lib.ptr(self._scratch), self._scratch.numel(), _lib.ptr(self._gnorm), stream),
                   'tfnas_sgd_clip_step')
        # what optimizer.step() would have recorded: learning-rate schedulers check these before their own step()
        if hasattr(opt, '_step_count'):
            opt._step_count += 1
        opt._opt_called = True


def train_step(model, x, target, criterion, optimizer, grad_clip=5.0, group=None, fused=True):
    """One iteration of train_eval.py:228-252 (forward, label-smoothed loss, backward, clip, SGD); with a process group the
    gradients are averaged with ONE all-reduce of the flat gradient arena before clipping (one process per GPU instead of apex
    DDP).  ``fused`` (GPU models with a plain torch.optim.SGD): RetrainState -- gradient arena + tfnas_sgd_clip_step; otherwise
    torch's clip_grad_norm_ + optimizer.step()."""
    import torch.distributed as dist
    model.train()
    st = None
    if fused and x.is_cuda and RetrainState.fusable(optimizer, model):
        st = getattr(model, '_retrain_state', None)
        if st is None or not st.intact():
            st = RetrainState(model)
            object.__setattr__(model, '_retrain_state', st)         # (not a submodule / buffer: stays out of state_dict)
        st.begin()
    try:
        logits = model(x)
        loss = criterion(logits, target)
        if st is None:
            optimizer.zero_grad()
        loss.backward()
    except BaseException:
        if st is not None:
            st.end()
        raise
    if st is not None:
        st.step(optimizer, grad_clip, group)
        return loss.detach(), logits.detach()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        grads = [p.grad for p in model.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, group=group)
        flat.mul_(1.0 / dist.get_world_size(group))
        torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])
    if grad_clip > 0:
        nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
    optimizer.step()
    return loss.detach(), logits.detach()


def validate(model, val_queue, criterion=None):
    """train_eval.py:271-293: eval mode (running statistics), top-1 / top-5 / loss."""
    from .search import AverageMeter, accuracy
    criterion = criterion or F.cross_entropy
    objs, top1, top5 = AverageMeter(), AverageMeter(), AverageMeter()
    model.eval()
    dev = next(model.parameters()).device
    for x, target in val_queue:
        x, target = x.to(dev, non_blocking=True), target.to(dev, non_blocking=True)
        with torch.no_grad():
            logits = model(x)
            loss = criterion(logits, target)
        p1, p5 = accuracy(logits, target, topk=(1, 5))
        vals = torch.stack([loss.float(), p1, p5]).tolist()
        n = x.size(0)
        objs.update(vals[0], n)
        top1.update(vals[1], n)
        top5.update(vals[2], n)
    return top1.avg, top5.avg, objs.avg


def build_derived_network(num_classes, model_path=None, config_path=None, dropout_rate=0.2, drop_connect_rate=0.2):
    """train_eval.py:102-115: the derived network from a search checkpoint (``--model_path``: parse the architecture, widths
    from its masks) or from an exported ``model.config`` (``--config_path``)."""
    import json
    import os
    if model_path and os.path.isfile(model_path):
        from .epoch import get_op_and_depth_weights, parse_architecture
        ck = torch.load(model_path, map_location='cpu', weights_only=False)
        parsed = parse_architecture(*get_op_and_depth_weights(ck['state_dict']))
        return Network(num_classes, parsed, geometry.get_mc_num_dddict(ck['mc_mask_dddict']), None, dropout_rate,
                       drop_connect_rate)
    if config_path and os.path.isfile(config_path):
        return NetworkCfg(num_classes, json.load(open(config_path)), None, dropout_rate, drop_connect_rate)
    raise ValueError('invalid model_path and config_path')


def run_retrain(save_dir, model, make_train_queue, make_val_queue, *, epochs=250, lr=0.2, momentum=0.9, weight_decay=4e-5,
                label_smooth=0.1, grad_clip=5.0, batch_size=256, snapshot=None, device='cuda', group=None, log=print):
    """The retrain schedule of train_eval.py:118-226: SGD + cosine learning rate (5 warm-up epochs of linearly increasing rate
    when batch_size > 256), label-smoothed training loss, plain cross-entropy validation every epoch, ``checkpoint.pth.tar`` /
    ``model_best.pth.tar`` with the reference's keys ('epoch', 'state_dict' with DataParallel's ``module.`` prefix,
    'best_acc_top1', 'best_acc_top5', 'optimizer'), resume from ``snapshot``, ``model.config`` written next to them."""
    import json
    import math
    import os
    import shutil
    from .search import AverageMeter, accuracy
    os.makedirs(save_dir, exist_ok=True)
    dev = torch.device(device)
    model = model.to(dev)
    num_classes = model.classifier.out_features
    with open(os.path.join(save_dir, 'model.config'), 'w') as f:
        json.dump(model.config, f, indent=4)
    crit_smooth = CrossEntropyLabelSmooth(num_classes, label_smooth)
    opt = torch.optim.SGD(model.parameters(), lr, momentum=momentum, weight_decay=weight_decay)
    best1 = best5 = 0.0
    start = 0
    if snapshot:
        ck = torch.load(snapshot, map_location=dev, weights_only=False)
        start, best1, best5 = ck['epoch'], ck['best_acc_top1'], ck['best_acc_top5']
        model.load_state_dict({k[len('module.'):] if k.startswith('module.') else k: v for k, v in ck['state_dict'].items()})
        opt.load_state_dict(ck['optimizer'])
    history = []
    for epoch in range(start, epochs):
        cur = 0.5 * lr * (1.0 + math.cos(math.pi * epoch / float(epochs)))          # CosineAnnealingLR(T_max=epochs).get_lr()
        warm = epoch < 5 and batch_size > 256
        for g in opt.param_groups:
            g['lr'] = cur * (epoch + 1) / 5.0 if warm else cur
        objs, top1 = AverageMeter(), AverageMeter()
        for x, y in make_train_queue(epoch):
            x, y = x.to(dev, non_blocking=True), y.to(dev, non_blocking=True)
            loss, logits = train_step(model, x, y, crit_smooth, opt, grad_clip, group)
            p1, = accuracy(logits, y, topk=(1,))
            vals = torch.stack([loss.float(), p1]).tolist()
            objs.update(vals[0], x.size(0))
            top1.update(vals[1], x.size(0))
        v1, v5, vobj = validate(model, make_val_queue(epoch))
        is_best = v1 > best1
        if is_best:
            best1, best5 = v1, v5
        state = {'epoch': epoch + 1, 'state_dict': {'module.' + k: v for k, v in model.state_dict().items()},
                 'best_acc_top1': best1, 'best_acc_top5': best5, 'optimizer': opt.state_dict()}
        path = os.path.join(save_dir, 'checkpoint.pth.tar')
        torch.save(state, path)
        if is_best:
            shutil.copyfile(path, os.path.join(save_dir, 'model_best.pth.tar'))
        history.append(dict(epoch=epoch, lr=opt.param_groups[0]['lr'], train_acc=top1.avg, train_obj=objs.avg, val_top1=v1,
                            val_top5=v5, val_obj=vobj))
        log('Epoch %d lr %e train_acc %f val_top1 %f val_top5 %f' % (epoch, opt.param_groups[0]['lr'], top1.avg, v1, v5))
    return history
