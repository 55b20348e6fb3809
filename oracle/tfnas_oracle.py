"""CPU ORACLE for the TF-NAS supernet-search hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32, CPU) restatement of the reference algorithm.  It exists so the
hand-written HIP path in ``tf-nas_amd/`` has something to be checked against on a box where
``/root/reference`` does not exist.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package never does.

Pinning: ``tests/test_oracle_vs_reference.py`` (runs in the build container, where the reference can be
imported) proves this restatement equal to the reference's own classes -- same seeds, same injected
Gumbel noise, same ``random`` indices -- to ~1e-6, and ``tests/golden/make_golden.py`` captures golden
vectors from the *reference* that ``tests/test_oracle_golden.py`` replays anywhere.  The reference
ships no tests of its own for this path, so those imports are the only pin.

What each piece follows (paths relative to /root/reference):
  mbconv_forward      models/layers.py:539-561   (MBInvertedResBlock.forward; ctor :433-537)
  _bn                 models/layers.py:469,498,533  BatchNorm2d(affine=False, track_running_stats=False)
  _act                models/layers.py:26-35 (Swish), :470-471 (ReLU)
  gumbel_softmax      torch.nn.functional.gumbel_softmax as called at models/model_search.py:62,66,87
  MixedOP.forward     models/model_search.py:58-91    get_lookup_latency :93-111
  MixedStage.forward  models/model_search.py:157-206
  Network.forward     models/model_search.py:281-304  parameter helpers :306-350
  w_step / a_step     train_search.py:366-426 (train_w_arch body); warmup_step :329-349 (train_wo_arch)
"""
import math
import random
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

BN_EPS = 1e-5
NUM_OPS = 8
OP_KERNEL = (3, 3, 5, 5, 3, 3, 5, 5)
OP_SE_MULT = (0, 0, 0, 0, 1, 2, 1, 2)
STAGE_CFG = OrderedDict([      # ics, ocs, strides, act   (models/model_search.py:221-275)
    ('stage1', ([16, 24], [24, 24], [2, 1], 'relu')),
    ('stage2', ([24, 40, 40], [40, 40, 40], [2, 1, 1], 'swish')),
    ('stage3', ([40, 80, 80, 80], [80, 80, 80, 80], [2, 1, 1, 1], 'swish')),
    ('stage4', ([80, 112, 112, 112], [112, 112, 112, 112], [1, 1, 1, 1], 'swish')),
    ('stage5', ([112, 192, 192, 192], [192, 192, 192, 192], [2, 1, 1, 1], 'swish')),
    ('stage6', ([192], [320], [1], 'swish')),
])


# --------------------------------------------------------------------------- functional pieces
def _bn(x):
    """Batch-statistics BN without affine or running stats: (x-mean)/sqrt(var_biased+eps)."""
    return F.batch_norm(x, None, None, None, None, True, 0.0, BN_EPS)


# Test hook (tests/_hipcheck.ReluInjector): a callable(pre_activation) -> bool mask or None.  With a mask the ReLU is evaluated
# as x * mask, i.e. forward AND backward take the injected side of relu'(0) -- used to replay the HIP kernels' own ReLU decisions
# in the oracle, so that elements whose pre-activation sits within fp32 rounding of 0 do not need a tolerance exemption.
RELU_HOOK = None


def _act(x, act):
    if act == 'relu':
        if RELU_HOOK is not None:
            m = RELU_HOOK(x)
            if m is not None:
                return x * m.to(x.dtype)
        return F.relu(x)
    if act == 'swish':
        return x * torch.sigmoid(x)
    raise ValueError(act)


def mbconv_forward(x, p, k, stride, act, has_res, detail=None):
    """One MBConv candidate.  ``p`` maps 'expand','dw','proj' (and 'se_rw','se_rb','se_ew','se_eb')
    to OIHW weight tensors; 'expand' may be absent (no inverted bottleneck when mid==in).
    If ``detail`` is a dict, raw intermediates (pre-BN conv outputs, gate) are stored in it."""
    y = x
    if p.get('expand') is not None:
        e = F.conv2d(y, p['expand'])
        eh = _bn(e)
        if detail is not None:
            detail['E'], detail['Eh'] = e, eh
        y = _act(eh, act)
    d = F.conv2d(y, p['dw'], None, stride, k // 2, 1, p['dw'].shape[0])
    dh = _bn(d)
    if detail is not None:
        detail['D'], detail['Dh'] = d, dh
    y = _act(dh, act)
    if p.get('se_rw') is not None:
        pooled = F.adaptive_avg_pool2d(y, 1)
        hidden = _act(F.conv2d(pooled, p['se_rw'], p['se_rb']), act)
        gate = torch.sigmoid(F.conv2d(hidden, p['se_ew'], p['se_eb']))
        if detail is not None:
            detail['pooled'], detail['gate'] = pooled, gate
        y = y * gate
    pr = F.conv2d(y, p['proj'])
    if detail is not None:
        detail['Z'], detail['P'] = y, pr
    y = _bn(pr)
    if has_res:
        y = y + x
    return y


def gumbel_softmax(logits, tau, exp_noise=None):
    """softmax((logits - log(e)) / tau) with e ~ Exp(1); same arithmetic as torch's F.gumbel_softmax
    (hard=False).  ``exp_noise`` injects the e draws; None draws from torch's default generator in the
    same way torch does (so a seeded run reproduces the reference's RNG stream)."""
    if exp_noise is None:
        e = torch.empty_like(logits).exponential_()
    else:
        e = exp_noise[..., :logits.shape[-1]].to(logits.dtype)
    gumbels = -e.log()
    return ((logits + gumbels) / tau).softmax(-1)


# --------------------------------------------------------------------------- modules (parameter layout == reference)
def _seq(**mods):
    return nn.Sequential(OrderedDict(mods))


class MBConv(nn.Module):
    """Parameter container with the reference's attribute names (train_search.py:164-193 reaches them)."""
    name = 'MBInvertedResBlock'

    def __init__(self, ic, mc, se, oc, k, stride, act):
        super().__init__()
        self.in_channels, self.mid_channels, self.se_channels, self.out_channels = ic, mc, se, oc
        self.kernel_size, self.stride, self.act_func = k, stride, act
        if mc > ic:
            self.inverted_bottleneck = _seq(conv=nn.Conv2d(ic, mc, 1, bias=False))
        else:
            self.inverted_bottleneck = None
            self.mid_channels = mc = ic
        self.depth_conv = _seq(conv=nn.Conv2d(mc, mc, k, stride, k // 2, groups=mc, bias=False))
        if se > 0:
            self.squeeze_excite = _seq(conv_reduce=nn.Conv2d(mc, se, 1, bias=True),
                                       conv_expand=nn.Conv2d(se, mc, 1, bias=True))
        else:
            self.squeeze_excite = None
        self.point_linear = _seq(conv=nn.Conv2d(mc, oc, 1, bias=False))
        self.has_residual = (ic == oc) and (stride == 1)

    def params(self):
        p = {'dw': self.depth_conv.conv.weight, 'proj': self.point_linear.conv.weight}
        if self.inverted_bottleneck is not None:
            p['expand'] = self.inverted_bottleneck.conv.weight
        if self.squeeze_excite is not None:
            p.update(se_rw=self.squeeze_excite.conv_reduce.weight, se_rb=self.squeeze_excite.conv_reduce.bias,
                     se_ew=self.squeeze_excite.conv_expand.weight, se_eb=self.squeeze_excite.conv_expand.bias)
        return p

    def forward(self, x, detail=None):
        return mbconv_forward(x, self.params(), self.kernel_size, self.stride, self.act_func,
                              self.has_residual, detail)


class MixedOP(nn.Module):
    def __init__(self, ic, oc, stride, act, mc_num_dict, lat_lookup):
        super().__init__()
        self.num_ops, self.lat_lookup, self.mc_num_dict = NUM_OPS, lat_lookup, mc_num_dict
        self.m_ops = nn.ModuleList(
            MBConv(ic, mc_num_dict[i], ic * OP_SE_MULT[i], oc, OP_KERNEL[i], stride, act) for i in range(NUM_OPS))
        self.log_alphas = nn.Parameter(F.log_softmax(torch.zeros(NUM_OPS), dim=-1))
        self.reset_switches()
        self.T = 1.0

    def reset_switches(self):
        self.switches = [True] * self.num_ops

    def set_temperature(self, T):
        self.T = T

    def _nth_active(self, pos):
        """Original index of the pos-th still-switched-on candidate (reference: fink_ori_idx)."""
        return [i for i, s in enumerate(self.switches) if s][pos]

    def get_lookup_latency(self, size):
        return [self.lat_lookup['{}_{}_{}_{}_{}_k{}_s{}_{}'.format(
            op.name, size, op.in_channels, op.se_channels, op.out_channels, op.kernel_size, op.stride,
            op.act_func)][op.mid_channels] for op in self.m_ops]

    def forward(self, x, sampling, mode, exp_noise=None, rand_pos=None):
        """exp_noise: tensor of Exp(1) draws ([8] soft; [#active] sampled-gumbel) or None (torch RNG).
        rand_pos: position among the remaining candidates for mode 'random' or None (python `random`)."""
        if sampling:
            active = self.log_alphas[self.switches]
            if mode in ('gumbel', 'gumbel_2'):
                w = gumbel_softmax(F.log_softmax(active, dim=-1), self.T, exp_noise)
                pos = int(torch.argmax(w).item())
                if mode == 'gumbel':
                    idx = pos                       # raw position; valid because all switches are on here
                    self.switches[idx] = False
                else:
                    idx = self._nth_active(pos)
                    self.reset_switches()
            elif mode in ('min_alphas', 'max_alphas'):
                pos = int((torch.argmin if mode == 'min_alphas' else torch.argmax)(active).item())
                idx = self._nth_active(pos)
                self.reset_switches()
            elif mode == 'random':
                pos = random.choice(range(len(active))) if rand_pos is None else int(rand_pos)
                idx = self._nth_active(pos)
                self.reset_switches()
            else:
                raise ValueError('invalid sampling mode...')
            self.last_idx = idx
            return self.m_ops[idx](x), 0
        w = gumbel_softmax(self.log_alphas, self.T, exp_noise)
        lats = self.get_lookup_latency(x.size(-1))
        out = sum(wi * op(x) for wi, op in zip(w, self.m_ops))
        out_lat = sum(wi * lat for wi, lat in zip(w, lats))
        self.last_w = w
        return out, out_lat


class MixedStage(nn.Module):
    def __init__(self, ics, ocs, ss, act, mc_num_ddict, lat_lookup):
        super().__init__()
        self.nblocks = len(ics)
        for b, (ic, oc, s) in enumerate(zip(ics, ocs, ss), start=1):
            setattr(self, 'block%d' % b, MixedOP(ic, oc, s, act, mc_num_ddict['block%d' % b], lat_lookup))
        self.start_res = 0 if (ics[0] == ocs[0] and ss[0] == 1) else 1
        self.betas = nn.Parameter(torch.zeros(len(ics) - self.start_res + 1))

    def blocks(self):
        return [getattr(self, 'block%d' % b) for b in range(1, self.nblocks + 1)]

    def forward(self, x, sampling, mode, exp_noise=None, rand_pos=None):
        outs, lats = [x], [0.]
        cum = None
        for b, blk in enumerate(self.blocks()):
            y, lat = blk(outs[-1], sampling, mode,
                         None if exp_noise is None else exp_noise[b],
                         None if rand_pos is None else rand_pos[b])
            cum = lat if cum is None else cum + lat
            outs.append(y)
            lats.append(cum)
        w = F.softmax(self.betas, dim=-1)
        out = sum(wk * r for wk, r in zip(w, outs[self.start_res:]))
        out_lat = sum(wk * l for wk, l in zip(w, lats[self.start_res:]))
        return out, out_lat


class _Stem(nn.Module):
    def __init__(self, ic, oc, k, stride):
        super().__init__()
        self.conv = nn.Conv2d(ic, oc, k, stride, k // 2, bias=False)


class _Classifier(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.linear = nn.Linear(i, o)


class Network(nn.Module):
    def __init__(self, num_classes, mc_num_dddict, lat_lookup):
        super().__init__()
        self.lat_lookup, self.mc_num_dddict = lat_lookup, mc_num_dddict
        self.first_stem = _Stem(3, 32, 3, 2)
        self.second_stem = MBConv(32, 32, 8, 16, 3, 1, 'relu')
        for name, (ics, ocs, ss, act) in STAGE_CFG.items():
            setattr(self, name, MixedStage(ics, ocs, ss, act, mc_num_dddict[name], lat_lookup))
        self.feature_mix_layer = _Stem(320, 1280, 1, 1)
        self.classifier = _Classifier(1280, num_classes)
        for m in self.modules():                       # model_search.py:352-364 (only biases are touched)
            if isinstance(m, (nn.Conv2d, nn.Linear)) and m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def stages(self):
        return [getattr(self, n) for n in STAGE_CFG]

    def cells(self):
        return [blk for st in self.stages() for blk in st.blocks()]

    def forward(self, x, sampling, mode='max', exp_noise=None, rand_pos=None):
        """exp_noise: [18, 8] Exp(1) draws in cell (module) order; rand_pos: 18 ints."""
        out_lat = self.lat_lookup['base'] if not sampling else 0.0
        x = _act(_bn(F.conv2d(x, self.first_stem.conv.weight, None, 2, 1)), 'relu')
        x = self.second_stem(x)
        c = 0
        for st in self.stages():
            n = st.nblocks
            x, lat = st(x, sampling, mode,
                        None if exp_noise is None else exp_noise[c:c + n],
                        None if rand_pos is None else rand_pos[c:c + n])
            out_lat = out_lat + lat
            c += n
        x = _act(_bn(F.conv2d(x, self.feature_mix_layer.conv.weight)), 'swish')
        x = F.adaptive_avg_pool2d(x, 1).flatten(1)
        return self.classifier.linear(x), out_lat

    # -- helpers with the reference's names
    def set_temperature(self, T):
        for c in self.cells():
            c.set_temperature(T)

    def reset_switches(self):
        for c in self.cells():
            c.reset_switches()

    def weight_parameters(self):
        return [v for k, v in self.named_parameters() if not (k.endswith('log_alphas') or k.endswith('betas'))]

    def arch_parameters(self):
        return [v for k, v in self.named_parameters() if k.endswith('log_alphas') or k.endswith('betas')]

    def log_alphas_parameters(self):
        return [v for k, v in self.named_parameters() if k.endswith('log_alphas')]

    def betas_parameters(self):
        return [v for k, v in self.named_parameters() if k.endswith('betas')]


# --------------------------------------------------------------------------- the search iteration
def make_optimizers(model, w_lr=0.025, w_mom=0.9, w_wd=1e-5, a_lr=0.01, a_wd=5e-4, a_betas=(0.5, 0.999)):
    """train_search.py:197-206."""
    opt_w = torch.optim.SGD(model.weight_parameters(), lr=w_lr, momentum=w_mom, weight_decay=w_wd)
    opt_a = torch.optim.Adam(model.arch_parameters(), lr=a_lr, betas=a_betas, weight_decay=a_wd)
    return opt_w, opt_a


def _set_requires_grad(model, weights, arch):
    for p in model.weight_parameters():
        p.requires_grad = weights
    for p in model.arch_parameters():
        p.requires_grad = arch


def w_step(model, x, target, opt_w, grad_clip=5.0, noise_g=None, rand_pos=None, bi_sampling=True):
    """Weight step (train_search.py:370-385; bi_sampling=False gives train_wo_arch :329-342).
    Returns (loss, logits_gumbel, gumbel_idx[18], random_idx[18] or None)."""
    _set_requires_grad(model, True, False)
    logits_g, _ = model(x, True, 'gumbel', exp_noise=noise_g)
    gidx = [c.last_idx for c in model.cells()]
    loss = F.cross_entropy(logits_g, target)
    ridx = None
    if bi_sampling:
        logits_r, _ = model(x, True, 'random', rand_pos=rand_pos)
        ridx = [c.last_idx for c in model.cells()]
        loss = loss + F.cross_entropy(logits_r, target)
    else:
        model.reset_switches()
    opt_w.zero_grad()
    loss.backward()
    if grad_clip > 0:
        nn.utils.clip_grad_norm_(model.weight_parameters(), grad_clip)
    opt_w.step()
    return loss.detach(), logits_g.detach(), gidx, ridx


def a_step(model, x, target, opt_a, target_lat=15.0, lambda_lat=0.1, grad_clip=5.0, noise=None):
    """Architecture step (train_search.py:404-422).  Returns (loss_a, loss_l, lat, arch grads before clip)."""
    _set_requires_grad(model, False, True)
    logits, lat = model(x, False, exp_noise=noise)
    loss_a = F.cross_entropy(logits, target)
    loss_l = torch.abs(lat / target_lat - 1.) * lambda_lat
    loss = loss_a + loss_l
    opt_a.zero_grad()
    loss.backward()
    grads = [p.grad.detach().clone() for p in model.arch_parameters()]
    if grad_clip > 0:
        nn.utils.clip_grad_norm_(model.arch_parameters(), grad_clip)
    opt_a.step()
    for p in model.arch_parameters():               # log-softmax projection of alphas AND betas (:421-422)
        p.data = F.log_softmax(p.detach().data, dim=-1)
    return loss_a.detach(), loss_l.detach(), lat.detach(), grads


def validate(model, batches, noises):
    """``validate`` (train_search.py:435-462): no-grad 'gumbel' forward per batch in train mode, reset_switches after
    each, running averages weighted by batch size.  Returns (top1.avg, top5.avg, objs.avg, [gumbel indices per batch])."""
    tot = n_all = t1 = t5 = 0.0
    idxs = []
    for (x, y), e in zip(batches, noises):
        with torch.no_grad():
            logits, _ = model(x, True, 'gumbel', exp_noise=e)
            loss = F.cross_entropy(logits, y)
        idxs.append([c.last_idx for c in model.cells()])
        model.reset_switches()
        _, pred = logits.topk(5, 1, True, True)
        hit = pred.t().eq(y.view(1, -1))
        n = x.size(0)
        t1 += float(hit[:1].sum()) * 100.0
        t5 += float(hit[:5].sum()) * 100.0
        tot += float(loss) * n
        n_all += n
    return t1 / n_all, t5 / n_all, tot / n_all, idxs


# --------------------------------------------------------------------------- derived ("retrain") network
class _ConvBN(nn.Module):
    """conv -> BatchNorm2d(affine, running stats) -> act   (models/layers.py ConvLayer with affine=True, 'weight_bn_act')."""

    def __init__(self, ic, oc, k, stride, act):
        super().__init__()
        self.bn = nn.BatchNorm2d(oc)                      # (registered before conv, like BasicLayer: state_dict order)
        self.conv = nn.Conv2d(ic, oc, k, stride, k // 2, bias=False)
        self.act = act

    def forward(self, x):
        return _act(self.bn(self.conv(x)), self.act)


class DerivedBlock(nn.Module):
    """MBInvertedResBlock with affine BatchNorm (models/layers.py:431-561, affine=True) incl. drop-connect on the residual
    branch (tools/utils.py:77-86).  ``drop_u``: optional injected U[0,1) draws [N] (else torch.rand, like the reference)."""

    def __init__(self, ic, mc, se, oc, k, stride, act):
        super().__init__()
        self.in_channels, self.se_channels, self.out_channels = ic, se, oc
        self.kernel_size, self.stride, self.act_func = k, stride, act
        self.drop_connect_rate = 0.0
        self.drop_u = None
        if mc > ic:
            self.inverted_bottleneck = _seq(conv=nn.Conv2d(ic, mc, 1, bias=False), bn=nn.BatchNorm2d(mc))
        else:
            self.inverted_bottleneck, mc = None, ic
        self.mid_channels = mc
        self.depth_conv = _seq(conv=nn.Conv2d(mc, mc, k, stride, k // 2, groups=mc, bias=False), bn=nn.BatchNorm2d(mc))
        self.squeeze_excite = _seq(conv_reduce=nn.Conv2d(mc, se, 1), conv_expand=nn.Conv2d(se, mc, 1)) if se > 0 else None
        self.point_linear = _seq(conv=nn.Conv2d(mc, oc, 1, bias=False), bn=nn.BatchNorm2d(oc))
        self.has_residual = ic == oc and stride == 1

    def forward(self, x):
        res = x
        if self.inverted_bottleneck is not None:
            x = _act(self.inverted_bottleneck.bn(self.inverted_bottleneck.conv(x)), self.act_func)
        x = _act(self.depth_conv.bn(self.depth_conv.conv(x)), self.act_func)
        if self.squeeze_excite is not None:
            s = F.adaptive_avg_pool2d(x, 1)
            s = self.squeeze_excite.conv_expand(_act(self.squeeze_excite.conv_reduce(s), self.act_func))
            x = x * torch.sigmoid(s)
        x = self.point_linear.bn(self.point_linear.conv(x))
        if self.has_residual:
            if self.training and self.drop_connect_rate > 0.0:
                keep = 1.0 - self.drop_connect_rate
                u = self.drop_u if self.drop_u is not None else torch.rand(x.size(0))
                x = x.div(keep) * torch.floor(keep + u.view(-1, 1, 1, 1))
            x = x + res
        return x


class DerivedNetwork(nn.Module):
    """models/model_eval.Network (:31-131): stems, the chosen candidate of every kept block, feature mix, pool, dropout, FC."""

    def __init__(self, num_classes, parsed_arch, mc_num_dddict, dropout_rate=0.0, drop_connect_rate=0.0):
        super().__init__()
        self.dropout_rate = dropout_rate
        count = 1 + sum(len(parsed_arch[st]) for st in parsed_arch)
        idx = 1
        self.first_stem = _ConvBN(3, 32, 3, 2, 'relu')
        self.second_stem = DerivedBlock(32, 32, 8, 16, 3, 1, 'relu')
        self.second_stem.drop_connect_rate = drop_connect_rate * idx / count
        for name, (ics, ocs, ss, act) in STAGE_CFG.items():
            stage = nn.ModuleList()
            for i, blk in enumerate(parsed_arch[name]):
                idx += 1
                op = parsed_arch[name][blk]
                b = DerivedBlock(ics[i], mc_num_dddict[name][blk][op], ics[i] * OP_SE_MULT[op], ocs[i], OP_KERNEL[op], ss[i], act)
                b.drop_connect_rate = drop_connect_rate * idx / count
                stage.append(b)
            setattr(self, name, stage)
        self.feature_mix_layer = _ConvBN(320, 1280, 1, 1, 'swish')
        self.classifier = _seq(linear=nn.Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)) and m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def blocks(self):
        return [b for name in STAGE_CFG for b in getattr(self, name)]

    def forward(self, x):
        x = self.second_stem(self.first_stem(x))
        for b in self.blocks():
            x = b(x)
        x = F.adaptive_avg_pool2d(self.feature_mix_layer(x), 1).flatten(1)
        if self.dropout_rate > 0.0:
            x = F.dropout(x, p=self.dropout_rate, training=self.training)
        return self.classifier.linear(x)


def label_smooth_loss(logits, target, num_classes, epsilon):
    """CrossEntropyLabelSmooth (train_eval.py:72-85)."""
    lp = F.log_softmax(logits, 1)
    t = torch.zeros_like(lp).scatter_(1, target.unsqueeze(1), 1)
    t = (1 - epsilon) * t + epsilon / num_classes
    return (-t * lp).mean(0).sum()


def initial_mc_num_dddict(e3=3, e6=6):
    d = OrderedDict()
    for name, (ics, ocs, ss, act) in STAGE_CFG.items():
        d[name] = OrderedDict(('block%d' % b, OrderedDict(
            (i, ic * (e3 if i % 2 == 0 else e6)) for i in range(NUM_OPS))) for b, ic in enumerate(ics, start=1))
    return d
