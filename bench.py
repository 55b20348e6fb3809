#!/usr/bin/env python3
"""Benchmark of the TF-NAS supernet search step on MI355X (BASELINE.json metric: search images/sec).

A "step" here is one ITERATION PAIR of the reference's train_w_arch loop (train_search.py:366-426):
w-step on a train batch, alpha-step on a val batch (every even iteration), w-step on the next train batch --
i.e. 2*B train images (and B val images) per step per GPU.  `value` = train images consumed per second by the
whole job (all ranks), inputs already resident in HBM, optimizer steps / clipping / projection included.

  python bench.py --gpus 1 --steps 10 --warmup 3                  (single GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, RCCL)

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant HIP kernel family of the timed region, timed live with HIP events on the launch
                stream (tfnas_prof_* hooks), vs its algorithmic flops / HBM bytes;
                roofline.pair = the whole iteration pair against SURVEY.md 8(d)'s algorithmic floor
                (t_lower_ms, achieved = t_lower / t, hbm_fraction vs F0, mfma_fraction of the 1x1 flops).
  kernel_roofline  every modelled kernel family of one profiled pair: ms per pair, nearer bound, fraction of it.
  retrain       BASELINE configs[4] at its own size: derived-network retrain step at 224x224, 256 images per GPU
                (retrain_images_per_s; finiteness + bit-determinism checked).
  w_step_ms / a_step_ms   GPU time of the two kinds of step (HIP events on the launch stream, a few pairs after
                the timed region).
  dropin_images_per_s     the same iteration pair written exactly like the reference's train_w_arch body
                (train_search.py:370-426: sequential model(x, True, 'gumbel') / model(x, True, 'random') forwards,
                sampling on the device, torch's clip_grad_norm_ / per-parameter projection loop) -- what a user gets
                from the two-line import swap of INTEGRATION.md without adopting tfnas_amd.search.
  cpu_baseline  the CPU oracle (plain PyTorch restatement, proven equal to the reference) timed on this host's
                cores on a bounded sample of the same loop at the reference's batch 32 (N=1, rank 0 only).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))

# Hardware queues: leave GPU_MAX_HW_QUEUES at its default (4).  Round 1 raised it to 8 under torch.distributed.run; with the
# path level of round 2 that setting costs 30 % of a pair once the all-reduce runs (108 vs 83 ms, 1 rank, forced all-reduce),
# and what round 1 was compensating for -- a 25 % slower w-step as soon as RCCL was initialised -- came from EAGER
# communicator creation (init_process_group(device_id=...)); the lazy default has no such effect.

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_FP32_MFMA_TF = 157.3    # v_mfma_f32_16x16x4_f32 rate (== fp32 vector peak)


def cell_table(B):
    """(name, N, H, W, ic, oc, stride, mids) of the 18 cells at initial widths."""
    from tfnas_amd import geometry as g
    rows = []
    for stage, block, ic, oc, s, act, size in g.iter_cells():
        mids = [g.init_mid_channels(ic, i) for i in range(8)]
        rows.append((stage + '.' + block, B, size, size, ic, oc, s, mids))
    return rows


def family_algorithmic(fam, B, mode=None):
    """Algorithmic (flops, bytes) summed over the KERNEL launches one iteration pair makes of kernel family `fam`,
    plus that launch count.  alpha-step: all 8 candidates of the 18 cells (backward kernels skip the first cell,
    whose input needs no gradient); w-step: 2 paths x 18 cells x 1 candidate x 2 w-steps (expected value over
    uniform candidate choice).  Depthwise families launch one kernel per kernel size present (2 in the soft
    mode).  mode = 'soft' / 'sampled': only the alpha-step launches / only the w-step launches (stems included).  fp32 = 4 B/element.  Only families that can dominate are modelled; others return None."""
    from tfnas_amd import geometry as g
    fl = by = 0.0
    n = 0
    backward = fam in ('k_project_dgrad', 'k_expand_dgrad', 'k_dw_bwd_data', 'k_se_pool<bwd>')
    for ci, (name, N, H, W, ic, oc, s, mids) in enumerate(cell_table(B)):
        P, Po = N * H * W, N * ((H - 1) // s + 1) * ((W - 1) // s + 1)
        Msoft = sum(mids)
        Mavg = Msoft / 8.0
        for M, launches in ((Msoft, 1), (Mavg, 4)):          # soft launch once, sampled launch 4x per pair
            G = 8 if launches == 1 else 1
            if (mode == 'soft' and launches != 1) or (mode == 'sampled' and launches == 1):
                continue
            if launches == 1 and backward and ci == 0:
                continue                                       # pruned: first cell of the alpha-step
            kernels = launches * (2 if (fam.startswith('k_dw_') and launches == 1) else 1)
            # E-free mode (functions.py policy): stride-2 cells with ic <= 24 in the alpha-step never form E; the
            # depthwise kernels recompute it from x (2*P*ic*M extra flops) and BN1's statistics are two passes over x
            efree = launches == 1 and s == 2 and ic <= 24
            # BN2-backward tables in the epilogue of k_project_dgrad (capi.hip): one-candidate launches and images <= 28 x 28
            HWo = Po // N
            folded = HWo >= 43 and (launches != 1 or HWo <= 784)
            rec_bytes = 4.0 * (Po / 128.0) * (1.0 + 128.0 / HWo) * 3.5 * M    # ~(5 + 2) / 2 sums per (row tile, image, channel)
            # fused per-image route (csrc/fx_kernels.hip; tfnas_fx_supported): the alpha-step launches of the stride-1 cells at 14 x 14
            # / 7 x 7 run expand + BN1 + act + depthwise in one kernel per direction (ehat kept in the E buffer for the backward, dE
            # never formed: partial sums of dx per channel slice, ~8 slices); BN1 statistics from the Gram matrix of x
            fx = launches == 1 and s == 1 and H * W <= 196 and 64 <= ic <= 192 and os.environ.get('TFNAS_FX', '1') != '0'
            nsl = 8.0
            if fam == 'k_expand_fwd' and fx:
                f, b, kernels = 2.0 * P * ic * ic, 4.0 * (2 * P * ic), 2
            elif fam == 'k_dw_fwd' and fx:
                f, b, kernels = 2.0 * Po * M * 17.0 + 2.0 * P * ic * M, 4.0 * (P * ic + Po * M + P * M), 1
            elif fam == 'k_dw_bwd_data' and fx:
                f, b, kernels = 2.0 * P * M * 17.0 + 2.0 * P * M * ic, 4.0 * (2 * Po * M + P * M + nsl * P * ic), 1
            elif fam == 'k_expand_dgrad' and fx:
                f, b = 2.0 * P * ic * ic, 4.0 * ((nsl + 3) * P * ic)
            elif fam == 'k_expand_fwd' and efree:
                f, b, kernels = 2.0 * P * ic * ic, 4.0 * (2 * P * ic), 2
            elif fam == 'k_dw_fwd' and efree:
                f, b = 2.0 * Po * M * 17.0 + 2.0 * P * ic * M, 4.0 * (P * ic + Po * M)
            elif fam == 'k_dw_bwd_data' and efree:
                f, b = 2.0 * P * M * 17.0 / (s * s) + 2.0 * P * ic * M, 4.0 * (2 * Po * M + P * ic + P * M)
            elif fam == 'k_expand_fwd':
                f, b = 2.0 * P * ic * M, 4.0 * (P * ic + P * M + M * ic)
            elif fam == 'k_project_fwd':
                f, b = 2.0 * Po * M * oc, 4.0 * (Po * M + G * Po * oc + M * oc)
            elif fam == 'k_project_dgrad':
                f, b = 2.0 * Po * M * oc, 4.0 * (2 * G * Po * oc + Po * M + M * oc)
                if folded:                                     # FOLD epilogue (capi.hip policy): + one read of D + the records
                    b += 4.0 * Po * M + rec_bytes
            elif fam == 'k_expand_dgrad':
                f, b = 2.0 * P * M * ic, 4.0 * (2 * P * M + P * ic + M * ic)
            elif fam == 'k_dw_fwd':
                kk = 17.0                                      # mean of 9 and 25 taps
                f, b = 2.0 * Po * M * kk, 4.0 * (P * M + Po * M)
            elif fam == 'k_dw_bwd_data':
                kk = 17.0
                f, b = 2.0 * P * M * kk / (s * s), 4.0 * (2 * Po * M + 2 * P * M)
                if launches != 1 and s == 2:
                    f += 2.0 * Po * M * kk                               # + the depthwise weight gradient (same pass)
            elif fam == 'k_se_pool<bwd>':                      # k_bn2_pool: ONE pass over (dZ, D); k_bn2_finish is tiny
                f, b = 8.0 * Po * M, 4.0 * (2 * Po * M)
                if folded:                                     # k_bn2_gather: only the records of the dgrad epilogue
                    f, b = 0.0, rec_bytes
            elif fam in ('k_project_wgrad', 'k_expand_wgrad', 'k_dw_wgrad'):
                if launches == 1:
                    continue                                   # no weight grads in the alpha-step
                if fam == 'k_project_wgrad':
                    f, b = 2.0 * Po * M * oc, 4.0 * (Po * M + 2 * Po * oc)
                elif fam == 'k_expand_wgrad':
                    # Gram form where E is >= 100 MB (gemm_kernels.hip: expand_wgrad_xg): dEh and x, E is not read
                    xg = 4.0 * P * M >= 100 * 2 ** 20
                    f, b = 2.0 * P * M * ic, 4.0 * ((1 if xg else 2) * P * M + P * ic)
                elif s == 2:
                    f, b, kernels = 0.0, 0.0, 0                         # stride 2: from the backward pass (k_dwd_bwd<.., WG>)
                else:
                    f, b = 2.0 * Po * M * 17.0, 4.0 * (2 * Po * M + P * M)
            else:
                return None
            fl += f * launches
            by += b * launches
            n += kernels
    # the stem cell (first_stem + second_stem as ONE TFNAS_MODE_STEM cell, G = 1: mid 32, 112x112, stride 1, oc 16) launches
    # the same depthwise / project / pooling kernels: forward once per step (3 per pair; the w-step shares one stem
    # evaluation between its two paths), backward and weight gradients in the 2 w-steps only (frozen weights in the alpha-step)
    P = float(B) * 112 * 112
    M, oc = 32.0, 16.0
    stem = {'k_dw_fwd': (2.0 * P * M * 9.0, 4.0 * 2 * P * M, 3), 'k_dw_bwd_data': (2.0 * P * M * 9.0, 4.0 * 4 * P * M, 2),
            'k_project_fwd': (2.0 * P * M * oc, 4.0 * (P * M + P * oc), 3),
            'k_project_dgrad': (2.0 * P * M * oc, 4.0 * (2 * P * oc + P * M), 2),
            'k_se_pool<bwd>': (8.0 * P * M, 4.0 * 2 * P * M, 2),
            'k_project_wgrad': (2.0 * P * M * oc, 4.0 * (P * M + 2 * P * oc), 2),
            'k_dw_wgrad': (2.0 * P * M * 9.0, 4.0 * 3 * P * M, 2)}.get(fam)
    if stem is not None and mode != 'soft':                   # (the stems are one-candidate launches in both step kinds: the
        fl += stem[0] * stem[2]                               #  profiler's split, G > 2, files all of them under 'sampled')
        by += stem[1] * stem[2]
        n += stem[2]
    return fl, by, n


def kernel_roofline(fam_ms, B):
    """Every modelled kernel family of ONE profiled iteration pair against its own roofline: ms per pair (HIP events on the
    launch streams, other streams running beside it in the w-step), algorithmic GB/s and TF/s over that time, the bound that
    is nearer and the fraction of it."""
    rows = []
    for fam, (cnt, ms) in sorted(fam_ms.items(), key=lambda kv: -kv[1][1]):
        if not cnt:
            continue
        alg = family_algorithmic(fam, B)
        row = dict(family=fam, launches=int(cnt), ms_per_pair=round(ms, 3))
        if alg is not None and ms > 0:
            fl, by, _ = alg
            tf, gbs = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
            fm, fh = tf / PEAK_FP32_MFMA_TF, gbs / PEAK_HBM_GBS
            row.update(bound='mfma' if fm >= fh else 'hbm', frac=round(max(fm, fh), 4), tflops=round(tf, 2), gbs=round(gbs, 1))
        rows.append(row)
    return rows


def pair_algorithmic(B, elt=4):
    """algorithmic bytes F0 and flops of ONE iteration pair (w-step, alpha-step, w-step) at batch B.
    alpha-step, cell c: (3X + 2Y)*s + 2*sum_i P_i*s;  w-step: sum over the two sampled paths of (3X + 2Y)*s + P*s + 2*P*4
    (expected value over a uniform candidate choice).  Flops: fwd + input-grad (alpha-step, 2x fwd of all 8 candidates),
    fwd + dX + dW (w-step, 3x fwd of 2 of the 8).  Returns dict(bytes, flops_1x1, flops_other)."""
    f0_a = f0_w = mac_pw = mac_other = 0.0
    for name, N, H, W, ic, oc, s, mids in cell_table(B):
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        X, Y = float(N) * ic * H * W, float(N) * oc * Ho * Wo
        params, pw, other = [], 0.0, 0.0
        for i, mc in enumerate(mids):
            k = 3 if i in (0, 1, 4, 5) else 5
            se = 0 if i < 4 else ic * (1 if i % 2 == 0 else 2)
            params.append(mc * ic + mc * k * k + oc * mc + (2 * se * mc + se + mc if se else 0))
            pw += float(N) * (H * W * ic * mc + Ho * Wo * mc * oc)
            other += float(N) * (Ho * Wo * mc * k * k + (2 * se * mc if se else 0))
        f0_a += (3 * X + 2 * Y) * elt + 2 * sum(params) * elt
        pmean = sum(params) / 8.0
        f0_w += 2 * ((3 * X + 2 * Y) * elt + pmean * elt + 2 * pmean * 4)
        mac_pw += pw
        mac_other += other
    # alpha-step = 2 x fwd(all 8); each w-step = 3 x fwd x 2 paths / 8 candidates
    mult = 2.0 + 2 * 3.0 * 2.0 / 8.0
    return dict(bytes=f0_a + 2 * f0_w, flops_1x1=2.0 * mac_pw * mult, flops_other=2.0 * mac_other * mult,
                bytes_alpha=f0_a, bytes_w=f0_w)


def width_sweep_configs():
    """(name, mc_num_dddict) of SURVEY 8(d) C4: e=(2,4), (3,6), (4,8) and the ragged widths fit_mc_num_by_latency gives for
    target_lat 10 / 15 / 18 ms (three candidates scaled in turn, as tests/test_gpu_network.py::test_width_sweep_matches_oracle)."""
    from collections import OrderedDict
    from tfnas_amd import geometry as g
    from tfnas_amd.elasticity import fit_mc_num_by_latency
    from tfnas_amd.latency import get_lookup_latency, load_lat_lookup
    lut = load_lat_lookup('gpu')
    out = [('e2_e4', g.uniform_mc_num_dddict(2, 4)), ('e3_e6', g.uniform_mc_num_dddict(3, 6)),
           ('e4_e8', g.uniform_mc_num_dddict(4, 8))]
    mcmax = g.get_mc_num_dddict(g.make_mc_mask_dddict(), is_max=True)
    keys = g.make_lat_lookup_key_dddict()
    for target in (10.0, 15.0, 18.0):
        mc = g.initial_mc_num_dddict()
        for op in (1, 7, 4):
            arch = OrderedDict((st, OrderedDict((b, op) for b in mc[st])) for st in mc)
            lat = get_lookup_latency(arch, mc, keys, lut)
            mc, _ = fit_mc_num_by_latency(arch, mc, mcmax, keys, lut, target, list(mc.keys()), -1 if lat > target else 1)
        out.append(('ragged_target%d' % int(target), mc))
    return out


def dropin_pair(model, opt_w, opt_a, bw, ba, target_lat=15.0, lambda_lat=0.1, grad_clip=5.0):
    """Two iterations of the reference's train_w_arch body (train_search.py:370-426) written the way the reference writes
    them, against the drop-in model: requires_grad toggling over named_parameters, two sequential sampled forwards with
    device-side sampling, torch clip + optimizer, per-parameter log_softmax projection."""
    import torch.nn as nn
    import torch.nn.functional as F
    for step, (x_w, target_w) in enumerate(bw):
        for p in model.weight_parameters():
            p.requires_grad = True
        for p in model.arch_parameters():
            p.requires_grad = False
        logits_g, _ = model(x_w, sampling=True, mode='gumbel')
        loss_g = F.cross_entropy(logits_g, target_w)
        logits_r, _ = model(x_w, sampling=True, mode='random')
        loss_r = F.cross_entropy(logits_r, target_w)
        loss_w = loss_g + loss_r
        opt_w.zero_grad()
        loss_w.backward()
        nn.utils.clip_grad_norm_(model.weight_parameters(), grad_clip)
        opt_w.step()
        if step % 2 == 0:
            x_a, target_a = ba
            for p in model.weight_parameters():
                p.requires_grad = False
            for p in model.arch_parameters():
                p.requires_grad = True
            logits, lat = model(x_a, sampling=False)
            loss = F.cross_entropy(logits, target_a) + torch.abs(lat / target_lat - 1.) * lambda_lat
            opt_a.zero_grad()
            loss.backward()
            nn.utils.clip_grad_norm_(model.arch_parameters(), grad_clip)
            opt_a.step()
            for p in model.arch_parameters():
                p.data = F.log_softmax(p.detach().data, dim=-1)


def _require_finite(model, what):
    """A throughput number from a diverged run is worthless (NaN arithmetic runs at full speed)."""
    bad = [n for n, p in model.named_parameters() if not bool(torch.isfinite(p).all())]
    if bad:
        raise RuntimeError('bench: non-finite parameters after %s: %s ...' % (what, bad[:3]))


def retrain_leg(dev, batch=256, steps=10, warmup=3):
    """BASELINE configs[4] at its own size on one GPU: the derived-network retrain step (train_eval.py:228-252: forward,
    label-smoothed loss, backward, clip, SGD) at 224x224, 256 images per GPU, on the HIP path (tfnas_amd/model_eval.py).
    Architecture: the all-candidate-1 full-depth network scaled to the 18 ms target of SURVEY 8(c).6 (the TF-NAS-A config
    itself is not in the reference repository).  Checks: every parameter finite, and two runs from the same state are
    bit-identical (deterministic kernels)."""
    from collections import OrderedDict
    from tfnas_amd import geometry as g, model_eval as me
    from tfnas_amd.elasticity import fit_mc_num_by_latency
    from tfnas_amd.latency import load_lat_lookup
    lut = load_lat_lookup('gpu')
    mc = g.initial_mc_num_dddict()
    arch = OrderedDict((st, OrderedDict((b, 1) for b in mc[st])) for st in mc)
    mc, lat = fit_mc_num_by_latency(arch, mc, g.get_mc_num_dddict(g.make_mc_mask_dddict(), is_max=True),
                                    g.make_lat_lookup_key_dddict(), lut, 18.0, list(mc.keys()), 1)
    gen = torch.Generator(device=dev).manual_seed(77)
    x = torch.randn(batch, 3, 224, 224, device=dev, generator=gen)
    y = torch.randint(0, 1000, (batch,), device=dev, generator=gen)
    crit = me.CrossEntropyLabelSmooth(1000, 0.1)

    def fresh():
        torch.manual_seed(0)
        m = me.Network(1000, arch, mc, lut, 0.0, 0.0).to(dev)        # (no dropout / drop-connect draws: determinism check)
        return m, torch.optim.SGD(m.parameters(), 0.2, momentum=0.9, weight_decay=4e-5)
    finals = []
    for _ in range(2):
        m, opt = fresh()
        for _ in range(2):
            me.train_step(m, x, y, crit, opt, 5.0)
        torch.cuda.synchronize()
        finals.append([p.detach().clone() for p in m.parameters()])
        del m, opt
    deterministic = all(torch.equal(a, b) for a, b in zip(*finals))
    del finals
    torch.manual_seed(0)
    model = me.Network(1000, arch, mc, lut, 0.2, 0.2).to(dev)
    opt = torch.optim.SGD(model.parameters(), 0.2, momentum=0.9, weight_decay=4e-5)
    for _ in range(warmup):
        me.train_step(model, x, y, crit, opt, 5.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        me.train_step(model, x, y, crit, opt, 5.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _require_finite(model, 'the retrain leg')
    out = dict(retrain_images_per_s=round(batch * steps / dt, 1), ms_per_step=round(dt / steps * 1e3, 2), batch_per_gpu=batch,
               steps=steps, warmup=warmup, dtype='fp32', deterministic=bool(deterministic), image='224x224',
               arch='all-op-1 full depth, widths scaled to 18 ms (%.3f ms in the LUT), dropout 0.2, drop-connect 0.2' % lat,
               params_M=round(sum(p.numel() for p in model.parameters()) / 1e6, 3))
    if not deterministic:
        raise RuntimeError('bench: the retrain step is not bit-deterministic')
    # the reference's retrain script also exists with mixed precision (train_eval_amp.py:176-180,331-333: apex O1).  The same
    # step with the 1x1 GEMMs on the bf16 MFMA pipe (tfnas_set_gemm_mode(TFNAS_GEMM_BF16): bf16 operands, fp32 accumulation, fp32
    # storage / statistics / master weights) -- a SECONDARY, reduced-precision figure with its loss beside the fp32 one's
    from tfnas_amd import _lib
    lib = _lib.lib()
    prev = lib.tfnas_gemm_mode()
    try:
        res = {}
        for tag, mode in (('fp32', prev), ('bf16', 1)):
            _lib.check(lib.tfnas_set_gemm_mode(mode), 'tfnas_set_gemm_mode')
            torch.manual_seed(0)
            m2 = me.Network(1000, arch, mc, lut, 0.0, 0.0).to(dev)
            o2 = torch.optim.SGD(m2.parameters(), 0.05, momentum=0.9, weight_decay=4e-5)
            for _ in range(warmup):
                me.train_step(m2, x, y, crit, o2, 5.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss2, _ = me.train_step(m2, x, y, crit, o2, 5.0)
            torch.cuda.synchronize()
            res[tag] = (batch * steps / (time.perf_counter() - t0), float(loss2))
            _require_finite(m2, 'the retrain leg (%s GEMMs)' % tag)
            del m2, o2
        out['bf16_gemm'] = dict(images_per_s=round(res['bf16'][0], 1), vs_fp32=round(res['bf16'][0] / res['fp32'][0], 3),
                                **{'loss_after_%d_steps' % (warmup + steps): round(res['bf16'][1], 4)}, fp32_loss=round(res['fp32'][1], 4),
                                note='bf16 MFMA operands in the 1x1 GEMMs only; tensors stay fp32 in HBM (DESIGN.md section 4a)')
    finally:
        lib.tfnas_set_gemm_mode(prev)
    del model, opt
    torch.cuda.empty_cache()
    return out


def run_gpu(args):
    from tfnas_amd import Network, load_lat_lookup, geometry, search, _lib
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    launched = 'RANK' in os.environ            # under torch.distributed.run: always bring RCCL up (also at N=1)
    if (world > 1 or launched) and os.environ.get('TFNAS_BENCH_NO_PG', '0') != '1':      # (NO_PG: diagnostics only)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        dist.init_process_group('nccl', rank=rank, world_size=world)     # (NOT device_id=...: see DESIGN.md 4a, eager RCCL init)
    B = args.batch
    torch.manual_seed(2)
    model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
    model.set_temperature(5.0)
    state = search.SearchState(model)
    opt_w, opt_a = search.make_optimizers(model)
    noise = search.NoiseSource(2)                                  # same seed on every rank -> same architectures
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)      # each rank its own shard of synthetic data
    pool = 4

    def batch():
        return (torch.randn(B, 3, 224, 224, device=dev, generator=gen),
                torch.randint(0, 100, (B,), device=dev, generator=gen))
    train = [batch() for _ in range(2 * pool)]
    val = [batch() for _ in range(pool)]

    def pair(i):
        search.search_iteration_pair(state, opt_w, opt_a, (train[(2 * i) % len(train)], train[(2 * i + 1) % len(train)]),
                                     val[i % len(val)], noise)

    lib = _lib.lib()
    nfam = lib.tfnas_prof_count()
    names = [lib.tfnas_prof_name(i).decode() for i in range(nfam)]

    def collect():
        out = {}
        for i in range(nfam):
            cnt, ms = C.c_uint64(0), C.c_double(0.0)
            _lib.check(lib.tfnas_prof_collect(i, C.byref(cnt), C.byref(ms)), 'tfnas_prof_collect')
            out[names[i]] = (cnt.value, ms.value)
        return out

    for i in range(max(args.warmup - 1, 0)):
        pair(i)
    # last warmup pair: time every kernel family to find the dominant one
    lib.tfnas_prof_enable((1 << nfam) - 1)
    pair(args.warmup)
    torch.cuda.synchronize()
    fam_ms = collect()
    lib.tfnas_prof_enable(0)
    dominant = max(fam_ms, key=lambda k: fam_ms[k][1])
    lib.tfnas_prof_enable(1 << names.index(dominant))               # only the dominant family in the timed region

    def barrier():
        if dist.is_initialized():
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    barrier()
    ar0 = (search.ALLREDUCE_CALLS, search.ALLREDUCE_BYTES)
    t0 = time.perf_counter()
    for i in range(args.steps):
        pair(args.warmup + 1 + i)
    barrier()
    dt = time.perf_counter() - t0
    ar1 = (search.ALLREDUCE_CALLS, search.ALLREDUCE_BYTES)
    if dist.is_initialized():
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    dom = collect()[dominant]
    dom_soft = (C.c_uint64(0), C.c_double(0.0))
    _lib.check(lib.tfnas_prof_last_split(names.index(dominant), C.byref(dom_soft[0]), C.byref(dom_soft[1])), 'tfnas_prof_last_split')
    dom_soft = (dom_soft[0].value, dom_soft[1].value)
    lib.tfnas_prof_enable(0)
    _require_finite(model, 'the timed region')

    # ---- after the timed region: GPU time of the w-step and the alpha-step (HIP events on the launch stream; the
    # w-step's side streams fork from and join it) over a few more pairs
    nsplit = max(1, min(10, args.steps))
    evs = []
    for i in range(nsplit):
        j = args.warmup + 1 + args.steps + i
        bw = (train[(2 * j) % len(train)], train[(2 * j + 1) % len(train)])
        ba = val[j % len(val)]
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        search.w_step(state, bw[0][0], bw[0][1], opt_w, 5.0, noise.exp(dev), noise.rand_pos())
        e[1].record()
        search.a_step(state, ba[0], ba[1], opt_a, 15.0, 0.1, 5.0, noise.exp(dev))
        e[2].record()
        search.w_step(state, bw[1][0], bw[1][1], opt_w, 5.0, noise.exp(dev), noise.rand_pos())
        e[3].record()
        evs.append(e)
    torch.cuda.synchronize()
    w_ms = sum(e[0].elapsed_time(e[1]) + e[2].elapsed_time(e[3]) for e in evs) / (2 * nsplit)
    a_ms = sum(e[1].elapsed_time(e[2]) for e in evs) / nsplit
    # multi-GPU self-evidence: how many ranks the first collective really spanned, and every rank's own
    # step times (a straggler or a rank on a colliding hardware queue shows up as spread here, not only in the max-over-ranks time)
    dist_info = None
    if dist.is_initialized():
        mine = torch.tensor([w_ms, a_ms, float(rank)], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        rows = sorted((int(t[2]), float(t[0]), float(t[1])) for t in allr)
        dist_info = dict(backend=dist.get_backend(), rccl_ranks=dist.get_world_size(), ranks_seen=[r for r, _, _ in rows],
                         w_step_ms_per_rank=[round(w, 3) for _, w, _ in rows], a_step_ms_per_rank=[round(a, 3) for _, _, a in rows],
                         w_step_ms_spread=round(max(w for _, w, _ in rows) - min(w for _, w, _ in rows), 3),
                         w_step_ms_min_max=[round(min(w for _, w, _ in rows), 3), round(max(w for _, w, _ in rows), 3)],
                         a_step_ms_min_max=[round(min(a for _, _, a in rows), 3), round(max(a for _, _, a in rows), 3)],
                         # the gradient exchange of this rank inside the timed region: collectives and payload per iteration pair
                         # (two weight steps' sampled-path gradients, ~35 MB each, + the 162 architecture scalars)
                         allreduce_calls_per_pair=round((ar1[0] - ar0[0]) / max(1, args.steps), 2),
                         allreduce_bytes_per_pair=int((ar1[1] - ar0[1]) / max(1, args.steps)),
                         gpu_max_hw_queues=os.environ.get('GPU_MAX_HW_QUEUES'))
        # replica consistency (tools/dp_check.py's hash): every rank hashes all of its parameters after the
        # timed region; data-parallel replicas must be bit-identical
        import hashlib
        h = hashlib.sha256()
        for k, p in model.named_parameters():
            h.update(k.encode())
            h.update(p.detach().cpu().numpy().tobytes())
        mine_h = torch.tensor([int(h.hexdigest()[i:i + 8], 16) for i in range(0, 32, 8)], device=dev, dtype=torch.int64)
        all_h = [torch.zeros_like(mine_h) for _ in range(dist.get_world_size())]
        dist.all_gather(all_h, mine_h)
        dist_info['replica_sha256_128'] = h.hexdigest()[:32]
        dist_info['replicas_bit_identical'] = bool(all(bool((t == all_h[0]).all()) for t in all_h))

    # ---- the reference-style loop on the drop-in model (world 1 only: it has no gradient all-reduce)
    dropin = None
    if world == 1 and not args.no_dropin:
        import random
        random.seed(2)
        model.reset_switches()
        nd, wd_ = max(1, min(10, args.steps)), 2
        for i in range(wd_ + nd):
            if i == wd_:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            dropin_pair(model, opt_w, opt_a, (train[(2 * i) % len(train)], train[(2 * i + 1) % len(train)]), val[i % len(val)])
        torch.cuda.synchronize()
        dropin = 2.0 * B * nd / (time.perf_counter() - t0)

    # ---- BASELINE configs[3] / SURVEY 8(d) C4: width sweep (uniform expand ratios over the reachable range + the ragged widths
    # elasticity scaling produces for three latency targets), latency lookup inside every soft forward; a few pairs each
    sweep = None
    if world == 1 and args.width_sweep:
        sweep = []
        del state, opt_w, opt_a, model
        torch.cuda.empty_cache()
        for name, mc in width_sweep_configs():
            torch.manual_seed(2)
            m2 = Network(100, mc, load_lat_lookup('gpu')).to(dev)
            m2.set_temperature(5.0)
            st2 = search.SearchState(m2)
            ow2, oa2 = search.make_optimizers(m2)
            for i in range(3 + 6):
                if i == 3:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                search.search_iteration_pair(st2, ow2, oa2, (train[(2 * i) % len(train)], train[(2 * i + 1) % len(train)]),
                                             val[i % len(val)], noise)
            torch.cuda.synchronize()
            sweep.append(dict(widths=name, images_per_s=round(2.0 * B * 6 / (time.perf_counter() - t0), 1)))
            _require_finite(m2, 'the width sweep (%s)' % name)
            if st2.runner is not None:
                st2.runner.close()
            del st2, ow2, oa2, m2
            torch.cuda.empty_cache()

    # (rounds 1-3 printed a secondary `bf16` line here: bf16 storage of E / D / dZ / dEh measured 0.99-1.04x of this fp32 number,
    #  the step is not bound by HBM bytes, and the mode was removed -- DESIGN.md section 7)

    retrain = None
    if world == 1 and args.retrain:
        retrain = retrain_leg(dev)

    result = None
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = 2.0 * B * world * args.steps / dt
        alg = family_algorithmic(dominant, B)
        roof = None
        if alg is not None and dom[0] > 0:
            fl, by, nl = alg
            avg_ms = dom[1] / dom[0]
            tf = fl / nl / (avg_ms * 1e-3) / 1e12
            gbs = by / nl / (avg_ms * 1e-3) / 1e9
            if tf / PEAK_FP32_MFMA_TF >= gbs / PEAK_HBM_GBS:
                roof = dict(bound='mfma', achieved=round(tf, 2), peak=PEAK_FP32_MFMA_TF, unit='TFLOP/s',
                            frac=round(tf / PEAK_FP32_MFMA_TF, 4))
            else:
                roof = dict(bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                            frac=round(gbs / PEAK_HBM_GBS, 4))
            if roof['frac'] > 1.0:
                raise SystemExit('bench.py: byte/flop model of %s gives frac %.2f > 1 -- the model is wrong' % (dominant, roof['frac']))
            # the same kernels at two sizes: alpha-step launches (all 8 candidates of a cell, one queue) and w-step launches (one
            # candidate, four queues side by side) -- each against its own algorithmic bytes / flops
            by_mode = {}
            for mname, (cnt_m, ms_m) in (('soft', dom_soft), ('sampled', (dom[0] - dom_soft[0], dom[1] - dom_soft[1]))):
                am = family_algorithmic(dominant, B, mname)
                if am is None or cnt_m <= 0 or am[2] <= 0:
                    continue
                a_ms_ = ms_m / cnt_m
                tf_m, gb_m = am[0] / am[2] / (a_ms_ * 1e-3) / 1e12, am[1] / am[2] / (a_ms_ * 1e-3) / 1e9
                by_mode[mname] = dict(launches_timed=cnt_m, avg_launch_ms=round(a_ms_, 4), gbs=round(gb_m, 1), tflops=round(tf_m, 2),
                                      frac=round(max(tf_m / PEAK_FP32_MFMA_TF, gb_m / PEAK_HBM_GBS), 4),
                                      alg_bytes_per_launch=am[1] / am[2])
            if dom_soft[0] > 0 and by_mode:
                roof['by_mode'] = by_mode
            roof.update(kernel=dominant, avg_launch_ms=round(avg_ms, 4), launches_timed=dom[0],
                        alg_flops_per_launch=fl / nl, alg_bytes_per_launch=by / nl, traffic=None,
                        share_of_step=round(dom[1] / args.steps / ms_per_step, 3))
        # whole pair against SURVEY 8(d)'s algorithmic floor: F0 bytes at 8 TB/s, 1x1 flops on the fp32 MFMA peak,
        # depthwise + SE flops on the fp32 vector peak (same 157.3 TF/s)
        pa = pair_algorithmic(B)
        t = ms_per_step * 1e-3
        t_hbm, t_mfma, t_valu = pa['bytes'] / (PEAK_HBM_GBS * 1e9), pa['flops_1x1'] / (PEAK_FP32_MFMA_TF * 1e12), \
            pa['flops_other'] / (PEAK_FP32_MFMA_TF * 1e12)
        t_lower = max(t_hbm, t_mfma, t_valu)
        pair_roof = dict(t_lower_ms=round(t_lower * 1e3, 3), achieved=round(t_lower / t, 4),
                         hbm_fraction=round(t_hbm / t, 4), mfma_fraction=round(t_mfma / t, 4),
                         f0_bytes=pa['bytes'], flops_1x1=pa['flops_1x1'], flops_dw_se=pa['flops_other'],
                         note='SURVEY.md 8(d): t_lower = max(F0/8 TB/s, 1x1 flops/157.3 TF/s fp32 MFMA, dw+SE flops/157.3 TF/s)')
        if roof is not None:
            roof['pair'] = pair_roof
        else:
            roof = dict(pair=pair_roof)
        result = dict(metric='supernet search images/sec (w-step + alpha-step)', value=round(value, 2), unit='images/s',
                      n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(ms_per_step, 3),
                      higher_is_better=True, scaling='weak', vs_baseline=None, dtype='fp32', data='synthetic',
                      config=dict(workload='TF-NAS supernet search iteration pair (w-step, alpha-step, w-step; '
                                           'train_search.py:366-426) on ImageNet-100-shaped 224x224 batches, '
                                           'initial widths, T=5, target_lat=15 (BASELINE configs[1] geometry, fp32)',
                                  batch_per_gpu=B, global_batch=B * world, train_images_per_step=2 * B * world,
                                  parallelism='dp%d' % world,
                                  gemm_arithmetic={0: 'fp32 MFMA', 1: 'bf16 MFMA (reduced precision)', 3: 'split-bf16 x2',
                                                   6: 'split-bf16 x3 MFMA: every fp32 operand as three bf16 planes, fp32 '
                                                      'accumulation (fp32-accurate products; the oracle tolerance is unchanged)'
                                                   }.get(int(lib.tfnas_gemm_mode()) & 0xff, str(lib.tfnas_gemm_mode()))),
                      roofline=roof, w_step_ms=round(w_ms, 3), a_step_ms=round(a_ms, 3),
                      all_images_per_s=round(3.0 * B * world * args.steps / dt, 2),
                      dropin_images_per_s=None if dropin is None else round(dropin, 2), width_sweep=sweep,
                      retrain=retrain, dist=dist_info,
                      kernel_ms_per_pair={k: round(v[1], 3) for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1][1])
                                          if v[0]},
                      kernel_roofline=kernel_roofline(fam_ms, B))
    if dist.is_initialized():
        dist.destroy_process_group()
    return result


def run_cpu_baseline(leg, B=32):
    """Time the CPU oracle on a bounded sample of the same loop at the reference's batch 32 (SURVEY 8(d) "CPU baseline beside
    it").  Three legs, each run in its own child process with a hard wall-clock limit (cpu_baseline_subprocess):
      best     thread count probed on a short step (8/16/32/64): oneDNN scales badly past one socket on big hosts; this
               is the headline `value`: 3 warm-up + 10 timed iteration pairs;
      t8       pinned to 8 threads (comparable with the 8-vCPU build container, SURVEY 6: ~8.5 img/s): 1 warm-up + 3 pairs;
      allcores torch.set_num_threads(os.cpu_count()), 1 warm-up + 3 pairs, killed after 45 s (256 threads made ONE pair
               take minutes on the GPU box in round 1) -- reported as null when it does not finish."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import tfnas_oracle as orc
    from tfnas_amd.latency import load_lat_lookup
    from tfnas_amd import search
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    torch.manual_seed(2)
    model = orc.Network(100, orc.initial_mc_num_dddict(), load_lat_lookup('gpu'))
    model.set_temperature(5.0)
    opt_w, opt_a = orc.make_optimizers(model)
    noise = search.NoiseSource(2)
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(B, 3, 224, 224, generator=g) for _ in range(3)]
    ys = [torch.randint(0, 100, (B,), generator=g) for _ in range(3)]

    def probe():
        t0 = time.perf_counter()
        orc.w_step(model, xs[0], ys[0], opt_w, 5.0, noise.exp('cpu'), noise.rand_pos(), bi_sampling=False)
        return time.perf_counter() - t0

    def pair():
        orc.w_step(model, xs[0], ys[0], opt_w, 5.0, noise.exp('cpu'), noise.rand_pos())
        orc.a_step(model, xs[1], ys[1], opt_a, 15.0, 0.1, 5.0, noise.exp('cpu'))
        orc.w_step(model, xs[2], ys[2], opt_w, 5.0, noise.exp('cpu'), noise.rand_pos())

    if leg == 'best':
        best_t, threads = None, 1
        for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(nt)
            probe()                                  # warm-up at this thread count
            t = probe()
            if best_t is None or t < best_t:
                best_t, threads = t, nt
        warm, n = 3, 10
    else:
        threads, warm, n = (min(8, ncpu) if leg == 't8' else ncpu), 1, 3
    torch.set_num_threads(threads)
    for _ in range(warm):
        pair()
    t0 = time.perf_counter()
    for _ in range(n):
        pair()
    dt = (time.perf_counter() - t0) / n
    return dict(threads=threads, pairs=n, warmup=warm, value=round(2 * B / dt, 3), ms_per_pair=round(dt * 1e3, 1),
                host_threads=ncpu, torch=torch.__version__, batch=B)


def cpu_baseline_subprocess():
    """Run the CPU legs in child processes with hard wall-clock limits so the bench line is always printed."""
    import subprocess

    def run(leg, timeout):
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', leg],
                                 capture_output=True, text=True, timeout=timeout,
                                 env=dict(os.environ, CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES=''))
            for line in reversed(out.stdout.strip().splitlines()):
                if line.startswith('{'):
                    return json.loads(line)
            return dict(value=None, note='failed: ' + out.stderr[-200:])
        except subprocess.TimeoutExpired:
            return dict(value=None, note='did not finish within %d s' % timeout)
    best = run('best', 150)
    t8 = run('t8', 90)
    allc = run('allcores', 45)
    if best.get('value') is None:
        return dict(value=None, unit='images/s', cores=0, kind='port', sample='cpu baseline: ' + best.get('note', ''),
                    threads8=t8, all_cores=allc)
    return dict(value=best['value'], unit='images/s', cores=best['threads'], kind='port',
                sample='%d iteration pairs (w-step, alpha-step, w-step) of oracle/tfnas_oracle.py at batch %d fp32 after %d '
                       'warm-up pairs; torch %s, %d threads (best of 8/16/32/64 probed) on a %d-thread host'
                       % (best['pairs'], best['batch'], best['warmup'], best['torch'], best['threads'], best['host_threads']),
                ms_per_step=best['ms_per_pair'], threads8=t8, all_cores=allc)


def _attach_pmc_traffic(res):
    """roofline.traffic = HBM bytes per launch of the dominant kernel family from the committed rocprofv3 PMC passes
    (profiles/round1_pmc_summary.json: separate FETCH_SIZE / WRITE_SIZE runs of this same command, FETCH_SIZE
    doubled per MI355X_MICROARCH.md's gfx950 correction); null when the profile has no entry for that family."""
    roof = res.get('roofline')
    cands = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_pmc_summary.json')) \
        if os.path.isdir(os.path.join(ROOT, 'profiles')) else []
    if not roof or not cands or 'kernel' not in roof:
        return
    path = os.path.join(ROOT, 'profiles', cands[-1])            # the latest round's PMC passes
    try:
        pmc = json.load(open(path))
        ent = pmc['families'].get(roof['kernel'])
        if ent and pmc.get('batch_per_gpu') == res['config']['batch_per_gpu']:
            roof['traffic'] = ent['hbm_bytes_per_launch']
            roof['traffic_source'] = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)' % cands[-1]
    except Exception:
        pass


def visible_devices():
    """GPUs this process can use (TFNAS_FAKE_DEVICES overrides it: tests of the launcher on CPU-only machines)."""
    fake = os.environ.get('TFNAS_FAKE_DEVICES')
    return int(fake) if fake is not None else torch.cuda.device_count()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher (the driver's command line; the reference's counterpart is
    the nn.DataParallel wrap of train_search.py:95,158): re-exec this script under `python -m torch.distributed.run` with one
    rank per GPU; rank 0 of the child job prints the one JSON line on the inherited stdout.  Fewer than N devices: ONE JSON line
    carrying "error", exit status 2."""
    n = args.gpus
    have = visible_devices()
    if have < n:
        print(json.dumps(dict(metric='supernet search images/sec (w-step + alpha-step)', value=None, unit='images/s', n_gpus=n,
                              error='bench.py --gpus %d: only %d device(s) visible on this node' % (n, have), devices_visible=have)))
        return 2
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')          # dmabuf IPC (RCCL across processes on this driver)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_rank():
    """TFNAS_BENCH_DRY=1 (tests/test_bench_launch.py): a rank of the self-launched job only proves the rendezvous -- gloo process
    group, one all-reduce -- and rank 0 prints a JSON line; no GPU is touched."""
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', rank=rank, world_size=world)
    t = torch.ones(1)
    dist.all_reduce(t)
    # the shape of the real line's `dist` object: one row per rank, gathered (here: rank ids only)
    mine = torch.tensor([float(rank)])
    rows = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(rows, mine)
    if rank == 0:
        print(json.dumps(dict(dry_run=True, n_gpus=world, ranks_seen=int(t.item()), rank_ids=sorted(int(r.item()) for r in rows),
                              argv=sys.argv[1:])), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50, help='timed iteration pairs (SURVEY 8(d): >= 50)')
    ap.add_argument('--warmup', type=int, default=10, help='untimed warm-up pairs (>= 10)')
    ap.add_argument('--no-dropin', action='store_true', help='skip the reference-style drop-in loop timing')
    ap.add_argument('--no-bf16', dest='bf16', action='store_false', help='(accepted and ignored: the bf16-storage mode was removed in round 3)')
    ap.add_argument('--no-width-sweep', dest='width_sweep', action='store_false',
                    help='skip the BASELINE configs[3] width sweep (6 widths x 6 pairs after the timed region)')
    ap.add_argument('--no-retrain', dest='retrain', action='store_false',
                    help='skip the BASELINE configs[4] leg (derived-network retrain step, 256 images at 224x224)')
    ap.add_argument('--batch', type=int, default=128, help='images per GPU per step-half (BASELINE configs[1]: 128)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(run_cpu_baseline(args.cpu_baseline_only)))
        return
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(self_launch(args))              # `python bench.py --gpus N`: bring up the N ranks ourselves
    if os.environ.get('TFNAS_BENCH_DRY') == '1':
        dry_rank()
        return
    res = run_gpu(args)
    if res is not None:
        if args.gpus == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline_subprocess()
        _attach_pmc_traffic(res)
        print(json.dumps(res))


if __name__ == '__main__':
    main()
