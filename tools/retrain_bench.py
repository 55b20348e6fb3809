#!/usr/bin/env python3
"""Throughput of the derived-network retrain step (BASELINE configs[4]: 224x224, batch 256 per GPU) on the HIP path
(tfnas_amd/model_eval.py): forward + label-smoothed loss + backward + clip + SGD, synthetic data.
  python tools/retrain_bench.py [--batch 256] [--steps 20]           (one rank per GPU under torch.distributed.run for N > 1)
The architecture is the all-candidate-1 (k3 e6) full-depth network scaled to the 18 ms target of SURVEY 8(c).6 -- the TF-NAS-A
config itself is not in the reference repository."""
import argparse
import json
import os
import sys
import time
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    args = ap.parse_args()
    from tfnas_amd import geometry as g, model_eval as me
    from tfnas_amd.elasticity import fit_mc_num_by_latency
    from tfnas_amd.latency import load_lat_lookup
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (('RANK', '0'), ('WORLD_SIZE', '1'), ('LOCAL_RANK', '0')))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29571')
        dist.init_process_group('nccl', rank=rank, world_size=world)     # (NOT device_id=...: see DESIGN.md 4a, eager RCCL init)
    lut = load_lat_lookup('gpu')
    mc = g.initial_mc_num_dddict()
    arch = OrderedDict((st, OrderedDict((b, 1) for b in mc[st])) for st in mc)
    mc, lat = fit_mc_num_by_latency(arch, mc, g.get_mc_num_dddict(g.make_mc_mask_dddict(), is_max=True),
                                    g.make_lat_lookup_key_dddict(), lut, 18.0, list(mc.keys()), 1)
    torch.manual_seed(0)
    model = me.Network(1000, arch, mc, lut, 0.2, 0.2).to(dev)
    opt = torch.optim.SGD(model.parameters(), 0.2, momentum=0.9, weight_decay=4e-5)
    crit = me.CrossEntropyLabelSmooth(1000, 0.1)
    x = torch.randn(args.batch, 3, 224, 224, device=dev)
    y = torch.randint(0, 1000, (args.batch,), device=dev)
    for _ in range(args.warmup):
        me.train_step(model, x, y, crit, opt, 5.0)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier(device_ids=[local])
    t0 = time.perf_counter()
    for _ in range(args.steps):
        me.train_step(model, x, y, crit, opt, 5.0)
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier(device_ids=[local])
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps(dict(metric='derived-network retrain images/sec', value=round(args.batch * world * args.steps / dt, 1),
                              ms_per_step=round(dt / args.steps * 1e3, 2), n_gpus=world, batch_per_gpu=args.batch, dtype='fp32',
                              arch='all-op-1 full depth, widths scaled to 18 ms (%.3f ms in the LUT)' % lat,
                              params_MB=round(sum(p.numel() for p in model.parameters()) / 1e6, 3))))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
