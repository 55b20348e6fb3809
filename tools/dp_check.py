#!/usr/bin/env python3
"""Run a few search iteration pairs of the HIP supernet and write a checksum of every parameter.

Used by tests/test_gpu_dist.py: the same script is run (a) as a plain process and (b) under
`python -m torch.distributed.run --nproc-per-node 1` with RCCL initialised and TFNAS_FORCE_ALLREDUCE=1 (the gradient
all-reduce path is then taken at world_size 1); the two parameter checksums must be bit-identical.  With N ranks
(tools/launch_scale.sh) every rank writes its own file and the replicas must agree.

  python tools/dp_check.py --out /tmp/plain.json [--pairs 2] [--batch 8]
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tf-nas_amd'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--pairs', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--same-data', action='store_true', help='every rank gets the same batch (replica check)')
    args = ap.parse_args()
    from tfnas_amd import Network, load_lat_lookup, geometry, search
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if 'RANK' in os.environ:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29541')
        dist.init_process_group('nccl', rank=rank, world_size=world)     # (NOT device_id=...: see DESIGN.md 4a, eager RCCL init)
    torch.manual_seed(2)
    model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
    model.set_temperature(5.0)
    state = search.SearchState(model)
    opt_w, opt_a = search.make_optimizers(model)
    noise = search.NoiseSource(2)
    gen = torch.Generator(device=dev).manual_seed(1000 + (0 if args.same_data else rank))

    def batch():
        return (torch.randn(args.batch, 3, 224, 224, device=dev, generator=gen),
                torch.randint(0, 100, (args.batch,), device=dev, generator=gen))
    for _ in range(args.pairs):
        search.search_iteration_pair(state, opt_w, opt_a, (batch(), batch()), batch(), noise)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k, p in model.named_parameters():
        h.update(k.encode())
        h.update(p.detach().cpu().numpy().tobytes())
    out = dict(rank=rank, world=world, sha256=h.hexdigest(), allreduce_forced=search.FORCE_ALLREDUCE_AT_WORLD_1,
               rccl=dist.is_initialized(), allreduce_calls=search.ALLREDUCE_CALLS)
    path = args.out if world == 1 else args.out.replace('.json', '.r%d.json' % rank)
    with open(path, 'w') as f:
        json.dump(out, f)
    if dist.is_initialized():
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
