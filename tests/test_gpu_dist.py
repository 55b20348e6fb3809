"""The HIP path under torch.distributed.run with RCCL up (1 rank -- the GPU box has one MI355X): the gradient all-reduce
path (TFNAS_FORCE_ALLREDUCE=1 takes it at world_size 1) plus RCCL's own streams must not change a single bit of the
search trajectory.  The N>1 logic is covered on CPU by tests/test_dp_gloo.py; tools/launch_scale.sh runs this same
script at 2/4/8 ranks on a multi-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_one_rank_torchrun_with_rccl_allreduce_is_bit_identical_to_plain_run(tmp_path):
    script = os.path.join(ROOT, 'tools', 'dp_check.py')
    plain, dist_ = str(tmp_path / 'plain.json'), str(tmp_path / 'dist.json')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    subprocess.run([sys.executable, script, '--out', plain], check=True, env=env, timeout=400)
    env2 = dict(env, TFNAS_FORCE_ALLREDUCE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1',
                    '--master-addr', '127.0.0.1', '--master-port', '29547', script, '--out', dist_],
                   check=True, env=env2, timeout=400)
    a, b = json.load(open(plain)), json.load(open(dist_))
    assert not a['rccl'] and a['allreduce_calls'] == 0
    assert b['rccl'] and b['allreduce_forced'] and b['allreduce_calls'] > 0      # the RCCL path really ran
    assert a['sha256'] == b['sha256']


def _two_rank_worker(rank, world, port, outdir):
    import hashlib
    import torch
    import torch.distributed as dist
    for p in (os.path.join(ROOT, 'tf-nas_amd'),):
        sys.path.insert(0, p)
    from tfnas_amd import Network, load_lat_lookup, geometry, search
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)      # gloo moves CUDA tensors too: 2 ranks on ONE GPU
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    torch.manual_seed(2)
    model = Network(100, geometry.initial_mc_num_dddict(), load_lat_lookup('gpu')).to(dev)
    model.set_temperature(5.0)
    state = search.SearchState(model)
    opt_w, opt_a = search.make_optimizers(model)
    noise = search.NoiseSource(2)                                      # same seed on every rank
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)         # different data shard per rank

    def batch():
        return (torch.randn(4, 3, 224, 224, device=dev, generator=gen), torch.randint(0, 100, (4,), device=dev, generator=gen))
    for _ in range(2):
        search.search_iteration_pair(state, opt_w, opt_a, (batch(), batch()), batch(), noise)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k, p in model.named_parameters():
        h.update(p.detach().cpu().numpy().tobytes())
    with open(os.path.join(outdir, 'r%d.json' % rank), 'w') as f:
        json.dump(dict(sha=h.hexdigest(), calls=search.ALLREDUCE_CALLS, overlap=state._comm_stream is not None), f)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_on_one_gpu_stay_bit_identical(tmp_path):
    """world_size 2 through the WHOLE data-parallel HIP path (packed two-region message, overlapped all-reduce of the late
    stages, 1/world folded into the fused SGD, arch-hash check, alpha-step reduce): both ranks run on the single GPU of the box
    with the gloo backend (it reduces CUDA tensors), each on its own data shard.  Replicas must end bit-identical -- and differ
    from a 1-rank run of shard 0, i.e. the other rank's gradients really arrived."""
    import torch.multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / ('r%d.json' % r))) for r in (0, 1))
    assert r0['sha'] == r1['sha']
    assert r0['calls'] >= 2 * (2 * 2 + 1) and r0['overlap']            # two regions per w-step, one message per alpha-step
    one = tmp_path / 'one'
    one.mkdir()
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_two_rank_worker, args=(1, port, str(one)), nprocs=1, join=True)
    assert json.load(open(one / 'r0.json'))['sha'] != r0['sha']
