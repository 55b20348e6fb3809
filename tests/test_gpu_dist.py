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
