"""N = 2 on the GPU box: two fresh child processes - started before THIS process has touched the
GPU, hence the file name that sorts first - share cuda:0 over a gloo rendezvous and run the real multi-GPU sequence of the HIP
path (tests/two_rank_worker.py): pack on rank 0 -> one broadcast of the weight blob -> bind on rank 1 -> each rank decodes its
shard -> gather -> bit-for-bit equal to the unsharded HIP decode, in both arithmetic modes.  An 8-GPU node runs exactly this
with backend "nccl" and one device per rank (bench.py --gpus N)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_broadcast_bind_decode_gather_equals_unsharded(tmp_path):
    import torch

    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: the child ranks must be started before that (run the whole -m gpu suite, or this file alone)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    out = str(tmp_path / "two_ranks.json")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "two_rank_worker.py"), str(r), "2", port, out], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            o += "\n[killed: timeout]"
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n=====\n".join(logs)
    res = json.load(open(out))
    assert set(res) == {"split_f16", "f32"}, res
    for prec, r in res.items():
        n = {"split_f16": 70, "f32": 384}[prec]
        assert r == {"y": True, "s": True, "w": True, "sizes": [n, n], "finite": True, "precision": prec}, (prec, r)
