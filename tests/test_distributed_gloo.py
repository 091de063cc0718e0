"""N > 1 host path on CPU: two gloo ranks shard a batch of utterances, broadcast the
packed weight blob from rank 0, decode their shards (the oracle stands in for the GPU
kernels here), and the gathered result equals the unsharded decode (to 1e-6: the CPU BLAS
blocks by batch size; bit-for-bit shard equivalence of the HIP path itself is checked on the
GPU in test_hip_parity.py::test_shard_equivalence_bitwise)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tacotron_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch_tts_amd import distributed as D

    try:
        dims = O.DecoderDims(d_mel=8, d_pre=16, d_ctx=32, h_att=32, h_dec=40)
        B, L, T = 7, 11, 9
        # rank 0 owns the weights: serialise them into one byte blob, broadcast, rebuild elsewhere
        wts = O.random_decoder_weights(dims, seed=5) if rank == 0 else None
        keys = sorted(O.random_decoder_weights(dims, seed=0).keys())
        shapes = {k: tuple(v.shape) for k, v in O.random_decoder_weights(dims, seed=0).items()}
        nbytes = 4 * sum(int(torch.tensor(shapes[k]).prod()) for k in keys)
        blob = torch.cat([wts[k].reshape(-1) for k in keys]).view(torch.uint8) if rank == 0 else None
        got = D.broadcast_blob(blob, nbytes, "cpu", src=0)
        flat = got.view(torch.float32)
        rebuilt, off = {}, 0
        for k in keys:
            n = int(torch.tensor(shapes[k]).prod())
            rebuilt[k] = flat[off : off + n].reshape(shapes[k]).clone()
            off += n
        mem = O.synthetic_memory(B, L, dims.d_ctx, lengths=[11, 9, 11, 3, 11, 6, 1], seed=3)
        masks = O.synthetic_masks(T, B, dims.d_pre, seed=4)
        lo, hi = D.shard_bounds(B, world, rank)
        assert D.shard_batch(mem, world, rank).shape[0] == hi - lo
        y, s, w = O.decode(rebuilt, dims, mem[lo:hi], max_steps=T - 1, masks=masks[:, :, lo:hi])
        sizes = [D.shard_bounds(B, world, r)[1] - D.shard_bounds(B, world, r)[0] for r in range(world)]
        y_all = D.gather_outputs(y, sizes)
        w_all = D.gather_outputs(w, sizes)
        # batch-global stop semantics across shards: the first shard to stop stops everyone
        steps = D.global_stop_step(5 if rank == 0 else 8, "cpu")
        if rank == 0:
            ry, rs, rw = O.decode(wts, dims, mem, max_steps=T - 1, masks=masks)
            ok = lambda a, b: a.shape == b.shape and float((a - b).abs().max()) <= 1e-6
            out.put({"y": ok(y_all, ry), "w": ok(w_all, rw), "steps": steps, "sizes": sizes})
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_broadcast_gather_matches_unsharded():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = q.get()
    assert res == {"y": True, "w": True, "steps": 5, "sizes": [4, 3]}


def test_shard_bounds_cover_the_batch_exactly():
    from torch_tts_amd.distributed import shard_bounds

    for n in (1, 7, 8, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_bounds(2048, 8, 3) == (768, 1024)
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)
