"""Child process of tests/test_00_two_ranks_gpu.py: one of two ranks sharing cuda:0 (gloo rendezvous on 127.0.0.1).
The real N > 1 sequence of SURVEY 8e on the HIP path: rank 0 packs the weights, ONE broadcast of the blob, every other rank
binds what it received, each rank decodes its shard of the utterances, the outputs are gathered; rank 0 compares the gathered
result bit for bit with its own unsharded HIP decode and writes the verdict as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
    import torch
    import torch.distributed as dist

    import torch_tts_amd as T
    from oracle import tacotron_oracle as O
    from torch_tts_amd import _lib
    from torch_tts_amd import distributed as D

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = {}
    try:
        dims = O.DecoderDims()  # LJSpeech dims
        L, Tn = 40, 16
        # whole batch and shards must lie in one regime of the launch schedule (test_shard_equivalence_bitwise): split-fp16
        # 65 .. 256 utterances, exact fp32 more than 320
        batch = {"split_f16": 140, "f32": 768}
        # rank 0 owns the real weights; the other rank starts from DIFFERENT ones, so only the broadcast blob can make it agree
        wts = O.random_decoder_weights(dims, seed=42 if rank == 0 else 999, nonzero_init_state=True)
        cell = T.Taco2ProdDecoderCell(dims.d_ctx, dims.d_mel, 1, [dims.h_att, dims.h_dec], dim_pre=dims.d_pre, dim_att=dims.h_att)
        dec = T.Decoder(cell, 1, dims.d_mel)
        dec.load_state_dict(wts, strict=False)
        dec = dec.to(dev).eval()
        per_prec = {}
        for prec in ("split_f16", "f32"):
            B = batch[prec]
            mem = O.synthetic_memory(B, L, dims.d_ctx, seed=5).to(dev)
            masks = O.synthetic_masks(Tn, B, dims.d_pre, seed=6).to(dev)
            eng = dec._engines.get(dec.decoder_cell.engine_dims(), dev)
            eng.set_precision(prec)
            D.broadcast_engine_weights(eng, dec.weight_tensors(), src=0)  # pack on 0, broadcast, bind elsewhere
            lo, hi = D.shard_bounds(B, world, rank)

            def run(m, k):
                n = m.shape[0]
                y = torch.empty(n, Tn, dims.d_mel, device=dev); s = torch.empty(n, Tn, device=dev); w = torch.empty(n, Tn, L, device=dev)
                t_out = torch.zeros(2, dtype=torch.int32, device=dev)
                eng.decode(m.contiguous(), t_begin=0, n_steps=Tn, stop_threshold=-2.0, check_stop=True, dropout_mode=_lib.DROPOUT_MASKS,
                           masks=k.contiguous(), seed=0, teacher=None, teacher_flags=None, y=y, s=s, w=w, t_out=t_out)
                assert t_out.tolist() == [Tn, 0], t_out.tolist()
                return y, s, w

            y, s, w = run(mem[lo:hi], masks[:, :, lo:hi])
            sizes = [D.shard_bounds(B, world, r)[1] - D.shard_bounds(B, world, r)[0] for r in range(world)]
            y_all, s_all, w_all = (D.gather_outputs(t, sizes) for t in (y, s, w))
            if rank == 0:
                ry, rs, rw = run(mem, masks)  # the unsharded decode on the packing handle
                per_prec[prec] = {"y": bool(torch.equal(y_all, ry)), "s": bool(torch.equal(s_all, rs)), "w": bool(torch.equal(w_all, rw)),
                                  "sizes": sizes, "finite": bool(torch.isfinite(ry).all()), "precision": eng.precision()}
        res = per_prec
    finally:
        dist.destroy_process_group()
    if rank == 0:
        with open(out_path, "w") as f:
            json.dump(res, f)


if __name__ == "__main__":
    main()
