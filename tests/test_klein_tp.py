"""World-size-2 gloo test (CPU) of the DiT tensor-parallel shard plan (SURVEY.md section 8e row 3;
ominix-mlx_amd/klein.py `shard_state_dict`): every rank runs the CPU restatement of
FluxKlein::forward_with_rope (oracle/ref_klein.py) on ITS shards -- its heads, its MLP columns, its rows
of the fused single-block projection -- and all-reduces the partial outputs of the row-split
projections exactly where the device engine calls RCCL.  The result must equal the single-device
restatement (float64: equal up to summation order)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_klein as rk


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _inputs(p):
    g = np.random.default_rng(11)
    s_txt, grid = 7, (3, 5)
    latent = g.standard_normal((grid[0] * grid[1], p.in_channels))
    txt = g.standard_normal((s_txt, p.txt_embed_dim))
    cos, sin = rk.compute_rope(np.concatenate([rk.create_txt_ids(s_txt), rk.create_img_ids(*grid)], 0))
    return latent, txt, cos, sin


def _rank_main(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import omx_import
    omx_import.load_package()
    from ominix_mlx_amd import klein
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = rk.KleinParams.tiny()
    shards = klein.shard_state_dict(rk.synth_weights(p), p.hidden_size, p.mlp_hidden, rank, world)
    calls = [0]

    def allreduce(x):
        calls[0] += 1
        t = torch.from_numpy(np.ascontiguousarray(x, np.float64))
        dist.all_reduce(t)
        return t.numpy()

    out = rk.KleinOracle(p, shards, tp=(world, allreduce)).forward_with_rope(*_inputs(p)[:2], 620.0, *_inputs(p)[2:])
    ret[rank] = (out, calls[0], {k: v.shape for k, v in shards.items()})
    dist.destroy_process_group()


def test_dit_tensor_parallel_equals_single_device():
    world = 2
    p = rk.KleinParams.tiny()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rank_main, args=(world, _free_port(), ret), nprocs=world, join=True)
        outs = [ret[r][0] for r in range(world)]
        n_calls = ret[0][1]
        shapes = ret[1][2]
    latent, txt, cos, sin = _inputs(p)
    want = rk.KleinOracle(p, rk.synth_weights(p)).forward_with_rope(latent, txt, 620.0, cos, sin)
    np.testing.assert_allclose(outs[0], want, rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(outs[0], outs[1])                 # replicated residual stream
    assert n_calls == 4 * p.depth + p.depth_single                   # 2 per double-block stream, 1 per single block
    h, mh = p.hidden_size, p.mlp_hidden
    assert shapes["double_blocks.0.img_to_q.weight"] == (h // 2, h)
    assert shapes["double_blocks.0.txt_to_out.weight"] == (h, h // 2)
    assert shapes["double_blocks.0.img_mlp_in.weight"] == (mh, h) and shapes["double_blocks.0.img_mlp_out.weight"] == (h, mh // 2)
    assert shapes["single_blocks.0.to_qkv_mlp.weight"] == ((3 * h + 2 * mh) // 2, h)
    assert shapes["single_blocks.0.to_out.weight"] == (h, (h + mh) // 2)
    assert shapes["x_embedder.weight"] == (h, p.in_channels)         # replicated
