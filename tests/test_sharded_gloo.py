"""CPU, world_size 2 over gloo: the N>1 exchange step (all_gather layout, global ids, route mask).
The per-shard scan is stood in for by the oracle (the HIP scan needs a GPU); what is under test is
ragroute_amd.sharded's candidate exchange, checked against an oracle search of the concatenated corpus."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from ragroute_amd import sharded as S
        from tests.util import int_data
        rng = np.random.default_rng(100)  # same on both ranks
        shards = [int_data(rng, 700 + 50 * s, 64) for s in range(world)]
        xq = int_data(rng, 6, 64)
        k = 8
        D, I = O.flat_search_ip(shards[rank], xq, k)  # stand-in for FlatIndex.search_prepared
        I = np.where(I >= 0, I + (rank << S.SHARD_SHIFT), -1)
        mask = torch.tensor([[True, True], [True, False], [False, True], [False, False], [True, True], [True, True]])
        Dm, Im = S.apply_route_mask(torch.from_numpy(D), torch.from_numpy(I), mask[:, rank])
        Dg, Ig = S.gather_candidates(Dm, Im)
        assert tuple(Dg.shape) == (6, world * k)
        # rank-major column blocks
        assert torch.equal(Dg[:, rank * k:(rank + 1) * k], Dm) and torch.equal(Ig[:, rank * k:(rank + 1) * k], Im)
        # the packed single-collective form must give the same layout
        buf, Dp, Ip = S.alloc_packed(6, k, "cpu")
        Dp.copy_(Dm)
        Ip.copy_(Im)
        Dg2, Ig2 = S.gather_packed(buf, 6, k)
        assert torch.equal(Dg2, Dg) and torch.equal(Ig2, Ig)
        Do, Io = O.merge_topk(Dg.numpy(), Ig.numpy(), k, True)
        # expectation: per query, oracle search over the union of the shards the mask selects
        for q in range(6):
            sel = [s for s in range(world) if mask[q, s]]
            if not sel:
                assert (Io[q] == -1).all()
                continue
            cat = np.concatenate([shards[s] for s in sel])
            gids = np.concatenate([np.arange(len(shards[s])) + (s << S.SHARD_SHIFT) for s in sel])
            Dr, Ir = O.flat_search_ip(cat, xq[q:q + 1], k)
            assert np.array_equal(Do[q], Dr[0])
            assert np.array_equal(Io[q], gids[Ir[0]])   # ascending global id tie rule == concatenated row order
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_candidate_exchange(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0") and os.path.exists(tmp_path / "ok1")
