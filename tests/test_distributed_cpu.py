"""CPU (gloo, world_size 2 and 4): the data-parallel gradient exchange of meanflow_audio_codec_amd.distributed."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from meanflow_audio_codec_amd.distributed import GradReducer
        g = torch.Generator().manual_seed(100 + rank)
        grads = {
            "big/kernel": torch.randn(1 << 21, generator=g),                 # > small_numel: own all-reduce
            "big16/kernel": torch.randn(3000, 700, generator=g).bfloat16(),  # bf16 bucket
            "blocks_0/bias": torch.randn(128, generator=g),
            "blocks_0/conv_block/Conv_0/kernel": torch.randn(3, 3, 16, 16, generator=g),
        }
        ref = {k: v.clone() for k, v in grads.items()}
        loss = torch.tensor(0.25 * (rank + 1))
        red = GradReducer(small_numel=1 << 20)
        total = red.reduce(grads, loss)
        # gather the per-rank originals to check the sums
        for k, v in ref.items():
            parts = [torch.empty_like(v) for _ in range(world)]
            dist.all_gather(parts, v)
            want = sum(p.float() for p in parts)
            tol = 1e-5 if v.dtype == torch.float32 else 5e-2
            assert (grads[k].float() - want).abs().max() <= tol * max(1.0, want.abs().max().item()), k
        assert abs(total.item() - 0.25 * sum(range(1, world + 1))) < 1e-6
        # per-block entry points used by the overlapped train_step
        blk = {k: v.clone() for k, v in ref.items()}
        red.reduce_tensors(list(blk.values()))
        for k in blk:
            # same sums; bitwise only for two ranks (a ring over more ranks adds in an order that depends on where an
            # element sits in its bucket, and the two entry points bucket differently)
            if world == 2:
                assert torch.equal(blk[k], grads[k]), k
            else:
                assert torch.allclose(blk[k].float(), grads[k].float(), rtol=2e-2 if blk[k].dtype == torch.bfloat16 else 1e-5,
                                      atol=1e-5), k
        assert abs(red.reduce_scalar(loss).item() - total.item()) < 1e-6
        # the two halves of the sharded optimizer's exchange (gloo: emulated with all_reduce / all_gather on views)
        full = torch.arange(8 * world, dtype=torch.float32) * (rank + 1)
        mine = torch.empty(8)
        red._reduce_scatter(mine, full.clone())
        want = torch.arange(8 * world, dtype=torch.float32)[rank * 8:(rank + 1) * 8] * sum(range(1, world + 1))
        assert torch.equal(mine, want)
        gathered = torch.zeros(8 * world)
        red._all_gather(gathered, mine)
        assert torch.equal(gathered, torch.arange(8 * world, dtype=torch.float32) * sum(range(1, world + 1)))
        assert red.shard_optimizer and red.rank == rank and not red._native
        # in-place all-gather (the deferred gathers pass this rank's slice of the destination itself)
        inplace = torch.zeros(8 * world)
        inplace[rank * 8:(rank + 1) * 8] = mine
        red._all_gather(inplace, inplace[rank * 8:(rank + 1) * 8])
        assert torch.equal(inplace, gathered)
        # flush_gathers on host tensors: gathers every deferred leaf in forward-use order, no events without a GPU
        class _State:
            work = {"blocks_1/output_proj1/kernel": torch.full((4 * world,), -1.0),
                    "blocks_0/input_proj2/kernel": torch.full((4 * world,), -1.0)}
        for w in _State.work.values():
            w[rank * 4:(rank + 1) * 4] = float(rank + 1)
        names = list(_State.work)
        assert sorted(names, key=red._use_order) == ["blocks_0/input_proj2/kernel", "blocks_1/output_proj1/kernel"]
        red.flush_gathers(_State, names)
        assert names == []
        for w in _State.work.values():
            assert torch.equal(w, torch.arange(1, world + 1, dtype=torch.float32).repeat_interleave(4))
        assert red.defer_gather
        # second call reuses the flat bucket
        total2 = red.reduce({k: v.clone() for k, v in ref.items()}, loss)
        assert abs(total2.item() - total.item()) < 1e-6
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_grad_reducer_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(r, "ok") for r in range(world)], res


def test_reducer_requires_process_group():
    from meanflow_audio_codec_amd.distributed import GradReducer
    if dist.is_initialized():
        pytest.skip("process group already initialised")
    with pytest.raises(RuntimeError):
        GradReducer()


def test_use_order_and_workdict_without_gpu():
    from meanflow_audio_codec_amd.distributed import GradReducer
    from meanflow_audio_codec_amd.models.train_state import WorkDict
    names = ["blocks_10/input_proj1/kernel", "blocks_2/output_proj2/kernel", "blocks_2/input_proj1/kernel",
             "encoder/dense1/kernel", "blocks_2/output_proj1/kernel", "blocks_2/input_proj2/kernel"]
    assert sorted(names, key=GradReducer._use_order) == [
        "blocks_2/input_proj1/kernel", "blocks_2/input_proj2/kernel", "blocks_2/output_proj1/kernel",
        "blocks_2/output_proj2/kernel", "blocks_10/input_proj1/kernel", "encoder/dense1/kernel"]
    w = WorkDict()
    w["a"] = torch.ones(2)
    assert w.pending == {} and torch.equal(w["a"], torch.ones(2)) and w.get("b") is None and w.get("a") is w["a"]
    w.wait_all()                       # nothing pending: no device call
    assert list(w) == ["a"] and len(w) == 1


def test_interleaved_row_ownership_balances_the_rt_rule():
    """distributed.shard_rows + loss_strategies._order_rows: under interleaved ownership every rank holds the same
    number (+-1) of global rows below data_size = int(B*p) (utils.py:41-44), as a LOCAL PREFIX, and the shards
    partition the global batch.  Pure host arithmetic (no GPU)."""
    from meanflow_audio_codec_amd import ops
    from meanflow_audio_codec_amd.distributed import shard_of, shard_rows
    from meanflow_audio_codec_amd.trainers.loss_strategies import _order_rows
    for Bg, prop in ((128, 0.5), (128, 0.7), (100, 0.7), (16, 0.9), (8, 0.0), (8, 1.0), (24, 0.3)):
        gsz = ops.data_size_of(Bg, prop)
        assert gsz == int(Bg * prop)
        for G in (1, 2, 4, 8):
            if Bg % G:
                continue
            B = Bg // G
            glob = torch.arange(Bg)
            seen, n_rt = [], []
            for k in range(G):
                kw = shard_rows(k, G, B)
                assert kw == dict(row0=k, row_stride=G, global_batch=Bg)
                mine = shard_of(glob, k, G)
                assert torch.equal(mine, k + G * torch.arange(B))
                seen.append(mine)
                t = torch.zeros(B, 1)
                perm, n_tan = _order_rows(t, t, B, kw["row0"], Bg, prop, True, kw["row_stride"])
                ds = B - n_tan
                assert ds == int((mine < gsz).sum()) and bool((mine[:ds] < gsz).all()) and bool((mine[ds:] >= gsz).all())
                if perm is not None:          # tangent rows first, then the r == t rows, original order inside each
                    assert torch.equal(perm, torch.cat([torch.arange(ds, B), torch.arange(0, ds)]))
                n_rt.append(ds)
            assert sum(n_rt) == gsz and max(n_rt) - min(n_rt) <= 1, (Bg, prop, G, n_rt)
            assert torch.equal(torch.sort(torch.cat(seen)).values, glob)
        # contiguous ownership (row_stride 1), still supported: prefix rule per global row
        perm, n_tan = _order_rows(torch.zeros(4, 1), torch.zeros(4, 1), 4, Bg - 4, Bg, prop, True, 1)
        assert 4 - n_tan == max(0, min(4, gsz - (Bg - 4)))
