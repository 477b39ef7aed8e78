"""BASELINE config C4 on the hardware available to the tests: the bin layers of one database split over several ranks,
every rank evaluating its slice with the real library on the GPU (not the oracle), results gathered to rank 0 through
shard.ResultGather -- the double-buffered run_into / join / gather sequence bench.py times -- and compared, bit for bit,
with one handle evaluating all bin layers.  One MI355X is visible to the tests, so the ranks share it and the
collective runs over `gloo` (RCCL refuses two ranks on one device); a one-rank RCCL group covers the device-tensor path.
Consecutive queries differ, so a gather that read a buffer too early or a run that overwrote it too soon shows up as a
mismatch."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, backend, b, nq, q, nslots=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    try:
        device = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from nested_hashing_psi_amd import pie, shard
        N, L, t, K, E = 4096, 2, 65537, 2, 5
        rng = np.random.default_rng(42)  # the same database, key and queries on every rank

        def rl(shape, moduli):
            out = np.zeros(shape + (L, N), dtype=np.uint64)
            for i in range(L):
                out[..., i, :] = rng.integers(0, int(moduli[i]), shape + (N,), dtype=np.uint64)
            return out

        stream = torch.cuda.Stream(device)
        cc = pie.PieContext(N, L, t, device=0, stream=stream.cuda_stream)
        db, masks, evk = rl((K, b, E), cc.q), rl((b,), cc.q), rl((L, 2), cc.q)
        queries = [(rl((K, E, 2), cc.q), rl((2,), cc.q)) for _ in range(nq)]
        cc.load_relin_key(evk)
        lo, hi = shard.bin_slice(b, rank, world)
        op = None
        if hi > lo:
            op = pie.BatchedFHEHIPPIE(cc, vectorizedHCT=np.ascontiguousarray(db[:, lo:hi]), preCalcRandomMask=np.ascontiguousarray(masks[lo:hi]))
        ct_words = 2 * L * N
        # query slots (bench.py's N > 1 path): every slot has its own context, stream, inputs and double-buffered gather; the
        # queries go round the slots, so the slots' collectives interleave in the same order on every rank
        slots = []
        for s_ in range(nslots):
            st_ = stream if s_ == 0 else torch.cuda.Stream(device)
            cc_ = cc if s_ == 0 else pie.PieContext(N, L, t, device=0, stream=st_.cuda_stream)
            op_ = op if s_ == 0 else (pie.BatchedFHEHIPPIE(cc_, attachTo=op) if op is not None else None)
            slots.append(dict(cc=cc_, op=op_, stream=st_,
                              d_idx=torch.zeros((K, E, 2, L, N), dtype=torch.int64, device=device),
                              d_minus=torch.zeros((2, L, N), dtype=torch.int64, device=device),
                              rg=shard.ResultGather(op_, b, hi - lo, ct_words, device, st_, kind="gather")))
        got = []
        for i, (idx, minus) in enumerate(queries):
            sl = slots[i % nslots]
            with torch.cuda.stream(sl["stream"]):   # the slot's next query overwrites its input arrays in stream order
                sl["d_idx"].copy_(torch.from_numpy(idx.view(np.int64)), non_blocking=False)
                sl["d_minus"].copy_(torch.from_numpy(minus.view(np.int64)), non_blocking=False)
            if sl["op"] is not None:
                sl["op"].setIndexDevice(sl["d_idx"].data_ptr())
                sl["op"].setMinusCompareElementDevice(sl["d_minus"].data_ptr())
            got.append((sl, sl["rg"].step()))
        for sl in slots:
            sl["rg"].drain()
        torch.cuda.synchronize(device)
        ok = True
        detail = ""
        if rank == 0:
            # the last two queries of every slot are still in its two buffer sets: compare them with an unsharded handle
            cc1 = pie.PieContext(N, L, t, device=0)
            cc1.load_relin_key(evk)
            full = pie.BatchedFHEHIPPIE(cc1, vectorizedHCT=db, preCalcRandomMask=masks)
            for i in range(max(0, nq - 2 * nslots), nq):
                idx, minus = queries[i]
                full.setMinusCompareElement(minus)
                full.setIndex(idx)
                full.run()
                want = full.getResultList().reshape(b, ct_words).view(np.int64)
                rows = got[i][0]["rg"].rows(got[i][1]).cpu().numpy()
                if rows.shape != want.shape or not (rows == want).all():
                    ok = False
                    detail += "query %d differs; " % i
            cc1.close()
        for sl in reversed(slots):
            sl["cc"].close()
        q.put((rank, ok, detail))
    except Exception as e:  # report instead of hanging the parent on q.get
        import traceback
        q.put((rank, False, repr(e) + traceback.format_exc()))
    finally:
        try:
            dist.destroy_process_group()
        except Exception:
            pass


def _run(world, backend, b, nq=5, nslots=1):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, b, nq, q, nslots)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
    assert all(ok for _, ok, _ in res), res


@pytest.mark.parametrize("world,b", [(2, 7), (3, 14), (4, 14)])
def test_sharded_ranks_on_one_gpu_match_unsharded(world, b):
    """uneven splits included: b = 14 over 4 ranks is 3 + 4 + 3 + 4, b = 7 over 2 is 3 + 4"""
    _run(world, "gloo", b)


def test_sharded_ranks_with_query_slots():
    """two ranks x three query slots each (bench.py's N > 1 configuration): nine different queries go round the slots, every
    slot double-buffers its own gather, and the last two queries of every slot equal the unsharded evaluation"""
    _run(2, "gloo", 7, nq=9, nslots=3)


def test_one_rank_rccl_device_gather():
    """the RCCL form of the same sequence (device tensors, asynchronous gather on RCCL's stream), also over three slots"""
    _run(1, "nccl", 6)
    _run(1, "nccl", 6, nq=7, nslots=3)
