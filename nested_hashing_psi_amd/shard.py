"""Bin-layer sharding of BatchedFHEHIPPIE::run() across GPUs (one process per GPU).

The outer loop over bin layers (reference BatchedFHEHIPPIE.cpp:91) has independent iterations: each
reads its own K*E plaintexts and mask, shares the read-only index / minus ciphertexts and the
relinearisation key, and writes its own resultList[bin] (.cpp:127).  Rank r therefore owns a contiguous
slice of bin layers, holds only that slice of the packed database, and the single collective of the
path is the final gather of result ciphertexts (RCCL over xGMI; `gloo` in the CPU tests): to the rank that
answers the client (gather_bins_to), or to every rank (gather_bins).  With point-to-point xGMI links the
gather to one rank moves each slice over its own link once; the all-gather moves 8x the bytes.

The other exchange of a sharded server is per query, before run(): the index matrix and the minus element
((K E + 1) 2 L W bytes, 29 MiB at the headline configuration) reach the server once (reference
BatchedFHEPSIServer.cpp:94-95,114-141) and every rank needs all of them.  QueryBroadcast distributes them from
the receiving rank; ResultGather.step() takes it as its first stage, so both collectives are in the sharded
server's steady state.
"""
import torch
import torch.distributed as dist


def bin_slice(b, rank, world):
    """[lo, hi): the bin layers rank `rank` of `world` evaluates (sizes differ by at most one)"""
    return (b * rank) // world, (b * (rank + 1)) // world


def max_bins(b, world):
    return -(-b // world)


def gather_bins(local, b, world, out=None, group=None):
    """local: [b_local, W] result ciphertexts of this rank's slice (any integer dtype, device tensor for
    RCCL / CPU tensor for gloo).  Returns [b, W] in bin order on every rank."""
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = bin_slice(b, rank, world)
    assert local.shape[0] == hi - lo, "local result count does not match this rank's bin slice"
    if world == 1:
        return local
    bmax = max_bins(b, world)
    W = local.shape[1]
    padded = torch.zeros((bmax, W), dtype=local.dtype, device=local.device)
    padded[: hi - lo] = local
    if out is None:
        out = torch.empty((world * bmax, W), dtype=local.dtype, device=local.device)
    parts = list(out.view(world, bmax, W).unbind(0))
    dist.all_gather(parts, padded, group=group)
    if b == world * bmax:
        return out
    keep = []
    for r in range(world):
        rlo, rhi = bin_slice(b, r, world)
        keep.append(out[r * bmax: r * bmax + (rhi - rlo)])
    return torch.cat(keep, dim=0)


def gather_bins_to(local, b, world, dst=0, out=None, group=None, async_op=False):
    """As gather_bins, but only rank `dst` receives: the server process that sends the result list to the client
    (reference BatchedFHEPSIServer.cpp:143-152).  Slices are padded to max_bins(b, world) rows; `out` ([world * bmax, W],
    rank dst only) receives them in rank order.  Returns (out or None, work handle or None)."""
    if world == 1:
        return local, None
    rank = dist.get_rank(group)
    bmax = max_bins(b, world)
    W = local.shape[1]
    lo, hi = bin_slice(b, rank, world)
    if local.shape[0] == bmax:
        padded = local
    else:
        padded = torch.zeros((bmax, W), dtype=local.dtype, device=local.device)
        padded[: hi - lo] = local[: hi - lo]
    parts = None
    if rank == dst:
        if out is None:
            out = torch.empty((world * bmax, W), dtype=local.dtype, device=local.device)
        parts = list(out.view(world, bmax, W).unbind(0))
    work = dist.gather(padded, gather_list=parts, dst=dst, group=group, async_op=async_op)
    return (out if rank == dst else None), work


def shared_seeds(device=None, src=0, group=None):
    """(evict, shuffle, mask) seeds of a sharded database: drawn once from the OS CSPRNG on rank `src` and broadcast, so that
    every rank cuts its bin-layer slice out of the same table (pie.BatchedFHEHIPPIE(binSlice=...) requires explicit seeds)"""
    import secrets
    t = torch.zeros(3, dtype=torch.int64, device=device if dist.get_backend(group) != "gloo" else "cpu")
    if dist.get_rank(group) == src:
        t = torch.tensor([secrets.randbits(63) for _ in range(3)], dtype=torch.int64, device=t.device)
    dist.broadcast(t, src=src, group=group)
    return tuple(int(v) for v in t.cpu())


def slot_groups(n_slots):
    """One process group per direction of every query slot: [(query distribution, result gather)] * n_slots.

    ProcessGroupNCCL runs all collectives of one group on one internal stream in issue order.  With every slot's QueryBroadcast
    and ResultGather on the default group, the gather of slot s -- which waits for run(s) -- would sit in front of the
    distribution of slot s + 1's query, so run(s + 1) could not start before run(s) had finished and the slots would not overlap
    at all.  Each (slot, direction) therefore gets a communicator of its own.  Every rank must call this at the same point
    (dist.new_group is collective)."""
    return [(dist.new_group(), dist.new_group()) for _ in range(n_slots)]


class QueryBroadcast:
    """Per-query input distribution of the sharded server: one flat array [K E 2 L N + 2 L N] (index matrix, then the minus
    element) travels from rank `src` to every rank, double-buffered like the gather.

    kind "broadcast":      dist.broadcast -- RCCL pipelines it round a ring, so the whole array crosses single links.
    kind "scatter_gather": rank src sends a different 1/world of the array to every rank, then an all-gather completes it on
                           all: with point-to-point xGMI every link carries 1/world of the bytes, twice.
    source "hbm":  the query is resident in rank src's HBM when step() is called (set_query_device)
    source "host": rank src holds it in page-locked host memory (set_query_host) and uploads it first -- the reference server's
                   situation, PCIe included.
    With the `gloo` backend (CPU tests, one-GPU rehearsals) the array is broadcast between host buffers and uploaded on every
    rank.  Everything is enqueued on `in_stream`; ready(s) makes another stream wait for buffer set s."""

    def __init__(self, words, device, src=0, kind="broadcast", group=None):
        self.words, self.src, self.kind, self.group = words, src, kind, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        self.cuda = self.device.type == "cuda"       # CPU: the gloo rehearsal of the sequence in the CPU tests (no streams)
        self.host_staged = dist.get_backend(group) == "gloo"
        self.in_stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.chunk = -(-words // self.world)
        padded = self.chunk * self.world
        self.d_in = [torch.zeros(padded, dtype=torch.int64, device=self.device) for _ in range(2)]
        self.mine = {}   # scatter_gather: this rank's 1/world of the array, one buffer per device the array is distributed on
        self.used = [False, False]
        self.h_in = None
        if self.host_staged and self.cuda:
            self.h_in = [torch.zeros(padded, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.done = [torch.cuda.Event(), torch.cuda.Event()] if self.cuda else [None, None]   # buffer set s is complete on this rank
        self.free = [None, None]                                  # ... and no longer read by the run that used it
        self.host_query = None
        self.dev_query = None

    def set_query_host(self, flat):
        """rank src: the next query as a page-locked int64 tensor [words] (other ranks: ignored)"""
        self.host_query, self.dev_query = flat, None

    def set_query_device(self, flat):
        """rank src: the next query, resident in HBM"""
        self.dev_query, self.host_query = flat, None

    def release(self, s, event):
        """the run that read buffer set s has been queued; `event` is recorded behind it"""
        self.free[s] = event

    def _distribute(self, buf):
        if self.world == 1:
            return
        if self.kind == "scatter_gather":
            mine = self.mine.get(buf.device)
            if mine is None:
                mine = self.mine[buf.device] = torch.zeros(self.chunk, dtype=torch.int64, device=buf.device)
            parts = list(buf.view(self.world, self.chunk).unbind(0)) if self.rank == self.src else None
            dist.scatter(mine, scatter_list=parts, src=self.src, group=self.group)
            dist.all_gather_into_tensor(buf, mine, group=self.group)
        else:
            dist.broadcast(buf, src=self.src, group=self.group)

    def step(self, s):
        srcq = self.host_query if self.host_query is not None else self.dev_query
        if not self.cuda:
            if self.rank == self.src:
                self.d_in[s][: self.words].copy_(srcq[: self.words])
            self._distribute(self.d_in[s])
            return
        with torch.cuda.stream(self.in_stream):
            if self.free[s] is not None:
                self.in_stream.wait_event(self.free[s])
            buf = self.d_in[s]
            if self.host_staged:
                h = self.h_in[s]
                if self.used[s]:
                    self.done[s].synchronize()       # the upload of the query that used this host buffer has finished
                if self.rank == self.src:
                    h[: self.words].copy_(srcq[: self.words])          # (device source: a blocking download, rehearsal only)
                self._distribute(h)
                buf.copy_(h, non_blocking=True)
            else:
                if self.rank == self.src:
                    buf[: self.words].copy_(srcq[: self.words], non_blocking=True)
                self._distribute(buf)
            self.done[s].record(self.in_stream)
            self.used[s] = True

    def ready(self, s, stream=None):
        if self.cuda:
            stream.wait_event(self.done[s])
        return self.d_in[s]


class ResultGather:
    """The sharded server's steady state: run() of query i on this rank's bin layers, then the path's only collective --
    the gather of the result ciphertexts to the rank that answers the client (reference BatchedFHEPSIServer.cpp:143-152)
    -- overlapped with run() of query i+1 through two result buffers.

    Stream order (everything is enqueued under `stream`, the stream the PieContext was created on):
      works[s].wait()        the stream waits until the gather of query i-2 has read buffer s
      op.run(into=buf[s])    the kernels that write buf[s] wait for the stream (piehip_run_into)
      op.join()              the stream waits for the run
      gather(buf[s])         RCCL's stream waits for the stream, i.e. for the run
    With the `gloo` backend (CPU tests, and the two-process test on one GPU) the results are copied to pinned host
    buffers on the same stream and the host waits for that copy before the gather.
    """

    def __init__(self, op, b, b_local, ct_words, device, stream, kind="gather", dst=0, group=None, query=None, query_split=None,
                 batch=1, query_words=None):
        """query: a QueryBroadcast -- step() then first distributes the next query from rank dst and points the operator at the
        received copy; query_split = words of the index matrix (the minus element follows it in the flat array).
        batch > 1: the operator evaluates `batch` queries per run() (setQueryBatch); the distributed array holds them one after
        the other, query_words each (index matrix, then minus element), and a "ciphertext" row of the gather is the `batch`
        result ciphertexts of one bin layer (ct_words = batch * 2 L N: the library's rows are [bin layer][query])"""
        self.op, self.b, self.dst, self.kind, self.group = op, b, dst, kind, group
        self.query, self.query_split = query, query_split
        self.batch, self.query_words = batch, query_words
        if batch > 1 and query is not None and query_words is None:
            raise ValueError("a query batch needs query_words (words per query in the distributed array)")
        self.run_done = [torch.cuda.Event(), torch.cuda.Event()]
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.stream = stream
        self.host_staged = dist.get_backend(group) == "gloo"
        self.bmax = max_bins(b, self.world)
        self.b_local = b_local
        assert b_local <= self.bmax
        self.my_out = [torch.zeros((self.bmax, ct_words), dtype=torch.int64, device=device) for _ in range(2)]
        cdev = "cpu" if self.host_staged else device
        self.staged = [torch.zeros((self.bmax, ct_words), dtype=torch.int64).pin_memory() for _ in range(2)] if self.host_staged else None
        need_all = kind == "all_gather" or self.rank == dst
        self.gathered = [torch.empty((self.world * self.bmax, ct_words), dtype=torch.int64, device=cdev) if need_all else None
                         for _ in range(2)]
        self.works = [None, None]
        self.i = 0

    def _collective(self, s):
        src = self.staged[s] if self.host_staged else self.my_out[s]
        if self.kind == "all_gather":
            return dist.all_gather_into_tensor(self.gathered[s], src, group=self.group, async_op=True)
        parts = list(self.gathered[s].view(self.world, self.bmax, -1).unbind(0)) if self.rank == self.dst else None
        return dist.gather(src, gather_list=parts, dst=self.dst, group=self.group, async_op=True)

    def step(self):
        s = self.i & 1
        self.i += 1
        if self.query is not None:
            self.query.step(s)                       # the inputs of this query: from rank dst to every rank (its own stream)
        with torch.cuda.stream(self.stream):
            if self.works[s] is not None:
                self.works[s].wait()                 # buffer set s is free again (query i-2 gathered)
            if self.query is not None:
                flat = self.query.ready(s, self.stream)
                if self.op is not None and self.b_local:
                    for q in range(self.batch):
                        base = flat.data_ptr() + 8 * q * (self.query_words or 0)
                        self.op.setIndexDevice(base, query=q)
                        self.op.setMinusCompareElementDevice(base + 8 * self.query_split, query=q)
            if self.op is not None and self.b_local:
                self.op.run(sync=False, into=self.my_out[s].data_ptr())
                self.op.join()
            if self.query is not None:
                self.run_done[s].record(self.stream)
                self.query.release(s, self.run_done[s])
            if self.host_staged:
                self.staged[s].copy_(self.my_out[s], non_blocking=True)
                self.stream.synchronize()
            self.works[s] = self._collective(s)
        return s

    def drain(self):
        with torch.cuda.stream(self.stream):
            for w in self.works:
                if w is not None:
                    w.wait()
        self.works = [None, None]

    def rows(self, s):
        """the b result ciphertexts of buffer set s in bin order (rank dst, or every rank after an all_gather); call after
        drain() and a stream synchronisation"""
        g = self.gathered[s]
        keep = []
        for r in range(self.world):
            lo, hi = bin_slice(self.b, r, self.world)
            keep.append(g[r * self.bmax: r * self.bmax + (hi - lo)])
        return torch.cat(keep, dim=0)
