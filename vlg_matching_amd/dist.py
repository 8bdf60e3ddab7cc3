"""One process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm): the read-only index is built once and its
contiguous HBM image is broadcast over xGMI, query batches are sharded, only three counters are reduced.
There is no exchange step inside a query, so no other collective exists on this path."""
import numpy as np


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced [begin, end) slice of n_items for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_by_work(weights, world):
    """Contiguous slices with roughly equal total weight (e.g. sum of SA-interval sizes per query): heavy-tailed
    occurrence counts make equal query counts a poor balance.  -> list of (begin, end)"""
    w = np.asarray(weights, dtype=np.float64)
    total = float(w.sum())
    if total <= 0:
        return [shard_bounds(len(w), r, world) for r in range(world)]
    cum = np.cumsum(w)
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        i = int(np.searchsorted(cum, target, side="left"))          # cum[i-1] < target <= cum[i]: the query that crosses the target
        if i < len(w) and (cum[i] - target) <= (target - (cum[i - 1] if i else 0.0)):
            i += 1                                                   # ... goes to the earlier rank when that is the nearer cut
        cuts.append(min(len(w), max(cuts[-1], i)))
    cuts.append(len(w))
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_by_affinity(l, r, qsub, world):
    """Query sets (index arrays, one per rank) for a batch whose queries share occurrence lists: a list that several queries use
    is located and sorted once per GPU that holds one of them, so contiguous slices repeat that work on every rank.  Queries
    are grouped by their LONGEST list (identified by its SA interval [l, r]; equal intervals are the same list), a group costs
    that list once plus the lists its queries join, and the groups go to the least loaded rank, heaviest first.
    l, r: per sub-pattern (vlg_queries_intervals); qsub[nq+1]: query i owns the sub-patterns [qsub[i], qsub[i+1])."""
    l = np.asarray(l, dtype=np.uint64)
    occ = (np.asarray(r, dtype=np.uint64) + np.uint64(1) - l).astype(np.float64)
    nq = len(qsub) - 1
    key = np.zeros(nq, dtype=np.uint64)
    big = np.zeros(nq)
    w = np.ones(nq)
    for i in range(nq):
        a, b = int(qsub[i]), int(qsub[i + 1])
        if b > a and occ[a:b].min() > 0:
            j = a + int(np.argmax(occ[a:b]))
            key[i], big[i], w[i] = l[j], occ[j], 1.0 + occ[a:b].sum()
        else:
            key[i] = np.uint64(0xFFFFFFFFFFFFFFFF)                 # nothing to locate: any rank
    order = np.argsort(key, kind="stable")
    cuts = np.flatnonzero(np.concatenate([[True], key[order][1:] != key[order][:-1]]))
    groups = [order[a:b] for a, b in zip(cuts, list(cuts[1:]) + [nq])]
    cost = [big[g[0]] + w[g].sum() for g in groups]
    load = np.zeros(world)
    out = [[] for _ in range(world)]
    for gi in np.argsort(cost)[::-1]:
        g = groups[gi]
        if key[g[0]] == np.uint64(0xFFFFFFFFFFFFFFFF) or len(g) * world > nq:        # free queries, or a group too big for one rank: spread them
            for part in np.array_split(g, world):
                t = int(np.argmin(load))
                out[t].append(part)
                load[t] += (big[g[0]] if len(part) else 0.0) + w[part].sum()
            continue
        t = int(np.argmin(load))
        out[t].append(g)
        load[t] += cost[gi]
    return [np.sort(np.concatenate(o)) if o else np.zeros(0, dtype=np.int64) for o in out]


class Comm:
    """An RCCL communicator made through the C-ABI (vlg_comm_*): what a host without PyTorch uses, and what
    vlg_index_broadcast / vlg_comm_allgatherv take.  One per process, bound to the process's current device."""

    def __init__(self, handle, n_ranks, rank):
        self._h, self.n_ranks, self.rank = handle, n_ranks, rank

    @classmethod
    def create(cls, n_ranks, rank, share_id):
        """share_id(bytes_or_None) -> bytes: hands rank 0's 128-byte id to every rank (called on every rank; rank 0 passes the id,
        the others None) -- e.g. a torch.distributed broadcast_object_list, a pipe, a file."""
        import ctypes as C
        from .capi import check, lib
        ident = None
        if rank == 0:
            buf = C.create_string_buffer(128)
            check(lib().vlg_comm_unique_id(buf))
            ident = buf.raw
        ident = share_id(ident)
        h = C.c_void_p()
        check(lib().vlg_comm_create(C.create_string_buffer(ident, 128), n_ranks, rank, C.byref(h)))
        return cls(h, n_ranks, rank)

    @classmethod
    def from_torch_dist(cls, dist):
        """A communicator over the ranks of an initialised torch.distributed job (the id travels through its object broadcast)."""
        def share(ident):
            box = [ident]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        return cls.create(dist.get_world_size(), dist.get_rank(), share)

    def library(self):
        from .capi import lib
        return (lib().vlg_comm_library() or b"").decode()

    def info(self):
        import ctypes as C
        from .capi import check, lib
        n, r = C.c_int(), C.c_int()
        check(lib().vlg_comm_info(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def broadcast_index(self, idx, root=0, stream=None):
        """vlg_index_broadcast: the root passes its VlgIndex and gets it back; the other ranks pass None and get a new one"""
        import ctypes as C
        from .capi import check, lib
        from .index import VlgIndex
        out = C.c_void_p()
        check(lib().vlg_index_broadcast(idx._h if idx is not None else None, self._h, root, stream, C.byref(out)))
        return idx if self.rank == root else VlgIndex(out)

    def allreduce_sum_u64(self, values, stream=None):
        """sum over the ranks modulo 2^64 (num_results, checksum, located occurrences) -> list of ints"""
        from .capi import check, lib
        v = np.array([int(x) % (1 << 64) for x in values], dtype=np.uint64)
        check(lib().vlg_comm_allreduce_sum_u64(self._h, v.ctypes.data, len(v), stream))
        return [int(x) for x in v]

    def allgatherv(self, d_send_ptr, counts, elem_bytes, d_recv_ptr, stream=None):
        from .capi import check, lib
        c = np.ascontiguousarray(counts, dtype=np.uint64)
        assert len(c) == self.n_ranks
        check(lib().vlg_comm_allgatherv(self._h, d_send_ptr, c.ctypes.data, int(elem_bytes), d_recv_ptr, stream))

    def alltoallv(self, d_send_ptr, send_counts, d_recv_ptr, recv_counts, elem_bytes, stream=None):
        """vlg_comm_alltoallv: grouped ncclSend / ncclRecv, one pair per peer"""
        from .capi import check, lib
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        rc = np.ascontiguousarray(recv_counts, dtype=np.uint64)
        assert len(sc) == len(rc) == self.n_ranks
        check(lib().vlg_comm_alltoallv(self._h, d_send_ptr, sc.ctypes.data, d_recv_ptr, rc.ctypes.data, int(elem_bytes), stream))

    def close(self):
        if self._h:
            from .capi import lib
            lib().vlg_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_exchange(dist):
    """An exchange callback for Workspace.set_exchange that moves the pieces through torch.distributed on the HOST (D2H, all_gather
    of padded uint8 tensors, H2D): for rehearsals of the collective search with several gloo ranks on one device, and for hosts
    whose ranks have no RCCL between them.  The production path is Workspace.set_comm (RCCL over xGMI)."""
    import ctypes as C
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def exchange(d_buf, counts, elem_bytes, n_ranks, rank, stream):
        if hip.hipStreamSynchronize(stream):
            return 2
        sizes = [c * elem_bytes for c in counts]
        offs = [0]
        for b in sizes:
            offs.append(offs[-1] + b)
        mx = max(max(sizes), 1)
        mine = torch.zeros(mx, dtype=torch.uint8)
        if sizes[rank] and hip.hipMemcpy(mine.data_ptr(), d_buf + offs[rank], sizes[rank], 2):      # device -> host
            return 3
        parts = [torch.zeros(mx, dtype=torch.uint8) for _ in range(n_ranks)]
        dist.all_gather(parts, mine)
        for r in range(n_ranks):
            if r != rank and sizes[r] and hip.hipMemcpy(d_buf + offs[r], parts[r].data_ptr(), sizes[r], 1):   # host -> device
                return 4
        return 0
    return exchange


def host_alltoall(dist):
    """An exchange callback for Workspace.set_exchange_alltoall that moves the packed pieces through torch.distributed on the HOST
    (D2H, point-to-point isend / irecv of uint8 tensors -- gloo has no all-to-all --, H2D): rehearsals of the pairwise exchange with
    several gloo ranks on one device.  The production path is Workspace.set_comm (vlg_comm_alltoallv: RCCL over xGMI)."""
    import ctypes as C
    import torch
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def exchange(d_send, send_counts, d_recv, recv_counts, elem_bytes, n_ranks, rank, stream):
        if hip.hipStreamSynchronize(stream):
            return 2
        ssz = [c * elem_bytes for c in send_counts]
        rsz = [c * elem_bytes for c in recv_counts]
        out = torch.zeros(max(sum(ssz), 1), dtype=torch.uint8)
        inn = torch.zeros(max(sum(rsz), 1), dtype=torch.uint8)
        if sum(ssz) and hip.hipMemcpy(out.data_ptr(), d_send, sum(ssz), 2):                       # device -> host
            return 3
        reqs, so, ro = [], 0, 0
        for r in range(n_ranks):
            if r == rank:
                inn[ro: ro + rsz[r]] = out[so: so + ssz[r]]
            else:
                if ssz[r]:
                    reqs.append(dist.isend(out[so: so + ssz[r]], dst=r))
                if rsz[r]:
                    reqs.append(dist.irecv(inn[ro: ro + rsz[r]], src=r))
            so += ssz[r]
            ro += rsz[r]
        for q in reqs:
            q.wait()
        if sum(rsz) and hip.hipMemcpy(d_recv, inn.data_ptr(), sum(rsz), 1):                       # host -> device
            return 4
        return 0
    return exchange


def replicate_index(idx, dist, device, src=0, comm=None):
    """Broadcast the index image from rank `src` to every rank's HBM and attach to it.  `idx` is None elsewhere.
    With a Comm (RCCL through the C-ABI) this is vlg_index_broadcast: one ncclBroadcast straight out of / into the index's own
    allocation; without one the image goes through a torch.distributed broadcast of a uint8 tensor (gloo rehearsals on CPU hosts)."""
    if comm is not None:
        return comm.broadcast_index(idx, root=src)
    import torch
    from .index import VlgIndex
    rank = dist.get_rank()
    nb = torch.tensor([idx.blob_bytes() if rank == src else 0], dtype=torch.int64, device=device)
    dist.broadcast(nb, src)
    blob = torch.empty(int(nb.item()), dtype=torch.uint8, device=device)
    if rank == src:
        idx.blob_export(blob.data_ptr(), blob.numel())
    dist.broadcast(blob, src)
    if rank == src:
        return idx
    return VlgIndex.attach_blob(blob.data_ptr(), blob.numel(), keep=blob)


def reduce_checksum(checksum, dist=None, device=None):
    """Sum of the ranks' checksums modulo 2^64 (gm_search.cpp:110-114 adds positions into a uint64): reduced as two 32-bit
    halves so that no partial sum overflows the int64 the collective adds in."""
    chk = int(checksum) % (1 << 64)
    if dist is None or dist.get_world_size() == 1:
        return chk
    import torch
    t = torch.tensor([chk & 0xFFFFFFFF, chk >> 32], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    lo, hi = (int(x) for x in t.tolist())
    return (lo + (hi << 32)) % (1 << 64)


def run_sharded(search_fn, queries, dist=None, weights=None):
    """Search this rank's slice of `queries` with search_fn(list) -> (counts[q], checksum, located) and reduce the totals.
    -> dict(local_range, counts (local), num_results, checksum, located)   [totals are global]"""
    rank = dist.get_rank() if dist is not None else 0
    world = dist.get_world_size() if dist is not None else 1
    b, e = (shard_by_work(weights, world)[rank] if weights is not None else shard_bounds(len(queries), rank, world))
    counts, checksum, located = search_fn(queries[b:e])
    # gm_search's checksum is a sum modulo 2^64 (gm_search.cpp:110-114): it is split into 32-bit halves for the reduction, so that
    # no partial sum overflows the int64 the collective adds in, and put together modulo 2^64 afterwards
    chk = int(checksum) % (1 << 64)
    tot = np.array([int(np.sum(counts)), chk & 0xFFFFFFFF, chk >> 32, int(located)], dtype=np.int64)
    if dist is not None and world > 1:
        import torch
        t = torch.from_numpy(tot.copy())
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tot = t.cpu().numpy()
    return {"local_range": (b, e), "counts": np.asarray(counts), "num_results": int(tot[0]),
            "checksum": (int(tot[1]) + (int(tot[2]) << 32)) % (1 << 64), "located": int(tot[3])}
