"""Time-sharding the stream across GPUs.

Frames only depend on the M*P - 1 samples before them, so G ranks each take one contiguous
segment (a multiple of D samples) and the only exchange is the *halo*: the last
``halo_samples`` = M*P - 1 - input_offset raw integer samples of segment g -- (P-1)*M with the
default offset at D = M -- are the filter state of segment g+1.  The sharded output is
bit-identical to the single-stream output because the halo is raw input, not a partial result.

Who does what: the C library orchestrates a shard's step (``pfb_process_shard_async``: exchange on
a side stream, the frames that do not touch the halo at once, the head frames when the halo has
landed) and calls back for the transport; ``make_exchange`` below is that callback over
``torch.distributed`` -- one process per GPU, backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" (host-staged) in rehearsals and the CPU tests.  A C++ host passes a callback that calls
ncclSend / ncclRecv instead (examples/sharded_rccl.cpp).  The message is tiny (61 KB at M = 1024,
P = 16): latency-, not bandwidth-bound, and hidden under the interior frames.

The reference's own batch job -- the loop over a folder of records in
matlab/create_pdws_channelized.m:22-143 -- shards by RECORD instead: every record gets a fresh
channelizer (:33), so ranks take whole files and nothing is exchanged but the PDW lists at the end
(``records_for_rank``, ``pdws_from_folder``).
"""
from __future__ import annotations

import ctypes as C


def segment_bounds(total_samples: int, world: int, decimation: int) -> list[tuple[int, int]]:
    """Cut [0, total) into `world` contiguous segments on frame boundaries (multiples of D)."""
    frames = total_samples // decimation
    per = [frames // world + (1 if r < frames % world else 0) for r in range(world)]
    bounds, s = [], 0
    for r in range(world):
        e = s + per[r] * decimation
        bounds.append((s, e))
        s = e
    return bounds


def _p2p(snd, rcv, send_to: int, recv_from: int, group=None):
    """One neighbour exchange: ``snd`` (uint8 tensor or None) to rank ``send_to``, ``rcv`` (or None) from rank
    ``recv_from``.  RCCL has no 16-bit integer type and gloo cannot touch device memory: raw bytes travel, and for
    gloo they are staged on the host."""
    import torch
    import torch.distributed as dist
    ref = snd if snd is not None else rcv
    on_host = dist.get_backend(group) == "gloo" and ref.is_cuda
    rcv_dev = rcv
    if on_host:
        snd = snd.cpu() if snd is not None else None
        rcv = torch.empty(rcv_dev.numel(), dtype=torch.uint8) if rcv_dev is not None else None
    ops = []
    if snd is not None:
        ops.append(dist.P2POp(dist.isend, snd, send_to, group))
    if rcv is not None:
        ops.append(dist.P2POp(dist.irecv, rcv, recv_from, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()  # nccl: the current stream waits, not the host
    if on_host and rcv_dev is not None:
        rcv_dev.copy_(rcv)


def _allgather(snd, rcv, nbytes: int, rank: int, world: int, recv_from: int, group=None):
    """The same exchange as ONE collective (SURVEY.md section 8e's alternative): every rank contributes its tail (zeros
    if it has nothing to send) and takes its predecessor's.  world x halo bytes, a few KB; no point-to-point channels."""
    import torch
    import torch.distributed as dist
    ref = snd if snd is not None else rcv
    on_host = dist.get_backend(group) == "gloo" and ref.is_cuda
    dev = "cpu" if on_host else ref.device
    mine = torch.zeros(nbytes, dtype=torch.uint8, device=dev) if snd is None else (snd.cpu() if on_host else snd)
    everyone = torch.empty(world * nbytes, dtype=torch.uint8, device=dev)  # flat: gloo takes no other shape
    dist.all_gather_into_tensor(everyone, mine, group=group)
    if rcv is not None:
        rcv.copy_(everyone[recv_from * nbytes:(recv_from + 1) * nbytes])


def exchange_halo(tail, halo_out, rank: int, world: int, group=None, ring: bool = True):
    """Tensor-level form: send ``tail`` (my segment's last halo samples) to rank+1 and receive the previous rank's into
    ``halo_out``.  With ring=False rank 0 receives nothing and the last rank sends nothing."""
    import torch
    if world == 1:
        return halo_out
    sending = ring or rank + 1 < world
    receiving = ring or rank > 0
    _p2p(tail.contiguous().view(torch.uint8).reshape(-1) if sending else None,
         halo_out.view(torch.uint8).reshape(-1) if receiving else None, (rank + 1) % world, (rank - 1) % world, group)
    return halo_out


def exchange_halo_allgather(tail, halo_out, rank: int, world: int, group=None):
    """exchange_halo(ring=True) as one all_gather."""
    import torch
    if world == 1:
        return halo_out
    snd = tail.contiguous().view(torch.uint8).reshape(-1)
    _allgather(snd, halo_out.view(torch.uint8).reshape(-1), snd.numel(), rank, world, (rank - 1) % world, group)
    return halo_out


class _DeviceBytes:
    """A raw device pointer as a ``__cuda_array_interface__`` object, so torch can wrap library-owned memory."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _wrap(ptr: int, nbytes: int, device):
    import torch
    if not ptr:
        return None
    if device is None:  # host memory (the CPU tests' stand-in channelizer)
        return torch.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=torch.uint8)
    return torch.as_tensor(_DeviceBytes(ptr, nbytes), device=torch.device("cuda", device))


def make_exchange(rank: int, world: int, group=None, mode: str = "p2p", device: int | None = None):
    """The halo-exchange callback of ``Channelizer.attach_shard`` over torch.distributed.

    The library hands over raw pointers (the tail of my segment, the landing zone inside the handle), the byte count,
    the two peer ranks and the side stream the transfers must be enqueued on.  ``mode``: "p2p" = neighbour
    send/recv (ncclSend / ncclRecv under the nccl backend), "allgather" = one all_gather of the tails.
    ``device``: the CUDA ordinal the pointers live on, None for host pointers."""
    if mode not in ("p2p", "allgather"):
        raise ValueError(mode)

    def exchange(d_send: int, d_recv: int, nbytes: int, send_to: int, recv_from: int, stream: int) -> int:
        snd = _wrap(d_send, nbytes, device) if send_to >= 0 else None
        rcv = _wrap(d_recv, nbytes, device) if recv_from >= 0 else None

        def go():
            if mode == "p2p":
                _p2p(snd, rcv, send_to, recv_from, group)
            else:
                _allgather(snd, rcv, nbytes, rank, world, recv_from, group)

        if device is None:
            go()
        else:
            import torch
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=torch.device("cuda", device))):
                go()
        return 0

    return exchange


class ShardedChannelizer:
    """One rank's view of a time-sharded channelizer run.

    >>> sc = ShardedChannelizer(ch, rank, world)          # ch: a Channelizer on this rank's GPU
    >>> y_local = sc.process_segment(my_segment_tensor)   # frames of my segment only
    """

    def __init__(self, channelizer, rank: int, world: int, group=None, mode: str = "p2p", ring: bool = False):
        self.ch, self.rank, self.world, self.group = channelizer, rank, world, group
        device = getattr(channelizer, "device_index", None)
        channelizer.attach_shard(rank, world, make_exchange(rank, world, group, mode, device) if world > 1 else None,
                                 ring=ring)

    def process_segment(self, segment, first_frame: int = 0, out=None, reset: bool = True):
        """``segment``: this rank's contiguous slice (device tensor, length a multiple of D, longer than the head
        frames).  ``first_frame``: global index of the segment's first frame (only matters for derotate).  Returns
        when the result is complete."""
        if reset:
            self.ch.reset()  # segment 0 starts the stream: zero state, like a fresh dsp.Channelizer
        self.ch.set_frame_index(first_frame)
        y = self.ch.process_shard(segment, out=out)
        self.ch.sync()
        return y


def records_for_rank(paths, rank: int, world: int) -> list[str]:
    """The records rank ``rank`` processes: sorted (the recorder's file names sort by time), dealt round-robin so
    that every rank's share spans the whole capture."""
    return sorted(paths)[rank::world]


def pdws_from_folder(channelizer, paths, rank: int = 0, world: int = 1, group=None, gather: bool = True, **kw):
    """create_pdws_channelized.m's loop over a folder, one process per GPU: this rank's records go through
    ``pdw.pdws_from_iq_file`` (record -> channelizer -> PDWs, matrix on the GPU); with gather=True every rank gets the
    PDWs of all records concatenated in file order -- the script's accumulated ``pdw`` struct -- via one
    all_gather_object (the only collective; PDW lists are small).  Returns (pdws, [(path, count), ...])."""
    import numpy as np
    from .pdw import PDW_DTYPE, pdws_from_iq_file
    mine = []
    for path in records_for_rank(paths, rank, world):
        pdws, _info = pdws_from_iq_file(channelizer, path, **kw)
        mine.append((path, pdws))
    if world > 1 and gather:
        import torch.distributed as dist
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        mine = [item for part in everyone for item in part]
    mine.sort(key=lambda item: item[0])
    parts = [p for _, p in mine]
    allp = np.concatenate(parts) if parts else np.zeros(0, dtype=PDW_DTYPE)
    return allp, [(path, len(p)) for path, p in mine]
