"""Time-sharding the stream across GPUs.

Frames only depend on the M*P - 1 samples before them, so G ranks each take one
contiguous segment (a multiple of D samples) and the only exchange is the *halo*: the
last ``history_samples`` raw integer samples of segment g become the filter state of
segment g+1.  The sharded output is bit-identical to the single-stream output because the
halo is raw input, not a partial result.  One process per GPU; the transport is
``torch.distributed`` point-to-point (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Message size is tiny (61 KB at M=1024, P=16), so this is
latency- not bandwidth-bound.

The reference's own batch job -- the loop over a folder of records in
matlab/create_pdws_channelized.m:22-143 -- shards by RECORD instead: every record gets a fresh
channelizer (:33), so ranks take whole files and nothing is exchanged but the PDW lists at the end
(``records_for_rank``, ``pdws_from_folder``).
"""
from __future__ import annotations


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


def exchange_halo(tail, halo_out, rank: int, world: int, group=None, ring: bool = True):
    """Send ``tail`` (my segment's last history_samples samples) to rank+1 and receive the
    previous rank's into ``halo_out``.  With ring=False rank 0 receives nothing (stream start:
    its state stays as it is) and the last rank sends nothing."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return halo_out
    # RCCL has no 16-bit integer type and gloo cannot touch device memory: ship raw bytes, and for
    # gloo stage them on the host.
    on_host = dist.get_backend(group) == "gloo" and tail.is_cuda
    snd = tail.contiguous().view(torch.uint8).reshape(-1)
    rcv = halo_out.view(torch.uint8).reshape(-1)
    if on_host:
        snd, rcv_dev, rcv = snd.cpu(), rcv, torch.empty(rcv.numel(), dtype=torch.uint8)
    ops = []
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    sending = ring or rank + 1 < world
    receiving = ring or rank > 0
    if sending:
        ops.append(dist.P2POp(dist.isend, snd, nxt, group))
    if receiving:
        ops.append(dist.P2POp(dist.irecv, rcv, prv, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if on_host and receiving:
        rcv_dev.copy_(rcv)
    return halo_out


def exchange_halo_allgather(tail, halo_out, rank: int, world: int, group=None):
    """The ring exchange as ONE collective: every rank contributes its tail, takes its predecessor's
    (SURVEY.md section 8e's alternative: world x history_samples raw samples, a few KB).  Same result as
    exchange_halo(ring=True); no point-to-point channels to set up."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return halo_out
    on_host = dist.get_backend(group) == "gloo" and tail.is_cuda
    snd = tail.contiguous().view(torch.uint8).reshape(-1)
    if on_host:
        snd = snd.cpu()
    nb = snd.numel()
    everyone = torch.empty(world * nb, dtype=torch.uint8, device=snd.device)  # flat: gloo takes no other shape
    dist.all_gather_into_tensor(everyone, snd, group=group)
    prv = (rank - 1) % world
    halo_out.view(torch.uint8).reshape(-1).copy_(everyone[prv * nb:(prv + 1) * nb])
    return halo_out


class ShardedChannelizer:
    """One rank's view of a time-sharded channelizer run.

    >>> sc = ShardedChannelizer(ch, rank, world)          # ch: a Channelizer on this rank's GPU
    >>> y_local = sc.process_segment(my_segment_tensor)   # frames of my segment only
    """

    def __init__(self, channelizer, rank: int, world: int, group=None):
        self.ch, self.rank, self.world, self.group = channelizer, rank, world, group

    def process_segment(self, segment, first_frame: int = 0, out=None):
        """``segment``: this rank's contiguous slice (device tensor, length a multiple of D).
        ``first_frame``: global index of the segment's first frame (only matters for derotate)."""
        import torch
        hist = self.ch.history_samples
        self.ch.reset()
        if self.world > 1:
            if segment.shape[0] < hist:
                raise ValueError("segment shorter than the halo")
            halo = torch.empty((hist,) + tuple(segment.shape[1:]), dtype=segment.dtype, device=segment.device)
            exchange_halo(segment[segment.shape[0] - hist:].contiguous(), halo, self.rank, self.world, self.group,
                          ring=False)
            if self.rank > 0:
                self.ch.prime(halo)
        self.ch.set_frame_index(first_frame)
        return self.ch(segment, out=out)


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
