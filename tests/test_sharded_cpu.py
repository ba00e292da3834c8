"""N>1 host logic on CPU (gloo, world_size 2): segment cutting and the halo exchange of
sdr_channelizer_amd/sharded.py.  The per-rank arithmetic is stood in for by the ORACLE (as the
checker -- there is no GPU here and the product has no CPU path): sharded == single stream,
bit for bit, because the halo is raw input samples."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.pfb_oracle import COracle, OracleConfig
from sdr_channelizer_amd import synth
from sdr_channelizer_amd.sharded import ShardedChannelizer, exchange_halo, segment_bounds

M, P, D, BW = 16, 4, 8, 12


class OracleBackedChannelizer:
    """Test double with the Channelizer surface ShardedChannelizer uses; float64 oracle inside."""

    def __init__(self, taps):
        self.o = COracle()
        self.taps = np.asarray(taps, np.float64)
        self.history_samples = M * P + D
        self.reset()

    def reset(self):
        self.hist = np.zeros((self.history_samples, 2), np.int16)
        self.frame_index = 0

    def prime(self, iq):
        iq = np.asarray(iq).reshape(-1, 2)
        self.hist = np.concatenate([self.hist, iq])[-self.history_samples:]

    def set_frame_index(self, f):
        self.frame_index = f

    def __call__(self, seg, out=None):
        seg = np.asarray(seg).reshape(-1, 2)
        x = self.o.unpack(np.concatenate([self.hist, seg]), BW)
        y = self.o.channelize(x, self.taps, OracleConfig(M, P, D))
        self.prime(seg)
        return y[self.history_samples // D:]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, taps, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bounds = segment_bounds(total, world, D)
        s, e = bounds[rank]
        seg = torch.from_numpy(synth.pulsed_iq_numpy(e - s, BW, np.int16, seed=4, start=s))
        sc = ShardedChannelizer(OracleBackedChannelizer(taps), rank, world)
        y = sc.process_segment(seg, first_frame=s // D)
        # ring variant used by bench.py: everyone receives, rank 0 from the last rank
        halo = torch.zeros((M * P + D, 2), dtype=torch.int16)
        exchange_halo(seg[-(M * P + D):].contiguous(), halo, rank, world, ring=True)
        ret[rank] = (np.asarray(y), halo.numpy().copy(), seg.numpy()[-(M * P + D):].copy())
    finally:
        dist.destroy_process_group()


def test_segment_bounds():
    assert segment_bounds(8 * 10, 2, 8) == [(0, 40), (40, 80)]
    b = segment_bounds(8 * 11 + 5, 3, 8)  # 11 frames over 3 ranks, tail samples dropped
    assert b == [(0, 32), (32, 64), (64, 88)]
    assert all((e - s) % 8 == 0 for s, e in b)


@pytest.mark.timeout(120)
def test_two_rank_time_sharding_matches_single_stream():
    world, total = 2, D * 400
    taps = np.random.default_rng(0).standard_normal(M * P)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), total, taps, ret), nprocs=world, join=True)
    iq = np.concatenate([synth.pulsed_iq_numpy(e - s, BW, np.int16, seed=4, start=s)
                         for s, e in segment_bounds(total, world, D)])
    o = COracle()
    single = o.channelize(o.unpack(iq, BW), taps, OracleConfig(M, P, D))
    sharded = np.concatenate([ret[r][0] for r in range(world)])
    assert sharded.shape == single.shape
    assert np.array_equal(sharded, single)  # raw-sample halo => identical bits
    for r in range(world):  # ring: each rank holds its predecessor's tail
        assert np.array_equal(ret[r][1], ret[(r - 1) % world][2])
