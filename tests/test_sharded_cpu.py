"""N>1 host logic on CPU (gloo, world_size 2): segment cutting and the halo exchange of
sdr_channelizer_amd/sharded.py -- the same callback (make_exchange, p2p and all_gather forms) the C library calls
from pfb_process_shard_async, here driven with host pointers.  The per-rank arithmetic is stood in for by the ORACLE
(as the checker -- there is no GPU here and the product has no CPU path): sharded == single stream, bit for bit,
because the halo is raw input samples, and exactly M*P - D of them ((P-1)*M at D = M)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.pfb_oracle import COracle, OracleConfig
from sdr_channelizer_amd import synth
from sdr_channelizer_amd.sharded import (ShardedChannelizer, exchange_halo, exchange_halo_allgather, records_for_rank,
                                         segment_bounds)

M, P, D, BW = 16, 4, 8, 12


HALO = M * P - D  # M*P - 1 - input_offset with the default offset D - 1: what pfb_halo_samples() returns


class OracleBackedChannelizer:
    """Test double with the Channelizer surface ShardedChannelizer uses (attach_shard / process_shard / reset /
    set_frame_index / sync); float64 oracle inside, host memory, and the same exchange callback contract as the library:
    exchange(d_send, d_recv, nbytes, send_to, recv_from, stream) with raw pointers."""

    decimation = D
    halo_samples = HALO

    def __init__(self, taps):
        self.o = COracle()
        self.taps = np.asarray(taps, np.float64)
        self.rank, self.world, self.exchange, self.ring = 0, 1, None, False
        self.last_halo = None
        self.reset()

    def reset(self):
        self.state = np.zeros((HALO, 2), np.int16)  # the stream's own past (zeros after reset)
        self.frame_index = 0

    def set_frame_index(self, f):
        self.frame_index = f

    def sync(self):
        pass

    def attach_shard(self, rank, world, exchange=None, ring=False):
        self.rank, self.world, self.exchange, self.ring = rank, world, exchange, ring

    def process_shard(self, seg, out=None):
        seg = np.ascontiguousarray(np.asarray(seg).reshape(-1, 2))
        assert seg.shape[0] % D == 0 and seg.shape[0] > HALO
        receiving = self.world > 1 and (self.ring or self.rank > 0)
        sending = self.world > 1 and (self.ring or self.rank + 1 < self.world)
        tail = np.ascontiguousarray(seg[-HALO:])
        # the last rank of a ring passes its carried state on (the library's rule, include/pfb_channelizer.h): its
        # successor, rank 0, works on the NEXT call's first segment
        passed = np.ascontiguousarray(self.state) if (self.ring and self.rank == self.world - 1) else tail
        halo = np.zeros((HALO, 2), np.int16)
        if sending or receiving:
            rc = self.exchange(passed.ctypes.data if sending else 0, halo.ctypes.data if receiving else 0, HALO * 4,
                               (self.rank + 1) % self.world if sending else -1,
                               (self.rank - 1) % self.world if receiving else -1, 0)
            assert rc == 0
        before = halo if receiving else self.state
        self.last_halo = before.copy()
        x = self.o.unpack(np.concatenate([before, seg]), BW)
        y = self.o.channelize(x, self.taps, OracleConfig(M, P, D))
        self.state = tail.copy()
        return y[HALO // D:]  # HALO is a whole number of frames: the rest are the segment's own


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, taps, mode, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bounds = segment_bounds(total, world, D)
        s, e = bounds[rank]
        seg = torch.from_numpy(synth.pulsed_iq_numpy(e - s, BW, np.int16, seed=4, start=s))
        ch = OracleBackedChannelizer(taps)
        sc = ShardedChannelizer(ch, rank, world, mode=mode)  # the library's callback contract over gloo, host pointers
        y = sc.process_segment(seg, first_frame=s // D)
        got_halo = ch.last_halo.copy()
        # ring variant used by bench.py: everyone receives, rank 0 from the last rank -- an ENDLESS stream, so two calls
        # with different data: call 2 of rank r is segment world + r of the stream
        ring = OracleBackedChannelizer(taps)
        rsc = ShardedChannelizer(ring, rank, world, mode=mode, ring=True)
        y1 = rsc.process_segment(seg, first_frame=s // D)  # (resets: the stream starts here)
        ring_halo1 = ring.last_halo.copy()
        s2 = total + s
        seg2 = torch.from_numpy(synth.pulsed_iq_numpy(e - s, BW, np.int16, seed=4, start=s2))
        y2 = rsc.process_segment(seg2, first_frame=s2 // D, reset=False)
        ring_halo2 = ring.last_halo.copy()
        # the tensor-level helpers agree with each other
        halo = torch.zeros((HALO, 2), dtype=torch.int16)
        exchange_halo(seg[-HALO:].contiguous(), halo, rank, world, ring=True)
        halo2 = torch.zeros((HALO, 2), dtype=torch.int16)
        exchange_halo_allgather(seg[-HALO:].contiguous(), halo2, rank, world)
        assert torch.equal(halo2, halo)
        ret[rank] = (np.asarray(y), halo.numpy().copy(), seg.numpy()[-HALO:].copy(), got_halo,
                     np.asarray(y1), np.asarray(y2), ring_halo1, ring_halo2, seg2.numpy()[-HALO:].copy())
    finally:
        dist.destroy_process_group()


def test_segment_bounds():
    assert segment_bounds(8 * 10, 2, 8) == [(0, 40), (40, 80)]
    b = segment_bounds(8 * 11 + 5, 3, 8)  # 11 frames over 3 ranks, tail samples dropped
    assert b == [(0, 32), (32, 64), (64, 88)]
    assert all((e - s) % 8 == 0 for s, e in b)


@pytest.mark.timeout(180)
@pytest.mark.parametrize("mode,world", [("p2p", 2), ("allgather", 2), ("p2p", 3), ("allgather", 4), ("p2p", 8)])  # 8: the driver's node
def test_two_rank_time_sharding_matches_single_stream(mode, world):
    total = D * 400
    taps = np.random.default_rng(0).standard_normal(M * P)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), total, taps, mode, ret), nprocs=world, join=True)
    iq = np.concatenate([synth.pulsed_iq_numpy(e - s, BW, np.int16, seed=4, start=s)
                         for s, e in segment_bounds(total, world, D)])
    o = COracle()
    single = o.channelize(o.unpack(iq, BW), taps, OracleConfig(M, P, D))
    sharded = np.concatenate([ret[r][0] for r in range(world)])
    assert sharded.shape == single.shape
    assert np.array_equal(sharded, single)  # raw-sample halo => identical bits
    for r in range(world):  # the tensor-level ring helper: each rank holds its predecessor's tail
        assert np.array_equal(ret[r][1], ret[(r - 1) % world][2])
    # the ring as the library runs it (an endless stream: call 2 continues call 1): two calls of `world` segments each
    # equal ONE pass over the 2 * total samples, bit for bit; rank 0 started from zero state and, in call 2, from the last
    # rank's tail of call 1 -- never from the last rank's CURRENT tail
    iq2 = np.concatenate([synth.pulsed_iq_numpy(e - s, BW, np.int16, seed=4, start=total + s)
                          for s, e in segment_bounds(total, world, D)])
    both = o.channelize(o.unpack(np.concatenate([iq, iq2]), BW), taps, OracleConfig(M, P, D))
    ring_rows = np.concatenate([ret[r][4] for r in range(world)] + [ret[r][5] for r in range(world)])
    assert np.array_equal(ring_rows, both)
    assert not ret[0][6].any() and np.array_equal(ret[0][7], ret[world - 1][2])
    for r in range(1, world):
        assert np.array_equal(ret[r][6], ret[r - 1][2]) and np.array_equal(ret[r][7], ret[r - 1][8])
    # open chain: rank 0 continued from its own (zero) state, rank r from exactly M*P - D samples of rank r - 1
    assert not ret[0][3].any()
    for r in range(1, world):
        assert np.array_equal(ret[r][3], ret[r - 1][2]) and ret[r][3].shape[0] == M * P - D


def _folder_worker(rank, world, port, paths, ret):
    """pdws_from_folder with the per-record extraction stubbed (no GPU here): the sharding and the gather are what
    is under test."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sdr_channelizer_amd.pdw as pdw_mod
        from sdr_channelizer_amd.sharded import pdws_from_folder

        def fake(ch, path, **kw):  # a record's "PDWs": toa = its index in the folder, one entry per (index + 1)
            i = int(os.path.basename(path).split("_")[1].split(".")[0])
            out = np.zeros(i + 1, dtype=pdw_mod.PDW_DTYPE)
            out["toa"] = i
            out["bin"] = rank
            return out, None

        pdw_mod.pdws_from_iq_file = fake
        allp, listing = pdws_from_folder(None, paths, rank, world)
        ret[rank] = (allp.copy(), listing)
    finally:
        dist.destroy_process_group()


def test_records_for_rank_partitions_the_folder():
    paths = [f"rec_{i:03d}.iq" for i in (5, 1, 4, 0, 3, 2, 6)]
    shares = [records_for_rank(paths, r, 3) for r in range(3)]
    assert shares[0] == ["rec_000.iq", "rec_003.iq", "rec_006.iq"] and shares[2] == ["rec_002.iq", "rec_005.iq"]
    assert sorted(p for s_ in shares for p in s_) == sorted(paths)


@pytest.mark.timeout(120)
def test_two_rank_record_sharding_gathers_every_records_pdws_in_file_order():
    world = 2
    paths = [f"/data/rec_{i:03d}.iq" for i in (3, 0, 4, 1, 2)]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_folder_worker, args=(world, _free_port(), paths, ret), nprocs=world, join=True)
    for r in range(world):
        allp, listing = ret[r]
        assert [os.path.basename(p) for p, _ in listing] == [f"rec_{i:03d}.iq" for i in range(5)]
        assert [c for _, c in listing] == [1, 2, 3, 4, 5]
        assert np.array_equal(allp["toa"], np.repeat(np.arange(5), np.arange(1, 6)))
        assert np.array_equal(allp["bin"], np.repeat(np.arange(5) % world, np.arange(1, 6)))  # who processed which record


@pytest.mark.timeout(180)
def test_bench_starts_its_own_ranks_and_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` outside torch.distributed.run (how the driver ran N = 1 in round 1): the parent starts two
    fresh rank processes before touching torch.  Here there is no GPU, so both ranks fail -- and the parent must report that
    and exit non-zero promptly instead of hanging in a collective or printing a line."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the ranks would run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--log2-samples", "20",
                        "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=150, env=env)
    assert r.returncode != 0
    assert "ranks failed" in r.stderr and r.stdout.strip() == ""
