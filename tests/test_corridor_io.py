"""Corridor wire format and replay: libdirect_ddp.so's host-side pack / unpack of
msgs/corridor in ROS 1 serialisation against golden bytes of an independent packer, and the replay batch
builder against a restatement of corridorRecCallBack / fastTrajPlanning (TRP:308-352, 796-823).
No GPU needed: these entry points are host code."""
import os
import struct

import numpy as np
import pytest

from direct_amd import abi, corridor_io, problems, solver
from tests import helpers


@pytest.fixture(scope="module")
def rec(built):
    g = np.load(os.path.join(helpers.GOLDEN_DIR, "corridor_msg.npz"))
    return corridor_io.Corridor(int(g["path_id"]), g["n_planes"], g["planes"], g["seeds"], g["centers"])


def golden_bytes():
    return open(os.path.join(helpers.GOLDEN_DIR, "corridor_msg.bin"), "rb").read()


def test_pack_matches_golden_bytes(rec):
    assert corridor_io.pack(rec) == golden_bytes()


def test_unpack_golden_bytes(rec):
    cor, used = corridor_io.unpack(golden_bytes(), n_seg_max=64, p_max=rec.p_max)
    assert used == len(golden_bytes()) and cor.path_id == 17 and cor.n_seg == 64
    assert (cor.n_planes == rec.n_planes).all() and np.array_equal(cor.seeds, rec.seeds)
    assert np.array_equal(cor.centers, rec.centers)
    for k in range(64):
        assert np.array_equal(cor.planes[k, :rec.n_planes[k]], rec.planes[k, :rec.n_planes[k]])


def test_malformed_buffers_are_rejected(rec):
    data = golden_bytes()
    for cut in (0, 3, 7, 8 + 20, len(data) - 1):
        with pytest.raises(solver.DirectError) as e:
            corridor_io.unpack(data[:cut], 64, rec.p_max)
        assert e.value.status == abi.DIRECT_ERR_INVALID
    with pytest.raises(solver.DirectError) as e:
        corridor_io.unpack(data, 63, rec.p_max)             # more polytopes than the caller can hold
    assert e.value.status == abi.DIRECT_ERR_UNSUPPORTED
    with pytest.raises(solver.DirectError) as e:
        corridor_io.unpack(data, 64, int(rec.n_planes.max()) - 1)
    assert e.value.status == abi.DIRECT_ERR_UNSUPPORTED
    huge = struct.pack("<iI", 1, 0xFFFFFFFF) + data[8:]     # absurd array length must not be trusted
    with pytest.raises(solver.DirectError):
        corridor_io.unpack(huge, 64, rec.p_max)


def test_recording_round_trip(rec, tmp_path):
    empty = corridor_io.Corridor(5, np.zeros(0, np.int32), np.zeros((0, 4, 4)), np.zeros((0, 3)), np.zeros((0, 3)))
    short = corridor_io.Corridor(-3, rec.n_planes[:3], rec.planes[:3], rec.seeds[:3], rec.centers[:3])
    f = tmp_path / "rec.bin"
    corridor_io.write_recording(str(f), [rec, empty, short])
    back = corridor_io.read_recording(str(f), 64, rec.p_max)
    assert [c.path_id for c in back] == [17, 5, -3] and [c.n_seg for c in back] == [64, 0, 3]
    assert corridor_io.pack(back[0]) == golden_bytes() and corridor_io.pack(back[2]) == corridor_io.pack(short)


def test_replay_batch_follows_the_reference_protocol(rec):
    """TRP:316-320 replays poly counts 2..64; TRP:806-811 takes the first n polytopes, start / goal =
    centers of the first / last of them; TRP:823 allocates the durations from start, seeds[1..n-1], goal."""
    b = corridor_io.replay_batch(rec, n_first=2)
    assert b.batch == 63 and b.n_seg_max == 64 and (b.n_seg == np.arange(2, 65)).all()
    for i in (0, 7, 62):
        n = i + 2
        assert np.array_equal(b.x0[i], np.r_[rec.centers[0], np.zeros(6)])
        assert np.array_equal(b.xd[i], np.r_[rec.centers[n - 1], np.zeros(6)])
        assert (b.n_planes[i, :n] == rec.n_planes[:n]).all() and np.array_equal(b.planes[i, :n], rec.planes[:n])
        assert np.array_equal(b.seeds[i, :n], rec.seeds[:n])
        T = problems.time_allocation(np.array([n], np.int32), rec.centers[None, 0], rec.centers[None, n - 1],
                                     rec.seeds[None, :n])
        assert np.allclose(b.T0[i, :n], T[0], rtol=1e-14) and (b.T0[i, n:] == 0).all()
    with pytest.raises(solver.DirectError):
        corridor_io.replay_batch(rec, n_first=2, batch=64)  # "no enough recorded polyhedrons" (TRP:802-804)


def test_unpack_survives_random_corruption(rec):
    """A wire-format reader must never read out of bounds: random byte flips, truncations and garbage
    either decode to something self-consistent or are rejected with an error code."""
    rng = np.random.default_rng(123)
    data = bytearray(golden_bytes())
    for trial in range(400):
        buf = bytearray(data)
        kind = trial % 4
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                buf[int(rng.integers(0, len(buf)))] = int(rng.integers(0, 256))
        elif kind == 1:
            buf = buf[:int(rng.integers(0, len(buf)))]
        elif kind == 2:
            pos = int(rng.integers(0, len(buf) - 4))
            buf[pos:pos + 4] = struct.pack("<I", int(rng.integers(0, 2 ** 32)))
        else:
            buf = bytearray(rng.integers(0, 256, int(rng.integers(0, 4000)), dtype=np.uint8).tobytes())
        try:
            cor, used = corridor_io.unpack(bytes(buf), 64, rec.p_max)
        except solver.DirectError as e:
            assert e.status in (abi.DIRECT_ERR_INVALID, abi.DIRECT_ERR_UNSUPPORTED)
            continue
        assert 0 <= cor.n_seg <= 64 and used <= len(buf) and (cor.n_planes <= rec.p_max).all()
        assert used == 8 + sum(52 + 32 * int(m) for m in cor.n_planes)
