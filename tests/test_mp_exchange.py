"""The host-side exchange of rh_ransac_mp (shared memory, no GPU): world processes, many rounds, payloads that change
size from round to round; every rank must see every rank's payload of the same round, and a bad group is an error."""
import multiprocessing as mp
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank(rank, world, name, rounds, q):
    try:
        sys.path.insert(0, ROOT)
        import ransac_jl_amd as R
        g = R.MpGroup(name, rank, world, slot_bytes=1 << 16)
        ok = True
        for it in range(rounds):
            size = 1 + (it * 37) % 5000
            mine = bytes([(rank * 31 + it + i) & 255 for i in range(size)])
            allp = g.allgather(mine)
            for r in range(world):
                ok &= allp[r] == bytes([(r * 31 + it + i) & 255 for i in range(size)])
        too_big = None
        try:
            g.allgather(b"x" * ((1 << 16) + 1))
        except R.RansacHipError as e:
            too_big = str(e)
        g.close()
        q.put((rank, ok, too_big))
    except Exception as e:   # noqa: BLE001
        q.put((rank, False, repr(e)))


@pytest.mark.parametrize("world", [2, 4])
def test_mp_exchange_rounds(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/rh_mp_xchg_%d_%d" % (os.getpid(), world)
    procs = [ctx.Process(target=_rank, args=(r, world, name, 300, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, ok, too_big in res:
        assert ok, (rank, too_big)
        assert too_big and "does not fit the exchange slot" in too_big
    assert not os.path.exists("/dev/shm" + name)     # rank 0 unlinked the name once everybody had mapped it


def test_mp_open_rejects_bad_arguments():
    import ransac_jl_amd as R
    with pytest.raises(R.RansacHipError):
        R.MpGroup("/rh_bad", 3, 2)
    g = R.MpGroup("/rh_mp_solo_%d" % os.getpid(), 0, 1)
    assert g.allgather(b"abc") == [b"abc"]
    g.close()
