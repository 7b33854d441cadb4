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
        # a payload beyond the slot travels in pieces (round 2: it failed with RH_E_CAPACITY on the overflowing rank
        # while the others waited for their time-out)
        big = bytes([(rank * 7 + i * 13) & 255 for i in range(3 * (1 << 16) + 17)])
        allp = g.allgather(big)
        big_ok = all(allp[r] == bytes([(r * 7 + i * 13) & 255 for i in range(3 * (1 << 16) + 17)]) for r in range(world))
        g.close()
        q.put((rank, ok, "big payload ok" if big_ok else "big payload differs"))
    except Exception as e:   # noqa: BLE001
        q.put((rank, False, repr(e)))


@pytest.mark.parametrize("world,rounds", [(2, 300), (4, 300), (32, 12)])
def test_mp_exchange_rounds(world, rounds):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    name = "/rh_mp_xchg_%d_%d" % (os.getpid(), world)
    # (32 ranks: the flags of ranks >= 30 used to lie inside rank 0's slot -- ADVICE round 2)
    procs = [ctx.Process(target=_rank, args=(r, world, name, rounds, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    for rank, ok, too_big in res:
        assert ok, (rank, too_big)
        assert too_big == "big payload ok", (rank, too_big)
    assert not os.path.exists("/dev/shm" + name)     # rank 0 unlinked the name once everybody had mapped it


def test_mp_open_rejects_bad_arguments():
    import ransac_jl_amd as R
    with pytest.raises(R.RansacHipError):
        R.MpGroup("/rh_bad", 3, 2)
    g = R.MpGroup("/rh_mp_solo_%d" % os.getpid(), 0, 1)
    assert g.allgather(b"abc") == [b"abc"]
    g.close()
