"""The staging pools behind search handles (hip/engine.py: _PinnedPool, _PinSlot, _FlagPool) — pure host logic, run here with a
stand-in for torch's pinned allocations.  ADVICE r5 (high) was a 6-slot RING handing a slot out again while an earlier answer
still lived in it; the pools are free lists: a slot is somebody's until it is released or its last reference dies."""
import gc
import threading

import torch

from rag_arc_amd.hip import engine as E


class _Torch:
    """torch with `pin_memory=True` accepted on a CPU-only build."""
    int64, float32, int32 = torch.int64, torch.float32, torch.int32

    @staticmethod
    def empty(n, dtype=None, pin_memory=False):
        return torch.empty(n, dtype=dtype)

    @staticmethod
    def zeros(n, dtype=None, pin_memory=False):
        return torch.zeros(n, dtype=dtype)


def test_a_slot_is_not_handed_out_again_while_it_is_held():
    pool = E._PinnedPool()
    held = [pool.acquire(_Torch, 256, 100) for _ in range(10)]          # ten answers alive at once (a 2560-query call)
    ptrs = {s.ids.data_ptr() for s in held}
    assert len(ptrs) == 10 and pool.allocated == 10
    for i, s in enumerate(held):
        ids, sc = s.views(256, 100)
        ids.fill_(i)
        sc.fill_(float(i))
    for i, s in enumerate(held):                                        # nobody wrote into anybody else's slot
        ids, sc = s.views(256, 100)
        assert int(ids.min()) == int(ids.max()) == i and float(sc.max()) == float(i)
    held[3].release()
    again = pool.acquire(_Torch, 256, 100)
    assert again.ids.data_ptr() in ptrs and pool.allocated == 10        # a released slot is reused, nothing new is allocated
    again.release()
    again.release()                                                     # (idempotent)
    assert len(pool._free) == 1


def test_a_dropped_handle_gives_its_slot_back_and_the_pool_stays_small():
    pool = E._PinnedPool()
    for _ in range(50):
        s = pool.acquire(_Torch, 8, 10)
        del s                                                           # last reference gone -> __del__ -> free list
        gc.collect()
    assert pool.allocated == 1 and len(pool._free) == 1
    big = pool.acquire(_Torch, 300, 1024)                               # larger than the free slot: a new one of the right size
    assert big.ids.numel() >= 300 * 1024 and pool.allocated == 2
    burst = [pool.acquire(_Torch, 8, 10) for _ in range(40)]
    for s in burst:
        s.release()
    assert len(pool._free) <= 16                                        # a burst's extra slots are not kept for ever


def test_slots_from_many_threads_never_alias():
    pool = E._PinnedPool()
    seen, lock, errors = [], threading.Lock(), []

    def work(tag):
        try:
            for _ in range(200):
                s = pool.acquire(_Torch, 16, 8)
                ids, _ = s.views(16, 8)
                ids.fill_(tag)
                with lock:
                    seen.append(s.ids.data_ptr())
                assert int(ids.min()) == int(ids.max()) == tag          # still ours after the others had their turn
                s.release()
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors and pool.allocated <= 8


def test_flag_words_are_a_free_list_not_a_ring():
    flags = E._FlagPool()
    words = [flags.acquire(_Torch) for _ in range(70)]                  # 70 launches in flight: more than one block of 64
    assert len({w.data_ptr() for w in words}) == 70
    for i, w in enumerate(words):
        w[0] = i
    assert [int(w[0]) for w in words] == list(range(70))
    for w in words[:10]:
        flags.give_back(w)
    reused = {flags.acquire(_Torch).data_ptr() for _ in range(10)}
    assert reused == {w.data_ptr() for w in words[:10]}
