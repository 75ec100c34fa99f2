"""Host bookkeeping of the block-paged KV cache (chatts_amd/kv_blocks.py): no GPU."""
import pytest

from chatts_amd.kv_blocks import BlockPool, KvPoolExhausted


def test_reserve_extend_release():
    p = BlockPool(n_blocks=10, block_size=64, n_slots=3, blocks_per_slot=8)
    assert p.blocks_for(0) == 0 and p.blocks_for(1) == 1 and p.blocks_for(64) == 1 and p.blocks_for(65) == 2
    assert p.reserve(0, 130) is True and p.rows[0] == [0, 1, 2] and 0 in p.active
    assert p.reserve(0, 100) is False and p.rows[0] == [0, 1, 2]           # never shrinks, nothing changed
    assert p.reserve(0, 200) is True and p.rows[0] == [0, 1, 2, 3]         # extends in logical order
    assert p.capacity_tokens(0) == 256
    p.reserve(1, 64)
    assert p.rows[1] == [4] and len(p.free) == 5 and p.check()
    p.release(0)
    assert p.rows[0] == [] and len(p.free) == 9 and 0 not in p.active and p.check()
    with pytest.raises(ValueError):
        p.reserve(2, 64 * 9)                                               # more than a table row holds
    with pytest.raises(ValueError):
        BlockPool(4, 48, 1, 4)
    with pytest.raises(ValueError):
        BlockPool(4, 32, 1, 4)


def test_eviction_only_touches_inactive_unprotected_slots():
    p = BlockPool(n_blocks=8, block_size=64, n_slots=4, blocks_per_slot=8)
    p.reserve(0, 192); p.reserve(1, 128); p.reserve(2, 128)               # 3 + 2 + 2 = 7 of 8
    assert p.fits([64]) and not p.fits([128])                             # everything else is active
    with pytest.raises(KvPoolExhausted):
        p.reserve(3, 192)
    assert p.rows[3] == [] and p.check()
    p.retire(1); p.retire(2)
    assert p.fits([192, 64]) and not p.fits([192, 192])
    evicted = []
    p.reserve(3, 192, protect={2}, on_evict=evicted.append)               # needs 3: 1 free + slot 1's two (slot 2 is protected)
    assert evicted == [1] and p.rows[1] == [] and len(p.rows[2]) == 2 and len(p.rows[3]) == 3 and p.evictions == 1 and p.check()
    with pytest.raises(KvPoolExhausted):                                  # only the protected slot is left to evict
        p.reserve(1, 64, protect={2})
    p.reserve(1, 64)                                                      # unprotected now: slot 2 goes
    assert p.rows[2] == [] and p.rows[1] and p.check()


def test_evict_order_and_own_blocks_count():
    p = BlockPool(n_blocks=6, block_size=128, n_slots=3, blocks_per_slot=4)
    p.reserve(0, 256); p.retire(0)
    p.reserve(1, 256); p.retire(1)
    p.reserve(2, 256); p.retire(2)
    assert p.available(for_slot=2) == 6
    seen = []
    p.reserve(2, 512, evict_order=[1, 0, 2], on_evict=seen.append)        # needs 2 more: the first slot of the order suffices
    assert seen == [1] and len(p.rows[2]) == 4 and len(p.rows[0]) == 2 and p.check()
    # a re-used slot keeps its own blocks and only asks for the difference
    p.retire(2)
    assert p.reserve(2, 300) is False and 2 in p.active


def test_prefix_blocks_are_shared_by_reference_and_never_written_while_shared():
    p = BlockPool(n_blocks=10, block_size=64, n_slots=3, blocks_per_slot=6)
    p.reserve(0, 300); p.retire(0)                       # 5 blocks: a finished sequence, resident
    p.reserve(1, 200)                                    # 4 blocks of its own
    own = list(p.rows[1])
    assert p.adopt(1, 0, 3) is True                      # the first 3 blocks of slot 1 are now slot 0's
    assert p.rows[1][:3] == p.rows[0][:3] and p.rows[1][3] == own[3]
    assert [p.refs[b] for b in p.rows[0][:3]] == [2, 2, 2] and p.shared_blocks() == 3
    assert set(own[:3]) <= set(p.free)                   # what slot 1 held there went back to the pool
    assert p.adopt(1, 0, 3) is False and p.check()       # idempotent
    # the donor is evictable, but its shared blocks stay alive for the other holder
    assert p._freeable([0]) == 2 and p.fits([64 * 6]) and not p.fits([64 * 7])       # 4 free + 2 of the donor's own
    p.release(0)
    assert [p.refs[b] for b in p.rows[1][:3]] == [1, 1, 1] and p.shared_blocks() == 0 and p.check()
    # a slot that is about to overwrite from block 1 on gets private copies of the shared blocks there (block 0 stays shared)
    p.reserve(2, 128); p.adopt(2, 1, 2); p.retire(1)
    shared = list(p.rows[1][:2])
    assert p.rows[2][:2] == shared
    p.reserve(1, 200, private_from=1)
    assert p.last_replaced == {1} and p.rows[1][0] == shared[0] and p.rows[1][1] != shared[1]
    assert p.rows[2][:2] == shared and p.refs[shared[1]] == 1 and p.refs[shared[0]] == 2 and p.check()
    with pytest.raises(ValueError):
        p.adopt(2, 1, 5)                                 # longer than slot 2's row


def test_eviction_counts_shared_blocks_once_and_swap_needs_room():
    p = BlockPool(n_blocks=5, block_size=64, n_slots=3, blocks_per_slot=4)
    p.reserve(0, 192); p.retire(0)                       # 3 blocks
    p.reserve(1, 128); p.adopt(1, 0, 2); p.retire(1)     # shares 2 of them; its own 2 went back: 2 free, 3 held
    assert len(p.free) == 2 and p.shared_blocks() == 2 and p.check()
    assert p.fits([64 * 5]) and not p.fits([64 * 6])     # evicting both inactive slots frees everything exactly once
    p.reserve(2, 64 * 4)                                 # needs 4: 2 free + slot 1 frees nothing alone, slot 0 and 1 together free 3
    assert len(p.rows[2]) == 4 and p.check() and p.evictions >= 1
    # privatising a shared block is part of the same all-or-nothing reservation
    q = BlockPool(n_blocks=4, block_size=64, n_slots=2, blocks_per_slot=3)
    q.reserve(0, 128); q.reserve(1, 128); q.adopt(1, 0, 2)       # both rows hold the same 2 blocks; slot 1's own two are free again
    free0 = len(q.free)
    assert free0 == 2
    q.reserve(1, 128, private_from=0)                    # wants 2 private blocks: enough room?
    assert q.check() and len(q.rows[1]) == 2 and not (set(q.rows[1]) & set(q.rows[0])) and len(q.free) == free0 - 2
    r = BlockPool(n_blocks=2, block_size=64, n_slots=2, blocks_per_slot=2)
    r.reserve(0, 128); r.retire(0)
    r.rows[1] = list(r.rows[0]); r.refs = [2, 2]; r.active.add(1)  # both hold both blocks, nothing free
    before = ([list(x) for x in r.rows], list(r.refs), list(r.free))
    with pytest.raises(KvPoolExhausted):
        r.reserve(1, 128, private_from=0, protect={0})   # no room to privatise and the donor is protected: nothing changes
    assert ([list(x) for x in r.rows], list(r.refs), list(r.free)) == before
