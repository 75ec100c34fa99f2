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
