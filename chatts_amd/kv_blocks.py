"""Host side of the block-paged KV cache (the vLLM engine's layout for the reference's serving path).

The device side is `ChattsKvCache.block_table` (include/chatts_amd.h): a layer's K / V live in a pool of fixed-size blocks
`[n_blocks, n_kv, block_size, 128]`, and position j of the sequence in cache slot s is row `j % block_size` of block
`table[s][j // block_size]`.  This module owns the table: which blocks a slot holds, which are free, and which resident
(finished, kept for prefix reuse) slots may be evicted when a new request needs room.  Pure Python, no device work - the model
mirrors a changed row into its device table (`ChatTSForCausalLM.reserve_kv`).

Policy: a request reserves `ceil((prompt + max_new_tokens) / block_size)` blocks at admission (nothing is allocated during
decode, so a captured decode graph never sees the table change under it); blocks return to the pool when the slot is
re-used or evicted, not when the request finishes - the finished sequence's K/V rows stay resident for prefix reuse.
"""


class KvPoolExhausted(RuntimeError):
    """the pool cannot cover the reservation even after evicting every resident, inactive slot"""


class BlockPool:
    def __init__(self, n_blocks, block_size, n_slots, blocks_per_slot):
        if block_size < 64 or block_size & (block_size - 1):
            raise ValueError(f"block_size {block_size} must be a power of two >= 64 (the kernels walk keys in tiles of up to 64)")
        if n_blocks < 1 or n_slots < 1 or blocks_per_slot < 1:
            raise ValueError("n_blocks, n_slots and blocks_per_slot must be positive")
        self.n_blocks, self.block_size = int(n_blocks), int(block_size)
        self.n_slots, self.blocks_per_slot = int(n_slots), int(blocks_per_slot)
        self.free = list(range(self.n_blocks))[::-1]          # pop() hands out block 0 first
        self.rows = [[] for _ in range(self.n_slots)]         # blocks of each slot in logical order
        self.active = set()                                   # slots with a running request: never evicted
        self.evictions = 0

    # ---- queries --------------------------------------------------------------------------------------------------------
    def blocks_for(self, n_tokens):
        return (max(int(n_tokens), 0) + self.block_size - 1) // self.block_size

    def capacity_tokens(self, slot):
        return len(self.rows[slot]) * self.block_size

    def available(self, for_slot=None):
        """blocks a reservation for `for_slot` could obtain: the free ones, the slot's own, every inactive slot's"""
        n = len(self.free)
        for s, row in enumerate(self.rows):
            if s == for_slot or s not in self.active:
                n += len(row)
        return n

    def fits(self, token_counts):
        """could requests of these sizes (prompt + new tokens each) all be admitted into free slots right now?"""
        need = sum(self.blocks_for(t) for t in token_counts)
        return need <= len(self.free) + sum(len(r) for s, r in enumerate(self.rows) if s not in self.active)

    # ---- changes --------------------------------------------------------------------------------------------------------
    def reserve(self, slot, n_tokens, protect=(), evict_order=None, on_evict=None):
        """Make slot's row cover n_tokens positions and mark the slot active.  Missing blocks come from the free list, then from
        inactive slots (never `slot`, `protect` or active ones) in `evict_order` (default: fewest blocks first), whose whole row
        is released; `on_evict(victim)` lets the owner forget what was resident there.  -> True when the row changed."""
        need = self.blocks_for(n_tokens)
        if need > self.blocks_per_slot:
            raise ValueError(f"{n_tokens} positions need {need} blocks, a slot's table row holds {self.blocks_per_slot}")
        row = self.rows[slot]
        missing = need - len(row)
        if missing > 0 and missing > len(self.free):
            keep = set(protect) | {slot} | self.active
            victims = [s for s in (evict_order if evict_order is not None else
                                   sorted(range(self.n_slots), key=lambda s: len(self.rows[s]))) if s not in keep and self.rows[s]]
            if len(self.free) + sum(len(self.rows[s]) for s in victims) < missing:
                raise KvPoolExhausted(f"{n_tokens} positions need {missing} more blocks of {self.block_size}; {len(self.free)} free, "
                                      f"{sum(len(self.rows[s]) for s in victims)} evictable of {self.n_blocks}")
            for v in victims:
                if len(self.free) >= missing:
                    break
                self.release(v)
                self.evictions += 1
                if on_evict:
                    on_evict(v)
        for _ in range(max(missing, 0)):
            row.append(self.free.pop())
        self.active.add(slot)
        return missing > 0

    def retire(self, slot):
        """the slot's request finished: its blocks stay (prefix reuse) but may be evicted from now on"""
        self.active.discard(slot)

    def release(self, slot):
        self.free.extend(reversed(self.rows[slot]))
        self.rows[slot] = []
        self.active.discard(slot)

    def check(self):
        """every block is owned exactly once (tests)"""
        owned = [b for r in self.rows for b in r]
        assert len(owned) == len(set(owned)) and not (set(owned) & set(self.free))
        assert sorted(owned + self.free) == list(range(self.n_blocks))
        return True
