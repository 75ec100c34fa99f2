"""Host side of the block-paged KV cache (the vLLM engine's layout for the reference's serving path).

The device side is `ChattsKvCache.block_table` (include/chatts_amd.h): a layer's K / V live in a pool of fixed-size blocks
`[n_blocks, n_kv, block_size, 128]`, and position j of the sequence in cache slot s is row `j % block_size` of block
`table[s][j // block_size]`.  This module owns the table: which blocks a slot holds, which are free, and which resident
(finished, kept for prefix reuse) slots may be evicted when a new request needs room.  Pure Python, no device work - the model
mirrors a changed row into its device table (`ChatTSForCausalLM.reserve_kv`).

Policy: a request reserves `ceil((prompt + max_new_tokens) / block_size)` blocks at admission (nothing is allocated during
decode, so a captured decode graph never sees the table change under it); blocks return to the pool when the slot is
re-used or evicted, not when the request finishes - the finished sequence's K/V rows stay resident for prefix reuse.
Prefix reuse across slots SHARES whole blocks by reference count (`adopt`) instead of copying rows: a request whose prompt
starts with what another slot holds points its first blocks at the same memory and prefills from the next block boundary;
a slot never writes a shared block - `reserve(private_from=...)` swaps the shared blocks it is about to overwrite for private
ones first.
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
        self.refs = [0] * self.n_blocks                       # table rows that hold the block (> 1: a shared prefix block)
        self.active = set()                                   # slots with a running request: never evicted
        self.evictions = 0
        self.last_replaced = set()                            # logical indices the last reserve() swapped for private blocks

    # ---- queries --------------------------------------------------------------------------------------------------------
    def blocks_for(self, n_tokens):
        return (max(int(n_tokens), 0) + self.block_size - 1) // self.block_size

    def capacity_tokens(self, slot):
        return len(self.rows[slot]) * self.block_size

    def _freeable(self, slots):
        """blocks that return to the free list if every row of `slots` is released (shared blocks only when all holders go)"""
        cnt = {}
        for s in slots:
            for b in self.rows[s]:
                cnt[b] = cnt.get(b, 0) + 1
        return sum(1 for b, c in cnt.items() if c == self.refs[b])

    def available(self, for_slot=None):
        """blocks a reservation for `for_slot` could obtain: the free ones, the slot's own, every inactive slot's"""
        return len(self.free) + self._freeable([s for s in range(self.n_slots) if s == for_slot or s not in self.active])

    def fits(self, token_counts):
        """could requests of these sizes (prompt + new tokens each) all be admitted into free slots right now?  (Counts every
        request as if it shared nothing: prefix sharing only ever needs fewer blocks.)"""
        need = sum(self.blocks_for(t) for t in token_counts)
        return need <= len(self.free) + self._freeable([s for s in range(self.n_slots) if s not in self.active])

    def shared_blocks(self):
        return sum(1 for r in self.refs if r > 1)

    # ---- changes --------------------------------------------------------------------------------------------------------
    def _take(self):
        b = self.free.pop()
        self.refs[b] = 1
        return b

    def _drop(self, b):
        self.refs[b] -= 1
        if self.refs[b] == 0:
            self.free.append(b)

    def reserve(self, slot, n_tokens, protect=(), evict_order=None, on_evict=None, private_from=None):
        """Make slot's row cover n_tokens positions and mark the slot active.  private_from = first logical block the request will
        WRITE (None: nothing is known, same as 0 for an empty row): every block of the row from there on that is shared with
        another slot is swapped for a private one (`last_replaced`; its old content is NOT copied - the caller recomputes those
        positions).  Missing blocks come from the free list, then from inactive slots (never `slot`, `protect` or active ones) in
        `evict_order` (default: fewest blocks first), whose whole row is released; `on_evict(victim)` lets the owner forget what
        was resident there.  Nothing changes when the pool cannot cover the request (KvPoolExhausted).  -> True when the row changed."""
        need = self.blocks_for(n_tokens)
        if need > self.blocks_per_slot:
            raise ValueError(f"{n_tokens} positions need {need} blocks, a slot's table row holds {self.blocks_per_slot}")
        row = self.rows[slot]
        swap = [] if private_from is None else [i for i in range(max(int(private_from), 0), len(row)) if self.refs[row[i]] > 1]
        missing = max(need - len(row), 0) + len(swap)
        self.last_replaced = set()
        if missing > len(self.free):
            keep = set(protect) | {slot} | self.active
            victims = [s for s in (evict_order if evict_order is not None else
                                   sorted(range(self.n_slots), key=lambda s: len(self.rows[s]))) if s not in keep and self.rows[s]]
            if len(self.free) + self._freeable(victims) < missing:
                raise KvPoolExhausted(f"{n_tokens} positions need {missing} more blocks of {self.block_size}; {len(self.free)} free, "
                                      f"{self._freeable(victims)} evictable of {self.n_blocks}")
            for v in victims:
                if len(self.free) >= missing:
                    break
                self.release(v)
                self.evictions += 1
                if on_evict:
                    on_evict(v)
        for i in swap:
            if self.refs[row[i]] > 1:                         # (an eviction above may just have made it private)
                self.refs[row[i]] -= 1
                row[i] = self._take()
                self.last_replaced.add(i)
        while len(row) < need:
            row.append(self._take())
        self.active.add(slot)
        return missing > 0

    def adopt(self, slot, src, n_blocks):
        """Prefix sharing: the first n_blocks logical blocks of `slot` become the SAME physical blocks as `src`'s (reference
        counted; what `slot` held there goes back to the pool).  The caller guarantees that neither slot writes those positions
        again without going through reserve(private_from=...)."""
        if n_blocks > len(self.rows[src]) or n_blocks > len(self.rows[slot]):
            raise ValueError("adopt: a row is shorter than the shared prefix")
        changed = False
        for i in range(n_blocks):
            old, new = self.rows[slot][i], self.rows[src][i]
            if old == new:
                continue
            self.refs[new] += 1
            self.rows[slot][i] = new
            self._drop(old)
            changed = True
        return changed

    def retire(self, slot):
        """the slot's request finished: its blocks stay (prefix reuse) but may be evicted from now on"""
        self.active.discard(slot)

    def release(self, slot):
        for b in reversed(self.rows[slot]):
            self._drop(b)
        self.rows[slot] = []
        self.active.discard(slot)

    def check(self):
        """reference counts match the rows, and every block is either held or free (tests)"""
        cnt = [0] * self.n_blocks
        for r in self.rows:
            assert len(r) == len(set(r))                      # a row never holds a block twice
            for b in r:
                cnt[b] += 1
        assert cnt == self.refs
        held = {b for b, c in enumerate(cnt) if c}
        assert not (held & set(self.free)) and len(self.free) == len(set(self.free))
        assert sorted(held | set(self.free)) == list(range(self.n_blocks))
        return True
