"""The closed forms of the PAIR LAYOUT of the advection lists (claymore_amd/csrc/mpm_kernels.hpp: div_small, pair_chunk, pair_slice and the
placement rule of prepare_blocks_kernel's sort_chunk_pairs), restated in Python and checked exhaustively: they are what the
sort kernel and G2P2G must agree on without talking to each other - the sort places a record by (slot, member), G2P2G finds slice t by
(position, lanes with A, lanes with B) from nothing but the chunk's size n and its number of full pairs Pf.  (The HIP kernels themselves are
checked on the GPU: tests/test_parity_gpu.py::test_the_sort_writes_the_pair_layout_it_promises.)"""
import itertools

import numpy as np


KPAIR_TAIL_MERGE = 256      # mpm_kernels.hpp: the last chunk of a block absorbs a tail of up to this many records
CHUNK_MAX = 512 + KPAIR_TAIL_MERGE


def pair_chunks(size):
    """The chunks of a block's list in the pair layout: (first record, records) - 512 each, the last one up to CHUNK_MAX (mpm_kernels.hpp: pair_chunks / pair_chunk_records)."""
    nfull, tail = size >> 9, size & 511
    n = nfull + (1 if (tail > 0 and not (nfull >= 1 and tail <= KPAIR_TAIL_MERGE)) else 0)
    return [(512 * c, 512 if c + 1 < n else size - 512 * c) for c in range(n)]


def test_chunks_tile_every_block_size():
    for size in range(0, 8193):
        ch = pair_chunks(size)
        assert sum(n for _, n in ch) == size and all(0 < n <= CHUNK_MAX for _, n in ch) and all(n == 512 for _, n in ch[:-1])
        assert [a for a, _ in ch] == [512 * c for c in range(len(ch))] and len(ch) <= 16
        assert sum(-(-n // 128) for _, n in ch) == -(-size // 128)                 # merging the tail never costs a slice
        if size > 512 and 0 < size % 512 <= KPAIR_TAIL_MERGE:
            assert ch[-1][1] == 512 + size % 512


def div_small(n, d):
    """n // d for 0 <= n <= 1024, 1 <= d <= 8 by a 16-bit reciprocal (the kernel's table: ceil(65536 / d) for d = 2..8)."""
    inv = 65536 if d == 1 else -(-65536 // d)
    return (n * inv) >> 16


def pair_chunk(n, pf):
    S = (n + 127) >> 7
    n1 = n - 2 * pf
    x = max(0, pf + n1 - 64 * S)
    px, L = pf + x, pf + n1 - x
    qb, qa = div_small(px, S), div_small(L, S)
    return dict(n=n, pf=pf, S=S, x=x, px=px, L=L, qb=qb, rb=px - qb * S, qa=qa, ra=L - qa * S)


def pair_slice(c, t):
    ca = c["qa"] + (1 if t < c["ra"] else 0)
    cb = c["qb"] + (1 if t < c["rb"] else 0)
    pos = t * (c["qa"] + c["qb"]) + min(t, c["ra"]) + min(t, c["rb"])
    return pos, ca, cb


def test_div_small_is_exact_on_its_domain():
    for d in range(1, 9):
        for n in range(0, 1025):
            assert div_small(n, d) == n // d, (n, d)


def test_slices_tile_the_chunk_for_every_size_and_pair_count():
    """For every chunk size n <= CHUNK_MAX and every feasible number of full pairs: the S = ceil(n / 128) slices are dense and in order, never wider than
    64 lanes, their B lanes a prefix of their A lanes, and together they hold exactly n records; the mismatched slots are X = max(0, Pf + n1 - 64 S)."""
    for n in range(1, CHUNK_MAX + 1):
        for pf in range(0, n // 2 + 1):
            c = pair_chunk(n, pf)
            at = 0
            for t in range(c["S"]):
                pos, ca, cb = pair_slice(c, t)
                assert pos == at and 0 <= cb <= ca <= 64, (n, pf, t)
                at += ca + cb
            assert at == n and c["L"] <= 64 * c["S"] and c["L"] + c["px"] == n
            assert c["x"] == max(0, pf + (n - 2 * pf) - 64 * c["S"]) and 2 * c["x"] <= n - 2 * pf        # the mismatched slots are made of singles only
            assert c["S"] == -(-n // 128)


def place(c, slot, member):
    """Position of a record in the chunk: its slot's slice p mod S, lane p // S; A's first, the B's behind them."""
    ln = div_small(slot, c["S"])
    d = slot - ln * c["S"]
    pos, ca, cb = pair_slice(c, d)
    assert ln < ca and (member == 0 or ln < cb)
    return pos + (ca if member else 0) + ln, d, ln


def test_the_placement_rule_is_a_bijection_and_keeps_a_keys_pairs_apart():
    """The sort's rule on random key histograms: pair slots in key-major order, then the mismatched slots (the last 2 X singles two by two), then the
    singles; every record lands on its own position in [0, n); a slice holds a key's pair slots in neighbouring lanes at most twice unless the key has
    more than 2 S pairs (the two arenas take one each); a single takes the arena its key's pair slot of the same slice does not use."""
    rng = np.random.default_rng(20)
    for trial in range(400):
        nkeys = int(rng.integers(1, 80))
        counts = rng.poisson(rng.uniform(0.5, 12.0 if trial % 3 else 18.0), nkeys) + (rng.random(nkeys) < 0.3)
        counts = counts[counts > 0].astype(int)
        limit = 512 if trial % 3 else CHUNK_MAX                      # (every third trial: a merged last chunk)
        while counts.sum() > limit:
            counts[np.argmax(counts)] -= 1
        counts = counts[counts > 0]
        n = int(counts.sum())
        if n == 0:
            continue
        pf = int((counts // 2).sum())
        n1 = int((counts % 2).sum())
        c = pair_chunk(n, pf)
        n1s = c["L"] - c["px"]
        taken = np.zeros(n, bool)
        pp = sp = 0
        pair_lanes = {}                                             # (slice, key) -> [(lane, arena)]
        singles = []
        for k, nk in enumerate(counts):
            for r in range(int(nk)):
                if r < (nk & ~1):
                    slot, member = pp + (r >> 1), r & 1
                else:
                    j = sp
                    slot, member = (c["px"] + j, 0) if j < n1s else (pf + ((j - n1s) >> 1), (j - n1s) & 1)
                pos, d, ln = place(c, slot, member)
                assert not taken[pos], (trial, k, r)
                taken[pos] = True
                if r < (nk & ~1) and member == 0:
                    pair_lanes.setdefault((d, k), []).append((ln, ln & 1))
                if r >= (nk & ~1) and member == 0:
                    singles.append((k, d, pp, int(nk) >> 1))
            pp += int(nk) >> 1
            sp += int(nk) & 1
        assert taken.all() and pp == pf and sp == n1
        for (d, k), lanes in pair_lanes.items():
            ls = sorted(l for l, _ in lanes)
            assert ls == list(range(ls[0], ls[0] + len(ls)))        # a key's pair slots of one slice: neighbouring lanes (the wrap-around rule)
            assert len(ls) <= -(-int(counts[k] // 2) // c["S"])      # no more of them than the key's share of the slice
        for k, d, ppk, mk in singles:                               # the arena rule of the sort, against the lanes above
            i0 = (d - ppk % c["S"]) % c["S"]
            if i0 < mk:
                used = (div_small(ppk + i0, c["S"])) & 1
                lanes = pair_lanes[(d, k)]
                assert lanes[0][1] == used                          # ... is the arena of the key's FIRST pair slot in that slice
