"""CPU model of the fused qkv + attention kernel's LDS schedule (easy_vitpose_amd/csrc/qkvattn.hip): the operand ring of its GEMM phase, the hand-over of
q / k / v through the SAME 144 KiB, and the next tile's first K-tile streaming in under the softmax / P V of the current one.  The order of LDS-DMA issues,
counted `vmcnt` waits, barriers, fragment reads, epilogue stores and attention reads of BOTH wave groups is replayed and checked mechanically:

  RAW   a fragment read of ring slot (buffer, X0 | X1 | W0 | W1) finds exactly the K-tile it expects, every wave's pieces retired by a counted wait at least
        one barrier earlier; an attention read of Q / K / V finds the epilogue's stores of THIS tile, at least one barrier earlier;
  WAR   a slot is restaged (DMA) or overwritten (epilogue stores) only after both groups are past every read of its previous content, one barrier earlier;
  DMA   no LDS-DMA piece is still in flight into a region when the epilogue stores q / k / v there (a late piece would land on top of them).

LDS map (QA in qkvattn.hip): ring buffer b at b * 73 728: X0 | X1 (24 576 each) | W0 (16 384) | W1 (8 192); attention layout [Q0 K0 Q1 K1 V0 V1] in six slots
of 24 576: Q0 = b0.X0, K0 = b0.X1, Q1 = b0.W0 + b0.W1, K1 = b1.X0, V0 = b1.X1, V1 = b1.W0 + b1.W1 -- so ring buffer 0 and b1.X0 lie inside Q / K, which are dead
once every wave has its scores, and nothing of V is touched before the tile's last barrier.

A transcription of the kernel's control flow (lines cited), not the kernel: the race screen of the real code is its bit-identity with the two-launch path on
the GPU (tests/test_gpu_api.py::test_fused_qkv_attention_is_bit_identical).  What this pins is the ARGUMENT -- the round-4 change that stopped the tile's last two
K-tiles from re-fetching (two new K-tile modes with their own wait counts) was transcribed here first.  Mutations at the end check that the checker bites."""
import itertools

import pytest

CELLS = [(b, s) for b in (0, 1) for s in ('X0', 'X1', 'W0', 'W1')]
PIECES = {'X0': 3, 'X1': 3, 'W0': 2, 'W1': 1}      # qkvattn.hip issue(): glds16 calls per wave and slot
REGION = {('Q', 0): [(0, 'X0')], ('K', 0): [(0, 'X1')], ('Q', 1): [(0, 'W0'), (0, 'W1')], ('K', 1): [(1, 'X0')],
          ('V', 0): [(1, 'X1')], ('V', 1): [(1, 'W0'), (1, 'W1')]}
NKEEP = 9                                          # qkvattn.hip: DMA pieces of one LA (3) + one LB (6)


def program(group, ntiles, nk, mut=None):
    """Event list of one wave group (0 = waves 0-3 = crop 0 in the attention phase, 1 = waves 4-7).  Events: ('issue', cell, tile, kt, pieces), ('wait', n),
    ('bar',), ('read', cell, version), ('vm', n) = other vector-memory operations in the same in-order counter, ('store', cell, version) = epilogue LDS stores."""
    ev = []

    def issue(buf, slot, tile, kt):
        ev.append(('issue', (buf, slot), tile, kt, PIECES[slot]))

    def ring_start(tile):                              # qkvattn.hip ring_start
        for s in ('W0', 'X0', 'W1', 'X1'):
            issue(0, s, tile, 0)
        for s in ('W0', 'X0', 'W1'):
            issue(1, s, tile, 1)
        ev.append(('wait', 6))
        ev.append(('bar',))
        if group == 1:
            ev.append(('bar',))                        # stagger: waves 4-7 one barrier behind

    def ktile(buf, mode, tile, t, ka, kb):             # qkvattn.hip ktile(): K-tile t of `tile` in ring buffer `buf`
        for s in ('W0', 'W1', 'X0'):                   # LA
            ev.append(('read', (buf, s), ('ring', tile, t)))
        if mode != 3:
            issue(buf ^ 1, 'X1', tile, ka)
        if mode == 3:
            ev.append(('wait', 0))                     # last K-tile: nothing left to fetch, LA drains the queue (X1 of this K-tile)
        elif mode != 1:
            ev.append(('wait', NKEEP))
        ev.append(('bar',))
        ev.append(('bar',))                            # MA
        ev.append(('read', (buf, 'X1'), ('ring', tile, t)))   # LB
        if mode < 2:
            for s in ('W0', 'X0', 'W1'):
                issue(buf, s, tile, kb)
            ev.append(('wait', NKEEP))
        elif mode == 2:
            ev.append(('wait', 3 + (1 if mut == 'tail+1' else 0)))   # second-last K-tile: all but this LA's three X1 pieces
        ev.append(('bar',))
        ev.append(('bar',))                            # MB

    ring_start(0)
    for tile in range(ntiles):
        has_next = tile + 1 < ntiles
        ktile(0, 1, tile, 0, 1, 2)
        ktile(1, 0, tile, 1, 2, 3)
        for kt in range(2, nk - 2, 2):
            ktile(0, 0, tile, kt, kt + 1, kt + 2)
            ktile(1, 0, tile, kt + 1, kt + 2, kt + 3)
        if mut in ('old_tail', 'old_tail_no_drain'):   # the tail before the round-4 change: K-tiles 0 / 1 of the SAME tile once more (valid, never read)
            ktile(0, 0, tile, nk - 2, nk - 1, 0)
            ktile(1, 0, tile, nk - 1, 0, 1)
        else:
            ktile(0, 2, tile, nk - 2, nk - 1, 0)
            ktile(1, 3, tile, nk - 1, 0, 0)
        ev.append(('vm', 12))                          # epilogue operands: bias, row sums (3 + 3 loads), six row statistics
        if mut != 'old_tail_no_drain':
            ev.append(('wait', 0))
        if group == 0:
            ev.append(('bar',))                        # undo the stagger
        ev.append(('bar',))                            # __syncthreads: every wave is done with the ring
        ev.append(('vm', 6))                           # crop 1's six row statistics
        for reg in REGION:                             # qkv epilogue: every wave stores rows of both crops into Q / K / V
            for cell in REGION[reg]:
                ev.append(('store', cell, ('att', tile)))
        ev.append(('bar',))                            # __syncthreads
        for what in ('Q', 'K'):                        # scores: waves 0-3 crop 0, waves 4-7 crop 1
            for cell in REGION[(what, group)]:
                ev.append(('read', cell, ('att', tile)))
        ev.append(('bar',))                            # __syncthreads: every wave has its scores, Q and K are dead
        if has_next:                                   # the next tile's K-tile 0 and X0 of its K-tile 1 stream in under softmax / P V
            for s in ('W0', 'X0', 'W1', 'X1'):
                issue(0, s, tile + 1, 0)
            issue(1, 'X0', tile + 1, 1)
            if mut == 'early_v':
                issue(1, 'X1', tile + 1, 1)            # mutation: a slot inside V0 restaged while P V still reads V
        for cell in REGION[('V', group)]:
            ev.append(('read', cell, ('att', tile)))
        ev.append(('vm', 6))                           # y stores
        if not has_next:
            break
        ev.append(('bar',))                            # __syncthreads: every wave is done reading V
        issue(1, 'W0', tile + 1, 1)                    # completes ring_start's state (prefetched)
        issue(1, 'W1', tile + 1, 1)
        ev.append(('wait', 6))
        ev.append(('bar',))
        if group == 1:
            ev.append(('bar',))
    ev.append(('wait', 0))                             # (end of the kernel: s_endpgm retires everything)
    return ev


def check(ntiles, nk, mut=None):
    progs = [program(g, ntiles, nk, mut) for g in (0, 1)]
    segs = []
    for p in progs:                                    # segment k of a group runs between global barriers k and k + 1
        s, cur = [], []
        for e in p:
            if e[0] == 'bar':
                s.append(cur)
                cur = []
            else:
                cur.append(e)
        s.append(cur)
        segs.append(s)
    assert len(segs[0]) == len(segs[1]), 'the two groups must execute the same number of barriers'
    queue = [[], []]                                   # per group: outstanding vector-memory operations, oldest first
    issued = {}                                        # cell -> {group: (version, epoch)}
    landed = {}                                        # (cell, group) -> (version, epoch of the retiring wait)
    stored = {}                                        # cell -> {group: (version, epoch)}   epilogue stores
    reads = {}                                         # cell -> [(version, epoch, group)]
    errors = []
    for epoch in range(len(segs[0])):
        for g in (0, 1):
            for e in segs[g][epoch]:
                if e[0] == 'issue':
                    _, cell, tile, kt, n = e
                    ver = ('ring', tile, kt)
                    for (v, ep, gg) in reads.get(cell, []):
                        if v != ver and ep >= epoch:
                            errors.append(f'WAR: group {g} restages {cell} with {ver} in epoch {epoch}, group {gg} reads {v} in epoch {ep}')
                    issued.setdefault(cell, {})[g] = (ver, epoch)
                    stored.pop(cell, None)
                    queue[g] += [(cell, ver, i == n - 1) for i in range(n)]
                elif e[0] == 'vm':
                    queue[g] += [(None, None, False)] * e[1]
                elif e[0] == 'wait':
                    n = e[1]
                    done, queue[g] = (queue[g][:-n], queue[g][-n:]) if n else (queue[g], [])
                    for cell, ver, last in done:
                        if cell is not None and last:
                            landed[(cell, g)] = (ver, epoch)
                elif e[0] == 'store':
                    _, cell, ver = e
                    for gg in (0, 1):
                        if any(c == cell for c, _, _ in queue[gg]):
                            errors.append(f'DMA: group {g} stores {ver} into {cell} in epoch {epoch} while a piece of group {gg} is still in flight into it')
                        iv, lv = issued.get(cell, {}).get(gg), landed.get((cell, gg))
                        if iv is not None and (lv is None or lv[0] != iv[0] or lv[1] >= epoch):
                            errors.append(f'DMA: group {g} stores {ver} into {cell} in epoch {epoch}; group {gg}\'s {iv} retired {lv}')
                    for (v, ep, gg) in reads.get(cell, []):
                        if v != ver and ep >= epoch:
                            errors.append(f'WAR: group {g} stores {ver} into {cell} in epoch {epoch}, group {gg} reads {v} in epoch {ep}')
                    stored.setdefault(cell, {})[g] = (ver, epoch)
                    issued.pop(cell, None)
                elif e[0] == 'read':
                    _, cell, ver = e
                    if ver[0] == 'ring':
                        for gg in (0, 1):
                            iv, lv = issued.get(cell, {}).get(gg), landed.get((cell, gg))
                            if iv is None or iv[0] != ver:
                                errors.append(f'RAW: group {g} reads {cell} expecting {ver} in epoch {epoch}, group {gg} last issued {iv}')
                            elif lv is None or lv[0] != ver or lv[1] >= epoch:
                                errors.append(f'RAW: group {g} reads {cell} = {ver} in epoch {epoch}, group {gg}\'s pieces retired {lv}')
                    else:
                        for gg in (0, 1):
                            sv = stored.get(cell, {}).get(gg)
                            if sv is None or sv[0] != ver or sv[1] >= epoch:
                                errors.append(f'RAW: group {g} reads {cell} expecting {ver} in epoch {epoch}, group {gg} stored {sv}')
                    reads.setdefault(cell, []).append((ver, epoch, g))
    for g in (0, 1):
        assert not queue[g], 'operations left in flight at kernel end'
    return errors


@pytest.mark.parametrize('ntiles,nk', list(itertools.product((1, 2, 3, 6), (4, 6, 12, 16, 20))))
def test_fused_kernel_lds_schedule_has_no_hazard(ntiles, nk):
    """every tile count per workgroup (1 .. 6 at the BASELINE batch) x every K depth (D = 256 .. 1280)"""
    errors = check(ntiles, nk)
    assert not errors, errors[:5]


def test_the_tail_of_round_3_was_hazard_free_too():
    """the re-fetching tail (K-tiles 0 / 1 of the same tile, valid and never read) the round-4 change replaced: slower, not wrong"""
    assert not check(3, 12, 'old_tail')


@pytest.mark.parametrize('mut,kind', [('tail+1', 'RAW'), ('old_tail_no_drain', 'DMA'), ('early_v', 'WAR')])
def test_the_checker_bites(mut, kind):
    """a second-last K-tile that leaves one piece too many in flight (the last K-tile's operands are read before they land); the round-3 tail (run-ahead DMAs in
    flight at the tile's end) with an epilogue that does not drain the queue (a piece lands on top of q / k / v); a slot inside V restaged under the P V products"""
    errors = check(3, 12, mut)
    assert errors and any(e.startswith(kind) for e in errors), (mut, errors[:3])
