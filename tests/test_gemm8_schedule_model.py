"""CPU model of the 8-phase kernel's operand ring (easy_vitpose_amd/csrc/gemm8.hip): the order of LDS-DMA issues, counted `vmcnt` waits, barriers and
fragment reads of BOTH wave groups is replayed and every hazard the DESIGN.md "Ordering argument" talks about is checked mechanically:

  RAW  a fragment read of slot (buffer, X0 | X1 | W0 | W1) must find exactly the K-tile it expects, with every wave's pieces of that slot retired by a
       counted wait that lies at least one barrier BEFORE the read (DMA data of other waves is only visible after wait + barrier);
  WAR  a slot may only be restaged once both wave groups have finished the reads of its previous content, at least one barrier earlier;
  the counted waits never count on a piece that was not issued (a too-large count silently waits for nothing).

The model is a transcription of the kernel's control flow (file:line cited below), not the kernel itself -- the race screen of the real code is the
bit-identity of its results with the 2-phase kernels on the GPU (tests/test_gpu_gemm_cfgs.py, tools/gemm8_check.py).  What this test pins is the
ARGUMENT: change the schedule in gemm8.hip, transcribe the change here first, and a broken count or a restage that comes one barrier too early shows up on
the CPU, for every tile count / K depth / tile width / epilogue kind, before a GPU sees it.  A mutation test at the end checks that the checker bites."""
import itertools

import pytest

SLOTS = ('X0', 'X1', 'W0', 'W1')


def program(group, ntiles, nk, bn, epi, mut=None, bm=256, fp8=False):
    """Event list of one wave group (0 = waves 0-3, 1 = waves 4-7).  Events: ('issue', slot, buf, tile, kt, pieces), ('wait', n), ('bar',),
    ('read', slot, buf, tile, kt), ('vm', n) = n other vector-memory operations (epilogue stores) entering the same in-order counter."""
    nw1 = 2 if bn == 256 else 1                       # gemm8_common.h G8<BN>: W1 is 128 (2 pieces per wave) or 64 rows (1 piece)
    pieces = {'X0': 2, 'X1': 2, 'W0': 2, 'W1': nw1}   # gemm8.hip issue(): glds16 calls per wave and slot
    nkeep, inflight = 6 + nw1, 4 + nw1                # gemm8.hip: NKEEP, G8::INFLIGHT
    if bm == 192:                                     # G8<256, 192>: an X half is 12 pieces -- waves 0-3 issue 2 of X0 and 1 of X1, waves 4-7 the reverse
        pieces['X0'], pieces['X1'] = (2, 1) if group == 0 else (1, 2)
        nkeep = 5 + nw1                               # the same for both groups (gemm8_common.h NKEEP)
        inflight = (4 if group == 0 else 3) + nw1     # gemm8.hip wait_lb(): one LB group of THIS wave group
    if fp8:                                           # gemm8f.hip: every X slot is followed by its scale dword, an (untracked, inline-asm) global load in the same
        pieces['X0'] = pieces['X1'] = 3                # in-order counter: NLA = 3, NLB = 5 + NW1; the scale registers are per-wave state, program order covers them
        nkeep, inflight = 8 + nw1, 5 + nw1
    if mut == 'nkeep+1':
        nkeep += 1
    if mut == 'inflight+1':
        inflight += 1
    ev = []
    issue_tile = [0]

    def issue(slot, buf, kt):
        ev.append(('issue', slot, buf, issue_tile[0], kt, pieces[slot]))

    def ring_start():                                 # gemm8.hip ring_start
        for s in ('W0', 'X0', 'W1', 'X1'):
            issue(s, 0, 0)
        for s in ('W0', 'X0', 'W1'):
            issue(s, 1, 1)
        ev.append(('wait', inflight))
        ev.append(('bar',))
        if group == 1:
            ev.append(('bar',))                       # stagger: waves 4-7 one barrier behind

    def ktile(buf, mode, tile, t, ka, kb, switch, nxt=None):    # gemm8.hip ktile(): K-tile t of `tile` lives in ring buffer `buf`
        for s in ('W0', 'W1', 'X0'):                  # LA: fragment reads
            ev.append(('read', s, buf, tile, t))
        issue('X1', buf ^ 1, ka)
        if mode != 1:
            ev.append(('wait', nkeep))
        ev.append(('bar',))                           # (lgkmcnt(0) in front of it: the reads are complete)
        ev.append(('bar',))                           # MA
        ev.append(('read', 'X1', buf, tile, t))       # LB
        if switch:
            issue_tile[0] = nxt                       # set_tile(next): from here on the ring fetches the next tile
        order = ('W0', 'X0', 'W1') if mut != 'early_x1' else ('W0', 'X0', 'W1', 'X1')
        for s in order:
            issue(s, buf, kb) if s != 'X1' else issue('X1', buf, kb)   # mutation: restage X1 of THIS buffer one segment early
        ev.append(('wait', inflight if mode == 2 else nkeep))
        ev.append(('bar',))
        ev.append(('bar',))                           # MB

    ring_start()
    for tile in range(ntiles):
        has_next = tile + 1 < ntiles
        ktile(0, 1, tile, 0, 1, 2, False)
        ktile(1, 0, tile, 1, 2, 3, False)
        for kt in range(2, nk - 2, 2):
            ktile(0, 0, tile, kt, kt + 1, kt + 2, False)
            ktile(1, 0, tile, kt + 1, kt + 2, kt + 3, False)
        # no next tile: the ring refetches K-tiles 0 / 1 of this tile (valid, never read)
        ktile(0, 0, tile, nk - 2, nk - 1, 0, True, tile + 1 if has_next else ('refetch', tile))
        ktile(1, 2, tile, nk - 1, 0, 1, False)
        if epi == 'resid_lds':                        # gemm8.hip: residual epilogue through LDS -- drain, re-align, staged passes, ring restart
            ev.append(('wait', 0))
            if group == 0:
                ev.append(('bar',))
            ev += [('bar',)] * 4
            ev.append(('lds_reuse',))                 # the staging passes overwrite the ring: nothing of it may be read afterwards without a restage
            ev += [('bar',)] * 2
            if has_next:
                ring_start()
        else:                                         # 16-bit / register residual epilogues: the ring runs on across the tile boundary
            if group == 0:
                ev.append(('bar',))                   # both groups run their epilogues together
            ev.append(('vm', 16 if epi == 'plain16' else 76))
            if group == 1:
                ev.append(('bar',))                   # stagger again
        if not has_next:
            break
    ev.append(('wait', 0))
    if epi != 'resid_lds' and group == 0:
        ev.append(('bar',))                           # pair the extra barrier of the staggered group
    return ev


def check(ntiles, nk, bn, epi, mut=None, bm=256, fp8=False):
    progs = [program(g, ntiles, nk, bn, epi, mut, bm, fp8) for g in (0, 1)]
    segs = []
    for p in progs:                                   # segment k of a group runs between global barriers k and k + 1
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
    queue = [[], []]                                  # per group: outstanding vector-memory operations, oldest first: (slot key or None, version)
    issued = {}                                       # (buf, slot) -> per group: (version, issue epoch)
    landed = {}                                       # (buf, slot, group) -> (version, epoch of the retiring wait)
    reads = {}                                        # (buf, slot) -> list of (version, epoch, group)
    errors = []
    for epoch in range(len(segs[0])):
        for g in (0, 1):
            for e in segs[g][epoch]:
                if e[0] == 'issue':
                    _, slot, buf, tile, kt, n = e
                    key, ver = (buf, slot), (tile, kt)
                    for (v, ep, gg) in reads.get(key, []):
                        if v != ver and ep >= epoch:
                            errors.append(f'WAR: group {g} restages {key} with {ver} in epoch {epoch}, group {gg} reads {v} in epoch {ep}')
                    issued.setdefault(key, {})[g] = (ver, epoch)
                    queue[g] += [(key, ver, i == n - 1) for i in range(n)]   # in-order retirement: the slot has landed when its LAST piece has
                elif e[0] == 'vm':
                    queue[g] += [(None, None, False)] * e[1]
                elif e[0] == 'wait':
                    n = e[1]
                    if n > len(queue[g]) and n > 0 and epoch > 0:
                        pass                          # fewer operations outstanding than the count leaves in flight: nothing to wait for (legal)
                    done, queue[g] = (queue[g][:-n], queue[g][-n:]) if n else (queue[g], [])
                    for key, ver, last in done:
                        if key is not None and last:
                            landed[(key[0], key[1], g)] = (ver, epoch)
                elif e[0] == 'read':
                    _, slot, buf, tile, kt = e
                    key, ver = (buf, slot), (tile, kt)
                    for gg in (0, 1):
                        iv = issued.get(key, {}).get(gg)
                        lv = landed.get((buf, slot, gg))
                        if iv is None or iv[0] != ver:
                            errors.append(f'RAW: group {g} reads {key} expecting {ver} in epoch {epoch}, group {gg} last issued {iv}')
                        elif lv is None or lv[0] != ver or lv[1] >= epoch:
                            errors.append(f'RAW: group {g} reads {key} = {ver} in epoch {epoch}, group {gg}\'s pieces retired {lv}')
                    reads.setdefault(key, []).append((ver, epoch, g))
                elif e[0] == 'lds_reuse':
                    issued.clear()
                    landed.clear()
    for g in (0, 1):
        assert not queue[g], 'operations left in flight at kernel end'
    return errors


@pytest.mark.parametrize('ntiles,nk,bn,epi', [c for c in itertools.product((1, 2, 3, 9), (4, 6, 12, 16, 48), (256, 192), ('plain16', 'resid_reg', 'resid_lds'))
                                             if not (c[2] == 192 and c[3] == 'resid_reg')])
def test_operand_ring_has_no_hazard(ntiles, nk, bn, epi):
    errors = check(ntiles, nk, bn, epi)
    assert not errors, errors[:5]


@pytest.mark.parametrize('ntiles,nk', list(itertools.product((1, 2, 3, 9), (4, 6, 12, 16, 48))))
def test_operand_ring_of_the_192_row_tile_has_no_hazard(ntiles, nk):
    """192 x 256 tiles: the two wave groups issue different piece counts per section (X halves of 12 pieces) -- the steady-state count is common, the
    one-LB-group waits are per group"""
    errors = check(ntiles, nk, 256, 'resid_reg', bm=192)
    assert not errors, errors[:5]


@pytest.mark.parametrize('mut', ['nkeep+1', 'inflight+1'])
def test_the_checker_bites_on_the_192_row_tile(mut):
    assert check(3, 12, 256, 'resid_reg', mut, bm=192), f'mutation {mut} went unnoticed'


@pytest.mark.parametrize('ntiles,nk,bn,epi', [c for c in itertools.product((1, 2, 3, 5), (4, 6, 8, 10, 24, 40), (256, 192), ('plain16', 'resid_reg', 'resid_lds'))
                                             if (c[2] == 192) == (c[3] == 'resid_lds')])
def test_operand_ring_of_the_mxfp8_kernel_has_no_hazard(ntiles, nk, bn, epi):
    """gemm8f.hip: K-tiles of 128 (K = 512 .. 5120), an X slot = two DMA pieces + its scale dword in the same counted stream; 256-wide tiles store from registers
    (qkv, fc1 -> plain16; fc2 at N = 1024 / 1280 -> resid_reg), 192-wide tiles take the LDS-staged residual epilogue"""
    errors = check(ntiles, nk, bn, epi, fp8=True)
    assert not errors, errors[:5]


@pytest.mark.parametrize('mut', ['nkeep+1', 'inflight+1'])
def test_the_checker_bites_on_the_mxfp8_kernel(mut):
    """a count that leaves one operation too many in flight would let an MFMA read a fragment -- or a scale dword -- that has not landed"""
    assert check(3, 6, 256, 'plain16', mut, fp8=True), f'mutation {mut} went unnoticed'


@pytest.mark.parametrize('mut', ['nkeep+1', 'inflight+1', 'early_x1'])
def test_the_checker_bites(mut):
    """counted waits that leave one piece too many in flight (inside a tile, at its boundary), and an X1 restage issued one segment early, must be flagged"""
    assert check(3, 12, 256, 'plain16', mut), f'mutation {mut} went unnoticed'
