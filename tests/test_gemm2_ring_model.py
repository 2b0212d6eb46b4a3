"""CPU model of the 2-phase kernel's operand ring (easy_vitpose_amd/csrc/gemm.hip, the generic K-loop: one k-block per barrier, and -- round 5, PIPE 6 --
TWO k-blocks per barrier): the order of LDS-DMA issues, counted `vmcnt` waits, barriers and fragment reads of a workgroup is replayed for every ring depth
and K the product can run, and the hazards the loop's comments argue about are checked mechanically:

  RAW  a read of ring slot s must find exactly the k-block it expects, with every wave's pieces of it retired by the counted wait in front of the barrier
       that precedes the read (every wave runs the same program, so "this wave's pieces are retired before the barrier" covers all of them);
  WAR  a slot may only be restaged after a barrier that every wave passes AFTER its reads of the slot's previous content;
  the counted waits never count on a k-block that was not issued (a too-large count waits for nothing), and nothing is left in flight at the loop's end.

A transcription of the control flow (gemm.hip "PIPE 6 ... KS" block), not the kernel itself -- the race screen of the real code is the bit identity of its
results across tile configurations on the GPU (tests/test_gpu_gemm_cfgs.py, test_small_batch_tile_rule_is_bit_identical).  Mutations at the end check that
the checker bites."""
import pytest


def program(stages, ks, nk, mut=None):
    """Event list of one wave: ('issue', kblock, slot), ('wait', allowed_in_flight), ('bar',), ('read', kblock, slot)."""
    ev = []
    pro = stages - ks                                        # gemm.hip: for (s = 0; s < STAGES - KS; ++s) if (s < nk) stage(s, s)
    for s in range(pro):
        if s < nk:
            ev.append(('issue', s, s))
    buf, pbuf = 0, stages - ks
    for kt in range(0, nk, ks):
        allowed = stages - 2 * ks
        if mut == 'allowed+1':
            allowed += 1
        cond = kt + stages - ks <= nk
        if mut == 'late-tail':                               # keeps the steady-state count one iteration too long
            cond = kt + stages - ks <= nk + ks
        if stages > 2 * ks and cond:
            ev.append(('wait', allowed))
        else:
            ev.append(('wait', 0))
        ev.append(('bar',))
        for j in range(ks):
            pb = (pbuf + j) % stages
            if mut == 'slot+1':
                pb = (pb + 1) % stages
            nxt = kt + stages - ks + j
            if mut == 'issue-early':
                nxt += 0
            if nxt < nk:
                ev.append(('issue', nxt, pb))
        for j in range(ks):
            ev.append(('read', kt + j, (buf + j) % stages))
        buf = (buf + ks) % stages
        pbuf = (pbuf + ks) % stages
    ev.append(('wait', 0))                                   # epilogue: s_waitcnt vmcnt(0) + __syncthreads()
    ev.append(('bar',))
    return ev


def check(ev, nk):
    """Replays one wave's program (all waves run it: barriers line them up).  Returns a list of violations."""
    bad = []
    queue = []                  # issued, not yet retired k-blocks of this wave, in order (the in-order vmcnt counter)
    retired_at = {}             # k-block -> barrier index after which it is visible to every wave (wait before barrier b => visible after b)
    pending_visible = set()     # retired by a wait, waiting for the next barrier
    slot_holds = {}             # slot -> k-block last issued into it
    last_read_bar = {}          # slot -> barrier count at the time of the last read of the slot's content
    nbar = 0
    reads = []
    for e in ev:
        if e[0] == 'issue':
            _, kb, slot = e
            if slot in last_read_bar and last_read_bar[slot] >= nbar and slot in slot_holds:
                bad.append(f'WAR: k-block {kb} restages slot {slot} in the barrier interval of a read of k-block {slot_holds[slot]}')
            if slot in slot_holds and slot_holds[slot] not in [r for r in reads] and slot_holds[slot] < nk:
                bad.append(f'overwrite: k-block {kb} restages slot {slot} before k-block {slot_holds[slot]} was read')
            slot_holds[slot] = kb
            queue.append(kb)
        elif e[0] == 'wait':
            allowed = e[1]
            if allowed > len(queue) and allowed > 0:
                bad.append(f'wait vmcnt({allowed} k-blocks) with only {len(queue)} in flight: the count waits for nothing')
            while len(queue) > allowed:
                pending_visible.add(queue.pop(0))
        elif e[0] == 'bar':
            nbar += 1
            for kb in pending_visible:
                retired_at[kb] = nbar
            pending_visible = set()
        else:
            _, kb, slot = e
            if slot_holds.get(slot) != kb:
                bad.append(f'RAW: read of k-block {kb} finds k-block {slot_holds.get(slot)} in slot {slot}')
            if kb not in retired_at or retired_at[kb] > nbar:
                bad.append(f'RAW: k-block {kb} read before its wait + barrier')
            last_read_bar[slot] = nbar
            reads.append(kb)
    if queue:
        bad.append(f'{len(queue)} k-blocks in flight at the end')
    if reads != list(range(nk)):
        bad.append(f'k order: {reads[:8]}...')
    return bad


# (STAGES, KS) of every configuration the product library instantiates (gemm.hip Cfg1/3/8/9/11: 2, Cfg15: 3, Cfg12: 4 with one k-block per barrier; Cfg30 / Cfg31: 6 with two)
# and of the measurement build's candidates (3-8 stages; PIPE 6 with 4, 5, 6, 8)
PRODUCT = [(2, 1), (3, 1), (4, 1), (6, 2)]
TOOLS = [(5, 1), (6, 1), (8, 1), (4, 2), (5, 2), (8, 2)]


@pytest.mark.parametrize('stages,ks', PRODUCT + TOOLS)
def test_ring_schedule_has_no_hazard(stages, ks):
    """Every K the path runs (k-blocks of 64: K = 256 ... 5120, deconv K = 4 x 256 ... 4 x 1280) and every short K down to one iteration."""
    for nk in range(ks, 82, ks):
        bad = check(program(stages, ks, nk), nk)
        assert not bad, (stages, ks, nk, bad[:3])


@pytest.mark.parametrize('mut', ['allowed+1', 'slot+1', 'late-tail'])
def test_checker_flags_broken_schedules(mut):
    """A wait that leaves one k-block too many in flight, a restage into the slot being read, a steady-state count kept into the tail: each must be flagged for
    some (STAGES, KS, K) -- the checker bites."""
    flagged = 0
    for stages, ks in PRODUCT + TOOLS:
        for nk in range(ks, 40, ks):
            if check(program(stages, ks, nk, mut), nk):
                flagged += 1
    assert flagged > 0, mut
    if mut != 'late-tail':                                   # the first two break EVERY depth with a steady state (the tail mutation needs nk > stages to matter)
        for stages, ks in [(4, 1), (6, 2)]:
            assert check(program(stages, ks, 16, mut), 16), (mut, stages, ks)
