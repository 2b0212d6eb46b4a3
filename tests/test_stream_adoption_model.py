"""CPU model of the cross-call stream ordering of a handle (easy_vitpose_amd/csrc/vitpose_api.hip: adopt_stream, check_ready, vp_infer_device_stream, vp_synchronize;
round 5).  At <= 16 crops the stream-ordered entry enqueues its chunk on the CALLER's stream, so the handle's workspaces are used from several streams over time and the
library alone must order consecutive calls.  The model replays random sequences of calls -- own-stream entries (vp_infer, vp_infer_device, ...), the stream-ordered entry
on the legacy default stream / on two caller streams at small and at large batches, vp_synchronize -- as operations on in-order streams with event records and waits, and
checks by reachability that

  * the work of call k happens-before the work of call k + 1, whatever streams the two ran on (WAR / RAW on the shared workspaces),
  * a small stream-ordered call runs ON the caller's stream (in order with its producers and consumers by construction), a large one is fenced on both sides,
  * everything enqueued so far happens-before the return of vp_synchronize,
  * a caller's stream is never touched by a LATER call on another stream (the caller may have destroyed it).

A transcription of the control flow, not the code itself; mutations at the end check that the checker bites.  The GPU side of the same logic:
tests/test_gpu_api.py::test_small_batch_ordered_entry_runs_on_the_callers_stream."""
import random

import pytest

OWN = 'own'


class Handle:
    def __init__(self, mut=None):
        self.mut = mut
        self.streams = {}               # name -> list of ops
        self.foreign_pending = False
        self.last_stream_id = None
        self.ev = {}                    # event name -> (stream, index) of its latest record
        self.calls = []                 # (call id, stream, index of its work op)
        self.touched_after = []         # (stream, call id) a call touched although it ran elsewhere

    def op(self, stream, kind, arg=None):
        self.streams.setdefault(stream, []).append((kind, arg))
        return len(self.streams[stream]) - 1

    def record(self, name, stream):
        self.ev[name] = (stream, self.op(stream, 'record', name))

    def wait(self, stream, name):
        self.op(stream, 'wait', self.ev[name])          # a wait captures the event's latest record at the time of the call

    # --- vitpose_api.hip adopt_stream
    def adopt(self, s, call):
        if not self.foreign_pending and s == OWN:
            return
        if self.foreign_pending and s == self.last_stream_id and s != OWN:
            return
        if not self.foreign_pending:
            if self.mut != 'no-record':
                self.record('ev_sw', OWN)
        if self.mut == 'no-wait-between-callers' and self.foreign_pending and s != OWN:
            pass
        elif 'ev_sw' in self.ev:
            self.wait(s, 'ev_sw')
        if s == OWN:
            self.foreign_pending = False

    def work(self, stream, call):
        self.calls.append((call, stream, self.op(stream, 'work', call)))

    # --- entries
    def own_entry(self, call):                          # vp_infer / vp_infer_device / submit / frame / ...: check_ready adopts the own stream
        if self.mut != 'own-entry-no-adopt':
            self.adopt(OWN, call)
        self.work(OWN, call)

    def ordered_entry(self, call, cs, n, max_n=16):     # vp_infer_device_stream
        if 0 < n <= max_n:
            self.adopt(cs, call)
            self.work(cs, call)
            self.record('ev_sw', cs)
            self.foreign_pending = True
            self.last_stream_id = cs
        else:                                           # the event fence of rounds 2-4
            self.adopt(OWN, call)
            self.record('ev_in', cs)
            self.wait(OWN, 'ev_in')
            self.work(OWN, call)
            self.record('ev_out', OWN)
            self.wait(cs, 'ev_out')

    def synchronize(self):                              # vp_synchronize: the host waits for ev_sw (if a caller's stream ran last) and for the own stream
        pts = []
        if self.foreign_pending and 'ev_sw' in self.ev and self.mut != 'sync-own-only':
            pts.append(self.ev['ev_sw'])
        pts.append((OWN, len(self.streams.get(OWN, [])) - 1))
        return pts


def happens_before(h, a, b):
    """Is op a = (stream, index) ordered before op b?  Backwards search from b over stream order and wait -> record edges."""
    seen, todo = set(), [b]
    while todo:
        s, i = todo.pop()
        if (s, i) in seen or i < 0:
            continue
        seen.add((s, i))
        if s == a[0] and i >= a[1]:
            return True
        for j in range(i, -1, -1):                      # everything earlier on the same stream, following the waits met on the way
            kind, arg = h.streams[s][j]
            if kind == 'wait' and arg not in seen:
                todo.append(arg)
            if s == a[0] and j == a[1]:
                return True
        # (the loop above already walked the whole prefix of s)
    return False


def run(seq, mut=None):
    h = Handle(mut)
    bad = []
    for k, (kind, cs, n) in enumerate(seq):
        before = {s: len(ops) for s, ops in h.streams.items()}
        if kind == 'own':
            h.own_entry(k)
        elif kind == 'ordered':
            h.ordered_entry(k, cs, n)
        else:
            done = h.synchronize()
            for call, s, i in h.calls:
                if not any(happens_before(h, (s, i), p) or (s, i) == p for p in done):
                    bad.append(f'vp_synchronize returns before the work of call {call} on {s}')
            continue
        # a call may only touch its own target streams: own, and the caller's stream it was given
        for s, ops in h.streams.items():
            if len(ops) > before.get(s, 0) and s not in (OWN, cs if kind == 'ordered' else OWN):
                bad.append(f'call {k} touched stream {s} it was not given')
        if kind == 'ordered' and 0 < n <= 16 and h.calls[-1][1] != cs:
            bad.append(f'small ordered call {k} did not run on the caller\'s stream')
        if len(h.calls) >= 2:
            (c0, s0, i0), (c1, s1, i1) = h.calls[-2], h.calls[-1]
            if not happens_before(h, (s0, i0), (s1, i1)):
                bad.append(f'work of call {c1} on {s1} is not ordered behind call {c0} on {s0}')
        if kind == 'ordered' and n > 16:                 # fenced on both sides: producers before, consumers after
            c1, s1, i1 = h.calls[-1]
            prod = (cs, before.get(cs, 0) - 1)
            if prod[1] >= 0 and not happens_before(h, prod, (s1, i1)):
                bad.append(f'large ordered call {k}: the caller\'s earlier work is not ordered before the library\'s')
            if not happens_before(h, (s1, i1), (cs, len(h.streams[cs]) - 1)):
                bad.append(f'large ordered call {k}: the caller\'s later work is not ordered behind the library\'s')
    return bad


def sequences(seed, count, length):
    rng = random.Random(seed)
    kinds = [('own', None, 0), ('ordered', 'null', 4), ('ordered', 'A', 8), ('ordered', 'B', 1), ('ordered', 'A', 64), ('ordered', 'null', 256), ('sync', None, 0)]
    for _ in range(count):
        yield [rng.choice(kinds) for _ in range(length)]


def test_every_call_is_ordered_behind_the_previous_one():
    for seq in sequences(1, 400, 14):
        bad = run(seq)
        assert not bad, (seq, bad[:3])
    # the alternation of the GPU test: default stream, side stream, own stream, host entry
    seq = [('own', None, 0)] + [('ordered', 'null', 4), ('ordered', 'A', 4), ('own', None, 0), ('own', None, 0)] * 4 + [('ordered', 'A', 4), ('sync', None, 0)]
    assert not run(seq)


@pytest.mark.parametrize('mut', ['no-record', 'no-wait-between-callers', 'own-entry-no-adopt', 'sync-own-only'])
def test_checker_flags_broken_orderings(mut):
    """No event recorded on the own stream before a caller's stream takes over; no wait when one caller's stream follows another; an own-stream entry that does not
    wait for a caller's stream; a vp_synchronize that only waits for the own stream: each must be flagged."""
    assert any(run(seq, mut) for seq in sequences(2, 300, 12)), mut
