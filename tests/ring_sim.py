"""Test backends for the C++ ring schedules (include/fa2_ring_mi355x.h: fa2_ring_backend).

libfa2_ring_mi355x.so does everything device-related through a table of callbacks.  The tables here let the
SAME schedule code that drives 8 GPUs over RCCL run without one:

  * SimWorld   -- a discrete-event simulator of P ranks.  Streams are FIFO queues, events carry HIP's
                  "wait for the most recent record" meaning, grouped send/recv are matched per (source,
                  destination) pair in issue order and copy bytes only once BOTH sides have started.  Nothing
                  runs while a rank's call enqueues; afterwards the operations of all ranks are executed in an
                  order chosen by a policy (seeded random, comm-eager, compute-eager, ...) among those the
                  fences allow.  A missing or mis-indexed fence or slot therefore shows up as a wrong result
                  under some order, a wrong peer or an unmatched exchange as a reported deadlock.
  * EagerWorld -- executes every operation at enqueue time and moves bytes with torch.distributed (gloo):
                  world_size-2 runs of the C++ schedule across real processes.

Compute callbacks are numpy restatements built on the oracle (oracle.ring_step, oracle.attention_forward):
the step kernel's state layout (fa2_mi355x.h: fa2_forward_step) on host memory, bf16 stored as uint16.
"""
import collections
import ctypes
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FA2_OK = 0
FA2_DTYPE_BF16, FA2_DTYPE_F32 = 0, 1
RELAY, MESH = 0, 1

_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_pvp = ctypes.POINTER(ctypes.c_void_p)

CB = {
    "stream_create": ctypes.CFUNCTYPE(_i, _vp, _pvp),
    "stream_destroy": ctypes.CFUNCTYPE(_i, _vp, _vp),
    "event_create": ctypes.CFUNCTYPE(_i, _vp, _pvp),
    "event_destroy": ctypes.CFUNCTYPE(_i, _vp, _vp),
    "event_record": ctypes.CFUNCTYPE(_i, _vp, _vp, _vp),
    "stream_wait_event": ctypes.CFUNCTYPE(_i, _vp, _vp, _vp),
    "group_start": ctypes.CFUNCTYPE(_i, _vp),
    "send": ctypes.CFUNCTYPE(_i, _vp, _vp, _sz, _i, _vp),
    "recv": ctypes.CFUNCTYPE(_i, _vp, _vp, _sz, _i, _vp),
    "group_end": ctypes.CFUNCTYPE(_i, _vp),
    "forward_step": ctypes.CFUNCTYPE(_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i,
                                     _i, _i, _vp),
    "state_finalize": ctypes.CFUNCTYPE(_i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _vp),
    "backward_block": ctypes.CFUNCTYPE(_i, _vp, *([_vp] * 9), _i, _i, _i, _i, _i, _f, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp,
                                       _i),
    "accumulate_bf16_2d": ctypes.CFUNCTYPE(_i, _vp, _vp, _vp, _sz, _sz, _sz, _i, _vp),
    "convert_f32_to_bf16": ctypes.CFUNCTYPE(_i, _vp, _vp, _vp, _sz, _vp),
}
ORDER = ["stream_create", "stream_destroy", "event_create", "event_destroy", "event_record", "stream_wait_event",
         "group_start", "send", "recv", "group_end", "forward_step", "state_finalize", "backward_block",
         "accumulate_bf16_2d", "convert_f32_to_bf16"]


class Backend(ctypes.Structure):
    """fa2_ring_backend, field for field."""
    _fields_ = [("user", _vp)] + [(n, CB[n]) for n in ORDER]


# ------------------------------------------------------------------------------------------------ memory views
def view(ptr, count, dtype):
    """A numpy array over `count` elements of host memory at address ptr (no copy)."""
    dt = np.dtype(dtype)
    buf = (ctypes.c_char * (int(count) * dt.itemsize)).from_address(int(ptr))
    return np.frombuffer(buf, dtype=dt, count=int(count))


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16(x):
    """Round to nearest even, like v_cvt_pk_bf16_f32 (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def round_bf16(x):
    return bf16_to_f32(f32_to_bf16(x))


def _heads(ptr, BH, n, hs, d, dtype):
    """[BH] views [n, d] of a tensor whose heads are hs rows apart, ptr at the first row of head 0."""
    hs = hs or n
    flat = view(ptr, ((BH - 1) * hs + n) * d, dtype)
    return [flat[h * hs * d:(h * hs + n) * d].reshape(n, d) for h in range(BH)]


def _rows(ptr, BH, n, hs):
    hs = hs or n
    flat = view(ptr, (BH - 1) * hs + n, np.float32)
    return [flat[h * hs:h * hs + n] for h in range(BH)]


# ------------------------------------------------------------------------------------------------ compute (numpy + oracle)
class NumpyCompute:
    """The compute entries of the table on host memory.  Each returns a closure to run later."""

    @staticmethod
    def forward_step(Q, K, V, O, L, Oacc, M, B, H, nq, nk, d, scale, dtype, first, last, q_hs, k_hs, causal, shift):
        import oracle
        BH = B * H

        def run():
            if dtype == FA2_DTYPE_F32:          # state: O un-normalised, L = l, M (ring_attention_kernel.cu:125-137)
                Qs, Ks, Vs, Os = (_heads(p, BH, n, 0, d, np.float32) for p, n in ((Q, nq), (K, nk), (V, nk), (O, nq)))
                Ls = _rows(L, BH, nq, 0)
                Ms = _rows(M, BH, nq, 0) if M else [None] * BH
                for h in range(BH):
                    m = Ms[h] if Ms[h] is not None else np.empty(nq, np.float32)
                    if first:
                        Os[h][:] = 0
                        Ls[h][:] = 0
                        m[:] = -np.inf
                    oracle.ring_step(np.ascontiguousarray(Qs[h]), np.ascontiguousarray(Ks[h]), np.ascontiguousarray(Vs[h]),
                                     Os[h], Ls[h], m, float(scale), bool(last))
                return
            Qs = _heads(Q, BH, nq, q_hs, d, np.uint16)
            Ks = _heads(K, BH, nk, k_hs, d, np.uint16)
            Vs = _heads(V, BH, nk, k_hs, d, np.uint16)
            As = _heads(Oacc, BH, nq, q_hs, d, np.float32)
            Ls, Ms = _rows(L, BH, nq, q_hs), _rows(M, BH, nq, q_hs)
            Os = _heads(O, BH, nq, q_hs, d, np.uint16) if O else None
            for h in range(BH):
                q, k, v = (np.ascontiguousarray(bf16_to_f32(a)) for a in (Qs[h], Ks[h], Vs[h]))
                if causal:                      # the schedules use it for the local block only: starts the state
                    assert first and shift == 0 and nq == nk
                    o, lse = oracle.attention_forward(q, k, v, float(scale), causal=True)
                    As[h][:] = o
                    Ms[h][:] = lse
                    Ls[h][:] = 1.0
                    continue
                acc = np.ascontiguousarray(As[h])
                l = np.ascontiguousarray(Ls[h])
                m = np.ascontiguousarray(Ms[h])
                if first:
                    acc[:] = 0
                    l[:] = 0
                    m[:] = -np.inf
                oracle.ring_step(q, k, v, acc, l, m, float(scale), False)
                As[h][:], Ls[h][:], Ms[h][:] = acc, l, m
            if last:
                _fin_heads(Os, Ls, As, Ms)
        return run

    @staticmethod
    def state_finalize(O, L, Oacc, M, rows, d, dtype):
        def run():
            o = view(O, rows * d, np.uint16).reshape(rows, d)
            acc = view(Oacc, rows * d, np.float32).reshape(rows, d)
            l, m = view(L, rows, np.float32), view(M, rows, np.float32)
            o[:] = f32_to_bf16(acc / l[:, None])
            l[:] = m + np.log(l)
        return run

    @staticmethod
    def backward_block(Q, K, V, O, L, dO, dQ, dK, dV, B, H, nq, nk, d, scale, dtype, q_hs, k_hs, q_row0, causal, shift, ws,
                       ws_bytes, phases):
        BH = B * H
        hs = q_hs or nq

        def run():
            assert dtype == FA2_DTYPE_BF16
            plane = view(ws, BH * hs, np.float32)           # D, dense [BH][hs]
            Qs, Gs, Os = (_heads(p, BH, nq, q_hs, d, np.uint16) for p in (Q, dO, O))
            Ks, Vs = (_heads(p, BH, nk, k_hs, d, np.uint16) for p in (K, V))
            Ls = _rows(L, BH, nq, q_hs)
            if phases & 6:
                dQs = _heads(dQ, BH, nq, q_hs, d, np.uint16)
                dKs, dVs = (_heads(p, BH, nk, k_hs, d, np.uint16) for p in (dK, dV))
            for h in range(BH):
                Dh = plane[h * hs + q_row0:h * hs + q_row0 + nq]
                g = bf16_to_f32(Gs[h]).astype(np.float64)
                if phases & 1:
                    Dh[:] = (g * bf16_to_f32(Os[h])).sum(1)
                if not phases & 6:
                    continue
                q, k, v = (bf16_to_f32(a).astype(np.float64) for a in (Qs[h], Ks[h], Vs[h]))
                P = np.exp(scale * (q @ k.T) - Ls[h].astype(np.float64)[:, None])
                if causal:
                    P[np.arange(nk)[None, :] > np.arange(nq)[:, None] + shift] = 0.0
                dS = P * (g @ v.T - Dh.astype(np.float64)[:, None])
                dQs[h][:] = f32_to_bf16(scale * (dS @ k))
                dKs[h][:] = f32_to_bf16(scale * (dS.T @ q))
                dVs[h][:] = f32_to_bf16(P.T @ g)
        return run

    @staticmethod
    def accumulate_bf16_2d(acc, src, rows, cols, pitch, init):
        def run():
            span = (rows - 1) * pitch + cols
            a, s = view(acc, span, np.float32), view(src, span, np.uint16)
            for r in range(rows):
                x = bf16_to_f32(s[r * pitch:r * pitch + cols])
                if init:
                    a[r * pitch:r * pitch + cols] = x
                else:
                    a[r * pitch:r * pitch + cols] += x
        return run

    @staticmethod
    def convert_f32_to_bf16(src, dst, n):
        def run():
            view(dst, n, np.uint16)[:] = f32_to_bf16(view(src, n, np.float32))
        return run


def _split_stream(name, a):
    """(arguments without the stream, stream): the stream is the last argument, except before `phases`."""
    if name == "backward_block":
        return a[:-2] + a[-1:], a[-2]
    return a[:-1], a[-1]


def _fin_heads(Os, Ls, As, Ms):
    for o, l, a, m in zip(Os, Ls, As, Ms):
        o[:] = f32_to_bf16(a / l[:, None])
        l[:] = m + np.log(l)


# ------------------------------------------------------------------------------------------------ the simulator
class Deadlock(AssertionError):
    pass


class _Group:
    def __init__(self, rank):
        self.rank, self.sends, self.recvs, self.started, self.left = rank, [], [], False, 0


class SimWorld:
    """P simulated ranks.  Usage: ctxs = [world.ctx(r) for r in range(P)]; call the C entry points for every rank
    (each call only enqueues), then world.run()."""

    def __init__(self, P, seed=0, policy="random"):
        self.P, self.policy, self.rng = P, policy, random.Random(seed)
        self.streams = {}          # id -> dict(rank, ops deque, kind)
        self.events = {}           # id -> last record token (None: never recorded)
        self.done_tokens = set()
        self.next_id = 16
        self.next_token = 1
        self.open_group = {}       # rank -> _Group being built
        self.in_group = {}         # rank -> bool
        self.seq = collections.Counter()      # (src, dst, kind) -> next sequence number
        self.posted = {}           # (src, dst, seq) -> {"send": (buf, bytes, group), "recv": (...)}
        self.errors = []
        self.trace = []
        self._keep = []
        self.lib = ring_lib()

    # -- handles
    def new_stream(self, rank, kind="compute"):
        sid = self.next_id
        self.next_id += 1
        self.streams[sid] = {"rank": rank, "ops": collections.deque(), "kind": kind}
        return sid

    def _new_event(self):
        eid = self.next_id
        self.next_id += 1
        self.events[eid] = None
        return eid

    # -- the table for one rank
    def backend(self, rank):
        w = self

        def guard(fn):
            def wrapped(*a):
                try:
                    r = fn(*a)
                    return FA2_OK if r is None else r
                except Exception as e:      # noqa: BLE001 -- report through the status, keep the traceback
                    import traceback
                    w.errors.append(traceback.format_exc())
                    return -6
            return wrapped

        def stream_create(user, out):
            out[0] = w.new_stream(rank, "comm")

        def stream_destroy(user, s):
            assert not w.streams[s]["ops"], "stream destroyed with work pending"
            del w.streams[s]

        def event_create(user, out):
            out[0] = w._new_event()

        def event_destroy(user, e):
            del w.events[e]

        def event_record(user, e, s):
            tok = w.next_token
            w.next_token += 1
            w.events[e] = tok
            w.streams[s]["ops"].append(("record", tok))

        def stream_wait_event(user, s, e):
            w.streams[s]["ops"].append(("wait", w.events[e]))       # None: never recorded -> no-op (HIP semantics)

        def group_start(user):
            assert not w.in_group.get(rank), "nested group"
            w.in_group[rank] = True
            w.open_group[rank] = {}

        def _post(kind, buf, nbytes, peer, s):
            assert 0 <= peer < w.P and peer != rank, f"rank {rank}: bad peer {peer}"
            key = (rank, peer, kind)
            n = w.seq[key]
            w.seq[key] += 1
            grp = w.open_group[rank].setdefault(s, _Group(rank)) if w.in_group.get(rank) else _Group(rank)
            (grp.sends if kind == "send" else grp.recvs).append((buf, nbytes, peer, n))
            grp.left += 1
            if not w.in_group.get(rank):
                w.streams[s]["ops"].append(("group", grp))

        def send(user, buf, nbytes, peer, s):
            _post("send", buf, nbytes, peer, s)

        def recv(user, buf, nbytes, peer, s):
            _post("recv", buf, nbytes, peer, s)

        def group_end(user):
            assert w.in_group.get(rank), "group_end without group_start"
            w.in_group[rank] = False
            for s, grp in w.open_group.pop(rank).items():
                w.streams[s]["ops"].append(("group", grp))

        def compute(name):
            def cb(user, *a):
                args, s = _split_stream(name, a)
                w.streams[s]["ops"].append(("compute", getattr(NumpyCompute, name)(*args), name))
            return cb

        fns = dict(stream_create=stream_create, stream_destroy=stream_destroy, event_create=event_create,
                   event_destroy=event_destroy, event_record=event_record, stream_wait_event=stream_wait_event,
                   group_start=group_start, send=send, recv=recv, group_end=group_end)
        for n in ("forward_step", "state_finalize", "backward_block", "accumulate_bf16_2d", "convert_f32_to_bf16"):
            fns[n] = compute(n)
        be = Backend()
        be.user = None
        for n in ORDER:
            c = CB[n](guard(fns[n]))
            self._keep.append(c)
            setattr(be, n, c)
        self._keep.append(be)
        return be

    def ctx(self, rank):
        h = _vp()
        st = self.lib.fa2_ring_ctx_create_with_backend(ctypes.byref(h), ctypes.byref(self.backend(rank)), rank, self.P)
        assert st == 0 and not self.errors, (st, self.errors)
        return h

    # -- execution
    def _transfer_ready(self):
        out = []
        for key, ent in self.posted.items():
            if "send" in ent and "recv" in ent:
                out.append(key)
        return out

    def _do_transfer(self, key):
        ent = self.posted.pop(key)
        (sbuf, sb, sg), (rbuf, rb, rg) = ent["send"], ent["recv"]
        assert sb == rb, f"size mismatch on pair {key}: send {sb} recv {rb}"
        ctypes.memmove(rbuf, sbuf, sb)
        sg.left -= 1
        rg.left -= 1
        self.trace.append(("xfer",) + key)

    def run(self, max_actions=1_000_000):
        """Executes everything that was enqueued; raises Deadlock if the fences / matching never let it finish."""
        assert not self.errors, self.errors
        for _ in range(max_actions):
            acts = []
            for sid, st in self.streams.items():
                if not st["ops"]:
                    continue
                op = st["ops"][0]
                if op[0] == "wait":
                    if op[1] is None or op[1] in self.done_tokens:
                        acts.append(("pop", sid))
                elif op[0] == "group":
                    g = op[1]
                    if not g.started:
                        acts.append(("start", sid))
                    elif g.left == 0:
                        acts.append(("pop", sid))
                else:
                    acts.append(("exec", sid))
            for key in self._transfer_ready():
                acts.append(("xfer", key))
            if not acts:
                pending = {sid: list(st["ops"])[:3] for sid, st in self.streams.items() if st["ops"]}
                if pending:
                    raise Deadlock(f"no runnable operation; pending heads: {pending}; half-posted: {list(self.posted)[:8]}")
                return
            kind, x = self._choose(acts)
            if kind == "xfer":
                self._do_transfer(x)
                continue
            st = self.streams[x]
            op = st["ops"][0]
            if kind == "start":
                g = op[1]
                g.started = True
                for buf, nb, peer, n in g.sends:
                    self.posted.setdefault((g.rank, peer, n), {})["send"] = (buf, nb, g)
                for buf, nb, peer, n in g.recvs:
                    self.posted.setdefault((peer, g.rank, n), {})["recv"] = (buf, nb, g)
                continue
            st["ops"].popleft()
            if op[0] == "record":
                self.done_tokens.add(op[1])
            elif op[0] == "compute":
                op[1]()
                self.trace.append((op[2], st["rank"]))
        raise AssertionError("simulation did not finish")

    def _choose(self, acts):
        pol = self.policy
        if pol == "random":
            return self.rng.choice(acts)

        def is_comm(a):
            return a[0] == "xfer" or self.streams[a[1]]["kind"] == "comm"
        if pol in ("comm_first", "compute_first"):
            want = pol == "comm_first"
            pref = [a for a in acts if is_comm(a) == want]
            return self.rng.choice(pref or acts)
        if pol == "lazy_transfer":          # bytes move as late as the fences allow
            pref = [a for a in acts if a[0] != "xfer"]
            return self.rng.choice(pref or acts)
        if pol in ("low_rank_first", "high_rank_first"):
            def rk(a):
                return a[1][1] if a[0] == "xfer" else self.streams[a[1]]["rank"]
            best = (min if pol == "low_rank_first" else max)(rk(a) for a in acts)
            return self.rng.choice([a for a in acts if rk(a) == best])
        raise ValueError(pol)


POLICIES = ["random", "comm_first", "compute_first", "lazy_transfer", "low_rank_first", "high_rank_first"]


# ------------------------------------------------------------------------------------------------ eager backend over gloo
class EagerWorld:
    """One rank of a real multi-process job: operations execute at enqueue time (a legal order: every wait in the
    schedules refers to an event recorded earlier in host order -- asserted), bytes move by torch.distributed."""

    def __init__(self, dist, rank, nranks):
        import torch
        self.dist, self.rank, self.P, self.torch = dist, rank, nranks, torch
        self.events, self.next_id, self.group, self.errors, self._keep = {}, 16, None, [], []
        self.lib = ring_lib()

    def backend(self):
        w, torch, dist = self, self.torch, self.dist

        def guard(fn):
            def wrapped(*a):
                try:
                    r = fn(*a)
                    return FA2_OK if r is None else r
                except Exception:      # noqa: BLE001
                    import traceback
                    w.errors.append(traceback.format_exc())
                    return -6
            return wrapped

        def new_id(user, out):
            out[0] = w.next_id
            w.next_id += 1

        def event_create(user, out):
            new_id(user, out)
            w.events[out[0]] = False

        def event_record(user, e, s):
            w.events[e] = True

        def stream_wait_event(user, s, e):
            pass        # eager: everything enqueued earlier has already run

        def group_start(user):
            w.group = []

        def _t(buf, nbytes):
            return torch.from_numpy(view(buf, nbytes, np.uint8))

        def send(user, buf, nbytes, peer, s):
            w.group.append(dist.P2POp(dist.isend, _t(buf, nbytes), peer))

        def recv(user, buf, nbytes, peer, s):
            w.group.append(dist.P2POp(dist.irecv, _t(buf, nbytes), peer))

        def group_end(user):
            for r in dist.batch_isend_irecv(w.group):
                r.wait()
            w.group = None

        def compute(name):
            def cb(user, *a):
                args, _ = _split_stream(name, a)
                getattr(NumpyCompute, name)(*args)()
            return cb

        fns = dict(stream_create=new_id, stream_destroy=lambda u, s: None, event_create=event_create,
                   event_destroy=lambda u, e: None, event_record=event_record, stream_wait_event=stream_wait_event,
                   group_start=group_start, send=send, recv=recv, group_end=group_end)
        for n in ("forward_step", "state_finalize", "backward_block", "accumulate_bf16_2d", "convert_f32_to_bf16"):
            fns[n] = compute(n)
        be = Backend()
        be.user = None
        for n in ORDER:
            c = CB[n](guard(fns[n]))
            self._keep.append(c)
            setattr(be, n, c)
        self._keep.append(be)
        return be

    def ctx(self):
        h = _vp()
        st = self.lib.fa2_ring_ctx_create_with_backend(ctypes.byref(h), ctypes.byref(self.backend()), self.rank, self.P)
        assert st == 0 and not self.errors, (st, self.errors)
        return h


# ------------------------------------------------------------------------------------------------ the library
_RING = None

SIGS = {
    "fa2_ring_ctx_create_with_backend": (_i, [_pvp, ctypes.POINTER(Backend), _i, _i]),
    "fa2_ring_default_backend": (_i, [ctypes.POINTER(Backend)]),
    "fa2_ring_ctx_destroy": (_i, [_vp]),
    "fa2_ring_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "fa2_ring_attention_forward": (_i, [_vp] * 6 + [_i, _i, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    "fa2_ring_attention_forward_causal": (_i, [_vp] * 6 + [_i, _i, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    "fa2_ring_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "fa2_ring_backward_block_workspace": (_i, [_i, _i, _i, _i, _i, _i, ctypes.POINTER(_sz), ctypes.POINTER(_sz)]),
    "fa2_ring_attention_backward": (_i, [_vp] * 10 + [_i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "fa2_ring_attention_backward_causal": (_i, [_vp] * 10 + [_i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "fa2_ring_exchange_kv": (_i, [_vp] * 5 + [_sz, _vp]),
}


def ring_lib():
    """libfa2_ring_mi355x.so with the entry points the tests call (loads without a GPU: no HIP call is made
    until a product backend is used)."""
    global _RING
    if _RING is None:
        lib_dir = os.path.join(ROOT, "cuda_flashattention_amd", "lib")
        ctypes.CDLL(os.path.join(lib_dir, "libfa2_mi355x.so"), mode=ctypes.RTLD_GLOBAL)
        h = ctypes.CDLL(os.path.join(lib_dir, "libfa2_ring_mi355x.so"), mode=ctypes.RTLD_GLOBAL)
        for n, (res, args) in SIGS.items():
            fn = getattr(h, n)
            fn.restype, fn.argtypes = res, args
        _RING = h
    return _RING


def zigzag_rows(N, rank, P):
    c = N // (2 * P)
    a, b = rank, 2 * P - 1 - rank
    return list(range(a * c, (a + 1) * c)) + list(range(b * c, (b + 1) * c))
