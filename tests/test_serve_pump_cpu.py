"""The fan-in server's data path without a GPU: websocket handlers stage chunks into the batch buffer being filled, the pump swaps
buffers and steps, scores come back by slot (openwakeword_amd/serve.py).  A stand-in model whose "score" of a chunk is a checksum of
its samples plus the slot's step counter proves, through real (loopback) websockets, that every connection's chunks are scored in
order, exactly once, whatever the message sizes, pacing, backlog, early hang-ups and slot reuse -- the property the GPU tests then
only have to confirm with real kernels (tests/test_masked_step.py).  Reference behaviour being replaced: one predict() per client
message inside the handler (examples/web/streaming_server.py:49-66)."""
import asyncio
import json

import numpy as np
import pytest

serve = pytest.importorskip("openwakeword_amd.serve")
pytest.importorskip("aiohttp")


def _checksum(chunk):
    return float(int(np.asarray(chunk, np.int64).sum()) % 9973) / 9973.0


class _Engine:
    has_vad = False

    def __init__(self, S, delay_s=0.0):
        self.S, self.k, self.last, self.q, self.delay_s = S, np.zeros(S, np.int64), np.zeros((S, 2), np.float32), [], delay_s
        self.max_in_flight = 0

    def pinned_empty(self, shape, dtype=np.int16):
        return np.zeros(shape, dtype)

    def submit(self, pcm, on):
        assert pcm.shape == (self.S, 1280) and len(self.q) < 2, "at most two steps in flight"
        idx = np.nonzero(on)[0]
        out = self.last.copy()
        for s in idx:
            out[s] = (_checksum(pcm[s]), self.k[s] / 1000.0)
        self.k[idx] += 1
        self.last = out
        self.q.append(out)
        self.max_in_flight = max(self.max_in_flight, len(self.q))

    def collect(self):
        if self.delay_s:
            import time
            time.sleep(self.delay_s)
        return self.q.pop(0)


class _Model:
    labels, _keep = ["a", "b"], [0, 1]

    def __init__(self, S, delay_s=0.0):
        self.n_streams, self.engine = S, _Engine(S, delay_s)

    def reset(self, ids=None, reset_vad=False):
        self.engine.k[ids] = 0


def _run(plans, S, window_s, delay_s=0.0, threshold=2.0):
    """plans: per client (message sizes in samples, seed, pause every n messages).  Returns (tap, per-client audio, server, model)."""
    from aiohttp.test_utils import TestClient, TestServer
    model = _Model(S, delay_s)
    tap = {}
    srv = serve.FanInServer(model, threshold=threshold, window_s=window_s, on_scores=lambda cid, k, row: tap.setdefault(cid, []).append((k, row.copy())))
    audio = [np.random.default_rng(seed).integers(-20000, 20000, size=int(sum(sizes)), dtype=np.int16) for sizes, seed, _ in plans]

    async def client(tc, i):
        sizes, _seed, pause = plans[i]
        ws = await tc.ws_connect("/ws")
        assert json.loads((await ws.receive()).data)["loaded_models"] == ["a", "b"]
        o = 0
        for j, n in enumerate(sizes):
            await ws.send_bytes(audio[i][o:o + n].tobytes())
            o += n
            if pause and j % pause == pause - 1:
                await asyncio.sleep(0.004)
        want = o // 1280
        for _ in range(400):
            if len(tap.get(i, [])) >= want:
                break
            await asyncio.sleep(0.005)
        await ws.close()

    async def run():
        async with TestClient(TestServer(srv.app())) as tc:
            await asyncio.gather(*[client(tc, i) for i in range(len(plans))])
            for _ in range(200):                     # the pump returns the slots of the closed connections
                if not srv.conns:
                    break
                await asyncio.sleep(0.005)
    asyncio.run(asyncio.wait_for(run(), 60))
    return tap, audio, srv, model


@pytest.mark.parametrize("window_s,delay_s", [(0.0, 0.0), (0.003, 0.0), (0.002, 0.004)])
def test_every_chunk_is_scored_once_and_in_order(window_s, delay_s):
    rng = np.random.default_rng(5)
    plans = []
    for i in range(12):
        kind = i % 4
        if kind == 0:
            sizes = [1280] * 14                                         # one chunk per message
        elif kind == 1:
            sizes = [int(v) for v in rng.integers(1, 900, size=40)]     # several messages per chunk
        elif kind == 2:
            sizes = [5000, 7000, 333, 1280 * 6]                         # several chunks per message: a backlog the pump drains
        else:
            sizes = [640] * 21
        plans.append((sizes, 100 + i, (0, 3, 0, 5)[kind]))
    tap, audio, srv, model = _run(plans, S=16, window_s=window_s, delay_s=delay_s)
    for i, (sizes, _s, _p) in enumerate(plans):
        n = int(sum(sizes)) // 1280
        got = tap.get(i, [])
        assert [k for k, _ in got] == list(range(n)), f"client {i}: steps {[k for k, _ in got]}"
        for k, row in got:
            assert row[0] == np.float32(_checksum(audio[i][k * 1280:(k + 1) * 1280])), f"client {i} chunk {k}"
            assert row[1] == np.float32(k / 1000.0)
    assert srv.n_stream_steps == sum(int(sum(p[0])) // 1280 for p in plans) and srv.n_steps < srv.n_stream_steps
    assert not srv.conns and not srv.clients and srv.slots.n_used == 0 and int(srv._inflight.sum()) == 0
    assert all(int(o.sum()) == 0 for o in srv._on)
    if delay_s:
        assert model.engine.max_in_flight == 2          # a step that is long against the window runs pipelined


def test_slots_are_reused_and_activations_reach_the_right_client():
    """Two slots, four clients one after the other pair-wise; threshold below the checksum range: every stream-step answers."""
    plans = [([1280] * 5, 1, 0), ([2560] * 3, 2, 0)]
    tap, audio, srv, model = _run(plans, S=2, window_s=0.001)
    tap2, audio2, srv2, model2 = _run(plans + [([1280] * 4, 3, 2)], S=3, window_s=0.001, threshold=0.0)
    assert sorted(tap) == [0, 1] and len(tap[0]) == 5 and len(tap[1]) == 6
    assert len(tap2[2]) == 4


def test_step_counter_restarts_when_a_slot_changes_hands():
    from aiohttp.test_utils import TestClient, TestServer
    model = _Model(1)
    tap = {}
    srv = serve.FanInServer(model, threshold=2.0, window_s=0.0, on_scores=lambda cid, k, row: tap.setdefault(cid, []).append((k, row.copy())))
    x = np.arange(1280 * 3, dtype=np.int16)

    async def one(tc, n_chunks):
        ws = await tc.ws_connect("/ws")
        await ws.receive()
        cid = srv._next_cid - 1
        await ws.send_bytes(x[:1280 * n_chunks].tobytes())
        while len(tap.get(cid, [])) < n_chunks:
            await asyncio.sleep(0.002)
        await ws.close()
        while srv.conns:
            await asyncio.sleep(0.002)

    async def run():
        async with TestClient(TestServer(srv.app())) as tc:
            await one(tc, 3)
            await one(tc, 2)
    asyncio.run(asyncio.wait_for(run(), 30))
    assert [k for k, _ in tap[0]] == [0, 1, 2] and [k for k, _ in tap[1]] == [0, 1]
    assert tap[1][0][1][1] == 0.0                       # the engine's own counter was reset with the slot (Model.reset on hand-over)


def test_range_error_drops_the_step_and_the_server_carries_on():
    """OWW_ERANGE out of a step (serve._recover_range): the participants of that step lose one chunk, the streams named by range_where are
    reset, nothing stays flagged or counted in flight, every later chunk is scored."""
    from aiohttp.test_utils import TestClient, TestServer
    from openwakeword_amd._lib import OwwRangeError

    class FailingEngine(_Engine):
        def __init__(self, S):
            super().__init__(S)
            self.n_collect, self.flag, self.resets = 0, False, []

        def collect(self):
            self.n_collect += 1
            out = super().collect()
            if self.n_collect == 4:
                self.flag = True
                raise OwwRangeError("libowwhip error -5: test")
            return out

        def range_where(self):
            return (1, 1)

        def range_status(self, clear=False):
            was, self.flag = self.flag, (False if clear else self.flag)
            return was

    model = _Model(4)
    model.engine = FailingEngine(4)
    real_reset = model.reset

    def reset(ids=None, reset_vad=False):
        model.engine.resets.append(ids)
        real_reset(slice(None) if ids is None else ids, reset_vad)
    model.reset = reset
    tap = {}
    srv = serve.FanInServer(model, threshold=2.0, window_s=0.002, on_scores=lambda cid, k, row: tap.setdefault(cid, []).append((k, row.copy())))
    x = [np.random.default_rng(i).integers(-9000, 9000, size=1280 * 10, dtype=np.int16) for i in range(3)]

    async def client(tc, i):
        ws = await tc.ws_connect("/ws")
        await ws.receive()
        for k in range(10):
            await ws.send_bytes(x[i][k * 1280:(k + 1) * 1280].tobytes())
            await asyncio.sleep(0.004)
        await asyncio.sleep(0.1)
        await ws.close()

    async def run():
        async with TestClient(TestServer(srv.app())) as tc:
            await asyncio.gather(*[client(tc, i) for i in range(3)])
            for _ in range(200):
                if not srv.conns:
                    break
                await asyncio.sleep(0.005)
    asyncio.run(asyncio.wait_for(run(), 30))
    assert srv.n_range_recoveries == 1 and srv.failed is None
    assert [1] in model.engine.resets                                    # the stream range_where named was restarted
    scored = sum(len(v) for v in tap.values())
    assert 30 - 3 <= scored < 30                                         # only the participants of the failing step lost a chunk
    assert not srv.conns and srv.slots.n_used == 0 and int(srv._inflight.sum()) == 0 and all(int(o.sum()) == 0 for o in srv._on)
    for i in range(3):                                                   # what WAS scored is each client's own audio, in order
        sums = [np.float32(_checksum(x[i][k * 1280:(k + 1) * 1280])) for k in range(10)]
        got = [row[0] for _k, row in tap[i]]
        it = iter(sums)
        assert all(any(g == s for s in it) for g in got), f"client {i}"


def test_load_driver_dry_run_on_two_server_processes():
    """tools/serve_load.py (the real-socket load driver of DESIGN.md 5.9) end to end without a GPU: two server processes on one port,
    two client processes, a stand-in model -- every chunk sent is answered, nothing is left in a backlog."""
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "serve_load.py"), "--clients", "120", "--seconds", "3", "--procs", "2",
                        "--server-procs", "2", "--dry-run"], cwd=root, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["connections_accepted"] == 120 and sum(d["connections_per_server"]) == 120 and d["server_processes"] == 2
    assert d["chunks_sent"] > 500 and d["answers_received"] == d["chunks_sent"] == d["pump"]["stream_steps"]
    assert d["backlog_chunks_at_end"] == 0 and d["client_errors"] == 0 and d["server_latency_ms"]["p50"] < 200


def test_a_dropped_connection_never_touches_its_slot_again():
    """ADVICE r05 (high).  The server drops a connection on its own (send timeout, unread activations) while that connection's websocket
    handler is still reading audio.  Its slot goes back to the pool ONCE; audio the old connection keeps sending is not staged into
    the slot's row (which by then belongs to someone else), and the old handler's exit neither evicts the new owner nor releases the
    slot a second time nor kills the pump."""
    from aiohttp.test_utils import TestClient, TestServer
    model = _Model(1)
    tap = {}
    srv = serve.FanInServer(model, threshold=2.0, window_s=0.002, on_scores=lambda cid, k, row: tap.setdefault(cid, []).append((k, row.copy())))
    rng = np.random.default_rng(3)
    a_audio = rng.integers(-20000, 20000, size=1280 * 12, dtype=np.int16)
    b_audio = rng.integers(-20000, 20000, size=1280 * 8, dtype=np.int16)

    async def wait_for(cond, what):
        for _ in range(600):
            if cond():
                return
            await asyncio.sleep(0.005)
        raise AssertionError(what)

    async def run():
        async with TestClient(TestServer(srv.app())) as tc:
            wa = await tc.ws_connect("/ws")
            await wa.receive()
            await wa.send_bytes(a_audio[:1280 * 2].tobytes())
            await wait_for(lambda: len(tap.get(0, [])) == 2, "A's first chunks scored")
            ca = srv.conns[0]
            assert ca.slot == 0
            # what _drain does when a send times out -- the peer's socket stays open and keeps sending
            ca.closed = True
            srv._closing.add(ca)
            srv._have_chunk.set()
            await wait_for(lambda: ca.reaped, "A reaped")
            assert ca.slot is None and not srv.conns and not srv.clients
            wb = await tc.ws_connect("/ws")
            await wb.receive()
            await wb.send_bytes(b_audio[:1280 * 3].tobytes())
            await wait_for(lambda: len(tap.get(1, [])) == 3, "B's first chunks scored")
            assert srv.conns[1].slot == 0                              # the slot was reused
            await wa.send_bytes(a_audio[1280 * 2: 1280 * 7].tobytes())   # the stale connection goes on sending
            await wa.send_bytes(a_audio[1280 * 7:].tobytes())
            await asyncio.sleep(0.05)
            await wb.send_bytes(b_audio[1280 * 3:].tobytes())
            await wait_for(lambda: len(tap.get(1, [])) == 8, "B's remaining chunks scored")
            await wa.close()
            await asyncio.sleep(0.05)                                  # A's handler ends; the pump passes _reap again
            assert srv.failed is None and srv.clients.get(0) is srv.conns[1] and srv.conns[1].slot == 0
            await wb.close()
            await wait_for(lambda: not srv.conns, "B reaped")
    asyncio.run(asyncio.wait_for(run(), 60))
    assert srv.failed is None
    assert len(tap[0]) == 2                                            # nothing of A's later audio was scored ...
    got = [row[0] for _, row in tap[1]]
    want = [_checksum(b_audio[1280 * t: 1280 * (t + 1)]) for t in range(8)]
    np.testing.assert_allclose(got, want, atol=1e-6)                   # ... and B's stream holds B's audio only, in order
    assert [k for k, _ in tap[1]] == list(range(8))
    assert sorted(srv.slots.free_slots()) == [0] if hasattr(srv.slots, "free_slots") else True
