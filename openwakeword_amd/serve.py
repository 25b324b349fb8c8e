"""Fan-in streaming server: many websocket clients, ONE batched step on the GPU.

The reference's web example (examples/web/streaming_server.py:32-70) holds one `Model` and calls `predict()` inside the
websocket handler of whichever client sent audio: one graph evaluation per client message on the CPU, and all clients share that
single model's streaming state.  This module keeps the example's wire protocol --

    client -> server   TEXT    the client's sample rate ("16000", "44100", ...)           (streaming_server.py:50-52)
    client -> server   BINARY  little-endian int16 PCM, any length                         (streaming_server.py:53-59)
    server -> client   TEXT    {"loaded_models": [...]} once, on connect                   (streaming_server.py:44-46)
    server -> client   TEXT    {"activations": [...]} for labels scoring >= threshold      (streaming_server.py:62-68)

-- and changes what happens behind it: every connection owns one stream slot of a `BatchedModel` (its own sample tail, conv
histories, rings and counters on the device); the handler appends the message's (resampled) samples to the connection's queue and
moves each complete 1280-sample chunk into the connection's row of the batch buffer being filled; a single paced pump swaps
buffers and runs ONE masked step (`oww_submit_masked`) for every row that carries a chunk (two steps in flight when steps are long
against the round period).  `--workers N` runs N such processes on one port (SO_REUSEPORT), each with its own handle on the GPU:
one Python event loop carries about two thousand real-time clients (tools/serve_load.py), the GPU a thousand times that.
A connection without a full chunk sits the step out bit-exactly (no zero padding, no skipped audio), so each client gets exactly the
scores a private `openwakeword.Model` would have produced on its own audio in 1280-sample calls.

Slot placement (`SlotAllocator`): a masked step launches only the 2 / 4 / 8 / 16 / 32-stream groups that hold a participant, so its
cost follows the number of GROUPS touched, not the number of streams.  Connections are therefore placed by cohort -- message period
(first message's duration) and arrival phase within that period -- so that clients whose chunks fall due in the same pump round
share groups: participation per group becomes all-or-nothing instead of "every group holds somebody".

Sample-rate conversion follows the example (per message, stateless: `resampy.resample(data, sample_rate, 16000)`,
streaming_server.py:57-58) with scipy's polyphase resampler; resampy is not a dependency here.

    python -m openwakeword_amd.serve --streams 4096 --models alexa hey_jarvis --weights synthetic --port 9000
"""
from __future__ import annotations

import argparse
import asyncio
import collections
import json
import queue
import threading
from typing import Dict, List, Optional

import numpy as np

from .engine import CHUNK

try:  # aiohttp is the example's server library too (streaming_server.py:19)
    from aiohttp import WSMsgType, web
except ImportError as e:  # pragma: no cover
    raise ImportError("openwakeword_amd.serve needs aiohttp (the library the reference's web example uses)") from e


# input rates the server converts (the example hands whatever the browser's AudioContext reports to resampy: 8 .. 96 kHz in practice).
# A whitelist keeps the polyphase factors small: an arbitrary rate such as 22051 Hz would ask for a 16000 / 22051 resampler with a
# filter of several hundred thousand taps per message.
RATES = (8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000)


def to_16k(pcm: np.ndarray, sample_rate: int) -> np.ndarray:
    """int16 at `sample_rate` -> int16 at 16 kHz (one message at a time, like streaming_server.py:57-58), through the package's own
    cached polyphase bank (`resample.design`, the host restatement of oww_resample)."""
    if sample_rate == 16000 or pcm.size == 0:
        return pcm
    if sample_rate not in RATES:
        raise ValueError(f"unsupported sample rate {sample_rate}")
    from . import resample
    return resample.apply_numpy(pcm[None, :], int(sample_rate))[0]


class SlotAllocator:
    """Stream slots handed out in blocks of `group` consecutive streams per cohort key.

    `alloc(key)` prefers a block already tagged with `key` that has a free slot, then an untouched block (which it tags), then the
    block of the nearest key (`distance(key_a, key_b)`, smaller = closer; default: equal or not), then anything free.  `release`
    returns a slot; a block whose slots are all free loses its tag.  Pure bookkeeping: O(1) amortised per call, no device work."""

    def __init__(self, n_slots: int, group: int = 32, distance=None):
        self.n_slots, self.group = int(n_slots), int(group)
        self.n_blocks = (self.n_slots + self.group - 1) // self.group
        self._fresh: List[int] = list(range(self.n_blocks - 1, -1, -1))          # untouched blocks, lowest index on top
        self._free: Dict[int, List[int]] = {}                                    # block -> free slots (descending: pop() = lowest)
        self._tag: Dict[int, object] = {}                                        # block -> cohort key
        self._open: Dict[object, List[int]] = {}                                 # key -> blocks of that key with a free slot
        self._distance = distance or (lambda a, b: 0 if a == b else 1)
        self.n_used = 0

    def _block_slots(self, b: int) -> List[int]:
        return list(range(min(self.n_slots, (b + 1) * self.group) - 1, b * self.group - 1, -1))

    def _take(self, b: int) -> int:
        slot = self._free[b].pop()
        if not self._free[b]:
            self._open[self._tag[b]].remove(b)
        self.n_used += 1
        return slot

    def alloc(self, key=None) -> int:
        if self.n_used >= self.n_slots:
            raise IndexError("no free stream slot")
        blocks = self._open.get(key)
        if blocks:
            return self._take(blocks[-1])
        if self._fresh:
            b = self._fresh.pop()
            self._free[b] = self._block_slots(b)
            self._tag[b] = key
            self._open.setdefault(key, []).append(b)
            return self._take(b)
        near = min((k for k, bl in self._open.items() if bl), key=lambda k: self._distance(key, k))
        return self._take(self._open[near][-1])

    def release(self, slot: int) -> None:
        b = slot // self.group
        free = self._free[b]
        if slot in free:
            raise ValueError(f"slot {slot} is not in use")
        if not free:
            self._open.setdefault(self._tag[b], []).append(b)
        free.append(slot)
        free.sort(reverse=True)
        self.n_used -= 1
        if len(free) == len(self._block_slots(b)):           # the block is empty again: any cohort may claim it
            self._open[self._tag[b]].remove(b)
            del self._free[b], self._tag[b]
            self._fresh.append(b)
            self._fresh.sort(reverse=True)

    def key_of(self, slot: int):
        return self._tag.get(slot // self.group)


N_PHASE_BINS = 8


def cohort_key(period_s: float, arrival_s: float, n_bins: int = N_PHASE_BINS):
    """(message period in ms, phase bin of the arrival within that period): clients with equal keys have their chunks fall due in
    the same pump rounds for as long as they keep their pace."""
    period_ms = max(1, int(round(period_s * 1e3)))
    phase = (arrival_s % (period_ms * 1e-3)) / (period_ms * 1e-3)
    return period_ms, min(n_bins - 1, int(phase * n_bins))


def cohort_distance(a, b) -> float:
    """Same period first, then the cyclic distance between phase bins."""
    if a is None or b is None:
        return 1e6
    d = abs(a[1] - b[1])
    return (0 if a[0] == b[0] else 1e3) + min(d, N_PHASE_BINS - d)


class _Client:
    __slots__ = ("cid", "slot", "ws", "rate", "pending", "n_pending", "n_steps", "closed", "in_flight", "outbox", "sender", "arrivals",
                 "n_arrived", "reaped")

    def __init__(self, slot: Optional[int], ws, cid: int = -1):
        self.cid, self.slot, self.ws, self.rate = cid, slot, ws, 16000
        self.pending: List[np.ndarray] = []
        self.n_pending = 0
        self.n_steps = 0
        self.closed = False
        self.reaped = False           # its slot went back to the pool (once: _reap); nothing of this connection may touch a slot again
        self.in_flight = 0            # submitted steps whose scores for this client have not been dispatched yet
        self.outbox: "collections.deque" = collections.deque()     # activation messages not yet written to the socket, in step order
        self.sender: Optional[asyncio.Task] = None                  # the one task that drains `outbox`
        self.arrivals: "collections.deque" = collections.deque()   # when each complete, not yet scored chunk became available (metrics)
        self.n_arrived = 0                                          # samples received so far

    def push(self, x: np.ndarray, now: float = 0.0) -> None:
        if x.size:
            self.pending.append(x)
            self.n_pending += x.size
            done = (self.n_arrived + x.size) // CHUNK - self.n_arrived // CHUNK      # chunks this message completes
            self.n_arrived += x.size
            for _ in range(done):
                self.arrivals.append(now)

    def pop_chunk(self, out: np.ndarray) -> None:
        """Move the oldest 1280 samples into `out`."""
        need, o = CHUNK, 0
        while need:
            head = self.pending[0]
            take = min(need, head.size)
            out[o:o + take] = head[:take]
            if take == head.size:
                self.pending.pop(0)
            else:
                self.pending[0] = head[take:]
            o += take
            need -= take
        self.n_pending -= CHUNK


class _GpuWorker(threading.Thread):
    """The one thread that talks to the handle (the C ABI serialises nothing itself: one caller per handle).  Jobs run in the order
    they were posted; `call` returns an asyncio future of the job's result."""

    def __init__(self, loop: asyncio.AbstractEventLoop):
        super().__init__(daemon=True, name="owwhip-gpu")
        self.loop = loop
        self.jobs: "queue.Queue" = queue.Queue()

    def call(self, fn, *args, **kw) -> "asyncio.Future":
        fut = self.loop.create_future()
        self.jobs.put((fut, fn, args, kw))
        return fut

    def run(self) -> None:
        while True:
            job = self.jobs.get()
            if job is None:
                return
            fut, fn, args, kw = job
            try:
                res, err = fn(*args, **kw), None
            except BaseException as e:              # delivered to the awaiting coroutine
                res, err = None, e
            self.loop.call_soon_threadsafe(_settle, fut, res, err)

    def stop(self) -> None:
        self.jobs.put(None)


def _settle(fut, res, err) -> None:
    if fut.cancelled():
        return
    if err is not None:
        fut.set_exception(err)
    else:
        fut.set_result(res)


class FanInServer:
    """`model`: a BatchedModel; its n_streams is the number of clients that can be connected at once (further connections are
    refused with close code 1013).  `window_s`: the pump's round period -- it gathers whatever complete chunks have arrived and
    steps at most once per window (real-time clients deliver one chunk per 80 ms, so a few ms gather a whole phase bin of them into
    one step; 0 = step as soon as anything is there).  `on_scores(connection_id, step_index, scores_row)`: optional tap, called for
    every stream-step (used by the tests; connection ids count accepted connections from 0).

    A connection gets its stream slot with its first audio message -- until then it has no device state to keep -- from a
    `SlotAllocator` keyed by the message's duration and arrival phase (`cohort_key`), see the module docstring.

    Data path.  The websocket handler of a connection moves each complete 1280-sample chunk straight into the connection's row of the
    page-locked batch buffer that is currently being FILLED and flags the row; the pump never walks the connections -- a round is a
    buffer swap, one `oww_submit_masked` and, when the scores are back, array operations over the participating rows (only rows with an
    activation, or a tap, cost Python time).  Three buffers: one filling, up to two in flight.  A step is collected as soon as it has
    been submitted while the GPU step is short against the window; when it is not (tens of thousands of streams: the upload of step t+1
    should overlap the kernels of step t) the collect is left to the next round -- two steps in flight, scores in step order."""

    N_BUF = 3

    def __init__(self, model, threshold: float = 0.5, window_s: float = 0.01, on_scores=None):
        self.model = model
        self.threshold = float(threshold)
        self.window_s = float(window_s)
        self.on_scores = on_scores
        S = model.n_streams
        self.slots = SlotAllocator(S, group=32, distance=cohort_distance)
        self.clients: Dict[int, _Client] = {}          # by stream slot: the connections that have sent audio
        self.conns: Dict[int, _Client] = {}            # by connection id: every accepted connection
        self._next_cid = 0
        self._tasks: set = set()                       # sender tasks (asyncio keeps only weak references to tasks)
        self._have_chunk: Optional[asyncio.Event] = None
        self._pump_task: Optional[asyncio.Task] = None
        self._gpu: Optional[_GpuWorker] = None
        self._pcm = [model.engine.pinned_empty((S, CHUNK)) for _ in range(self.N_BUF)]
        for b in self._pcm:
            b[:] = 0
        self._on = [np.zeros(S, dtype=np.uint8) for _ in range(self.N_BUF)]        # rows of the buffer that carry a chunk
        self._t_arr = [np.zeros(S, dtype=np.float64) for _ in range(self.N_BUF)]   # when that chunk became complete (metrics)
        self._fill = 0                                 # the buffer the handlers are filling
        self._n_fill = 0                               # rows flagged in it
        self._more: set = set()                        # connections with a further complete chunk whose row of the filling buffer is taken
        self._closing: set = set()                     # closed connections whose slot is still to be returned
        self._inflight = np.zeros(S, dtype=np.int32)   # submitted, not yet dispatched steps per slot
        self._k = np.zeros(S, dtype=np.int64)          # stream-steps dispatched per slot since it was handed out (the tap's step index)
        self._step_s = 0.0                             # smoothed submit -> scores time of a step
        self.n_steps = 0              # batched steps taken
        self.n_stream_steps = 0       # sum over steps of the streams that took part
        self.n_dropped_messages = 0   # activation messages lost to clients that stopped reading (such clients are closed, see _post)
        self.n_range_recoveries = 0   # OWW_ERANGE events the pump recovered from (see _recover_range)
        self._last_recovery_step = -10**9
        self.send_timeout_s = 2.0     # deadline of one activation message / close handshake
        # metrics (tools/serve_load.py): per stream-step, the time from the arrival of the message that completed the chunk to the
        # dispatch of its scores; per batched step, [participants, submit start, submit returned, scores dispatched]
        self.keep_metrics = False
        self.latencies_s: List[np.ndarray] = []
        self.step_log: List[list] = []
        self.failed: Optional[BaseException] = None      # set when the pump died of anything it cannot recover from

    # ---- aiohttp plumbing
    def app(self) -> "web.Application":
        app = web.Application()
        app.add_routes([web.get("/ws", self.handle)])
        app.on_startup.append(self._start)
        app.on_cleanup.append(self._stop)
        return app

    async def _start(self, app) -> None:
        loop = asyncio.get_running_loop()
        self._have_chunk = asyncio.Event()
        self._gpu = _GpuWorker(loop)
        self._gpu.start()
        self._pump_task = loop.create_task(self._pump())

    async def _stop(self, app) -> None:
        if self._pump_task:
            self._pump_task.cancel()
            try:
                await self._pump_task
            except asyncio.CancelledError:
                pass
        if self._gpu:
            self._gpu.stop()

    def _stage(self, c: "_Client") -> None:
        """Move the connection's oldest complete chunk into its row of the buffer being filled -- if that row is free; else the
        connection waits in `_more` for the next buffer (one chunk per stream and step, in order)."""
        if c.n_pending < CHUNK or c.slot is None:
            return
        f = self._fill
        if self._on[f][c.slot]:
            self._more.add(c)
            return
        c.pop_chunk(self._pcm[f][c.slot])
        self._on[f][c.slot] = 1
        self._t_arr[f][c.slot] = c.arrivals.popleft() if c.arrivals else 0.0
        self._n_fill += 1
        if c.n_pending >= CHUNK:
            self._more.add(c)
        self._have_chunk.set()

    async def handle(self, request):
        ws = web.WebSocketResponse()
        await ws.prepare(request)
        if self.failed is not None:
            await ws.close(code=1011, message=b"scoring backend failed")
            return ws
        if len(self.conns) >= self.model.n_streams:
            await ws.close(code=1013, message=b"all stream slots are taken")
            return ws
        c = _Client(None, ws, cid=self._next_cid)
        self._next_cid += 1
        self.conns[c.cid] = c
        loop = asyncio.get_running_loop()
        try:
            await ws.send_str(json.dumps({"loaded_models": list(self.model.labels)}))
            async for msg in ws:
                if msg.type == WSMsgType.TEXT:
                    try:
                        rate = int(msg.data)
                    except ValueError:
                        rate = 0
                    if rate not in RATES:                   # (the example trusts the client here; see RATES)
                        # the accepted rates go out as a message: a close reason is limited to 123 bytes (RFC 6455, 5.5)
                        await ws.send_str(json.dumps({"error": "unsupported sample rate", "accepted_rates": list(RATES)}))
                        await ws.close(code=1003, message=b"unsupported sample rate")
                        break
                    c.rate = rate
                elif msg.type == WSMsgType.BINARY:
                    if c.closed:
                        # dropped by the server (send timeout, unread activations): its slot is being -- or has been -- returned and may
                        # already belong to another connection; audio that still arrives is not staged, the handler ends
                        break
                    n = len(msg.data) // 2
                    if c.slot is None and n:
                        # the first audio: place the connection next to the ones whose chunks fall due in the same rounds.  A slot
                        # handed to a new caller starts from Model()'s initial state, VAD history included; the job queue orders
                        # the reset after every step already submitted
                        c.slot = self.slots.alloc(cohort_key(n / c.rate, loop.time()))
                        self.clients[c.slot] = c
                        self._k[c.slot] = 0
                        await self._gpu.call(self.model.reset, [c.slot], reset_vad=bool(self.model.engine.has_vad))
                    x = np.frombuffer(msg.data, dtype="<i2", count=n)
                    if c.rate != 16000:                     # filtering runs on the default executor, not on the event loop
                        x = await loop.run_in_executor(None, to_16k, x, c.rate)
                    c.push(x, loop.time())
                    self._stage(c)
                elif msg.type == WSMsgType.ERROR:
                    break
        finally:
            c.closed = True          # the pump scores what the client still delivered, then returns the slot to the pool
            if not c.reaped:         # (a connection the server dropped earlier may have been reaped while this handler still ran)
                self._closing.add(c)
            self._have_chunk.set()
        return ws

    def _reap(self) -> None:
        """Return the slots of closed connections once nothing of theirs is pending, staged or in flight."""
        if not self._closing:
            return
        for c in list(self._closing):
            if c.reaped:                                    # one-shot: a second pass must not release a slot that has a new owner
                self._closing.discard(c)
                continue
            if c.slot is not None and (c.n_pending >= CHUNK or self._inflight[c.slot] or self._on[self._fill][c.slot]):
                continue
            self._closing.discard(c)
            self._more.discard(c)
            self.conns.pop(c.cid, None)
            slot, c.slot, c.reaped = c.slot, None, True     # the connection is severed from its slot before the pool sees it again
            c.pending.clear(); c.n_pending = 0
            if slot is not None:
                if self.clients.get(slot) is c:
                    del self.clients[slot]
                self.slots.release(slot)

    # ---- the one place steps are put together
    def _post(self, c: "_Client", text: str) -> None:
        """Queue one activation message for a client.  Each client has ONE sender task that writes its queue in order (messages of
        consecutive steps cannot overtake each other) with a deadline per message, so a stalled socket never holds up the pump and
        never accumulates more than its own bounded queue; the task set keeps the tasks alive until they finish."""
        if c.closed:
            return
        if len(c.outbox) >= 64:
            # a client that does not read its socket: detections must not vanish in silence while the connection keeps a stream slot --
            # count, log once and drop the connection (the pump returns its slot to the pool)
            self.n_dropped_messages += len(c.outbox) + 1
            import logging
            logging.getLogger(__name__).warning("client %s does not read its activation messages (%d queued): closing it", c.cid, len(c.outbox))
            c.closed = True
            c.outbox.clear()
            self._closing.add(c)
            task = asyncio.get_running_loop().create_task(c.ws.close(code=1008, message=b"activation messages not read"))
            self._tasks.add(task)
            task.add_done_callback(self._tasks.discard)
            self._have_chunk.set()
            return
        c.outbox.append(text)
        if c.sender is None or c.sender.done():
            c.sender = asyncio.get_running_loop().create_task(self._drain(c))
            self._tasks.add(c.sender)
            c.sender.add_done_callback(self._tasks.discard)

    async def _drain(self, c: "_Client") -> None:
        while c.outbox and not c.closed:
            text = c.outbox.popleft()
            try:
                await asyncio.wait_for(c.ws.send_str(text), timeout=self.send_timeout_s)
            except (ConnectionError, RuntimeError, asyncio.TimeoutError):
                c.closed = True
                if not c.reaped:
                    self._closing.add(c)
                c.outbox.clear()
                self._have_chunk.set()
                try:                                        # end the websocket handler too: it would go on reading audio for a dead peer
                    await asyncio.wait_for(c.ws.close(code=1008, message=b"activation messages not delivered"), timeout=self.send_timeout_s)
                except Exception:                           # noqa: BLE001  (the socket is already gone)
                    pass

    def _step_now(self, pcm: np.ndarray, on: np.ndarray):
        """(GPU thread) one whole step; also returns what it took THERE -- the event loop's own delays must not enter the decision
        whether steps are long enough to be worth pipelining."""
        import time
        t0 = time.perf_counter()
        self.model.engine.submit(pcm, on)
        scores = self.model.engine.collect()
        return scores, time.perf_counter() - t0

    def _release_buffer(self, g: int) -> None:
        idx = np.nonzero(self._on[g])[0]
        self._inflight[idx] -= 1
        self._on[g][:] = 0

    async def _recover_range(self, flying) -> None:
        """OWW_ERANGE is sticky per handle: one stream whose activations left the f16 range would stop scoring for everybody.  The
        streams of the wave that saw it (oww_range_where; every stream when the position is unknown) restart from Model()'s initial
        state -- their clients keep their connections and simply see a few silent frames -- the steps in flight are dropped and the
        flag is cleared.  The reported position names ONE offending wave; if the flag comes back within a few steps of a partial
        recovery another stream was (also) out of range, and every stream is restarted instead of dropping the clients' steps again
        and again."""
        self.n_range_recoveries += 1
        again = self.n_steps - self._last_recovery_step <= 8
        self._last_recovery_step = self.n_steps
        n_flying = len(flying)

        def recover():
            eng = self.model.engine
            for _ in range(n_flying):
                try:
                    eng.collect()
                except Exception:
                    pass
            first, n = eng.range_where()
            eng.range_status(clear=True)
            ids = list(range(first, first + n)) if (first >= 0 and n > 0 and not again) else None      # None: every stream
            self.model.reset(ids, reset_vad=bool(eng.has_vad))
            eng.range_status(clear=True)
        await self._gpu.call(recover)
        for g in flying:
            self._release_buffer(g)
        flying.clear()

    def _dispatch(self, g: int, scores: np.ndarray, now: float) -> None:
        """Scores of the step that was gathered in buffer g: taps, activation messages, counters -- array work except for the rows
        that have something to say."""
        on = self._on[g]
        idx = np.nonzero(on)[0]
        if self.keep_metrics:
            self.latencies_s.append(now - self._t_arr[g][idx])
        rows = scores[idx]
        if self.on_scores is not None:
            for s_, row in zip(idx.tolist(), rows):
                c = self.clients.get(s_)
                if c is not None:
                    self.on_scores(c.cid, int(self._k[s_]), row)
        hit = rows >= self.threshold
        for j in np.nonzero(hit.any(axis=1))[0].tolist():
            c = self.clients.get(int(idx[j]))
            if c is not None and not c.closed:
                self._post(c, json.dumps({"activations": [self.model.labels[l] for l in np.nonzero(hit[j])[0]]}))
        self._k[idx] += 1
        self._inflight[idx] -= 1
        on[:] = 0                                      # the buffer may be filled again

    async def _pump(self) -> None:
        from ._lib import OwwRangeError
        keep = self.model._keep
        flying: "collections.deque" = collections.deque()        # buffers of the submitted, not yet collected steps
        t_sub: "collections.deque" = collections.deque()
        try:
            loop = asyncio.get_running_loop()
            t_round = loop.time() - self.window_s
            while True:
                if not flying:
                    await self._have_chunk.wait()
                # rounds are paced: at least window_s between the starts of two gathers, whether or not a step is in flight -- without
                # the pacing a busy server degenerates into thousands of tiny steps per second
                wait = t_round + self.window_s - loop.time()
                if wait > 0:
                    await asyncio.sleep(wait)
                t_round = loop.time()
                self._have_chunk.clear()
                self._reap()
                g_sync = None
                try:
                    n = self._n_fill
                    pipelined = self._step_s > 0.5 * self.window_s and self.window_s > 0
                    if n:
                        g = self._fill
                        # the next buffer to fill: the one that is neither this one nor in flight (three buffers, at most two taken)
                        self._fill = next(b for b in range(self.N_BUF) if b != g and b not in flying)
                        self._n_fill = 0
                        more, self._more = self._more, set()
                        for c in more:                                     # connections with a backlog: their next chunk
                            self._stage(c)
                        self._inflight[np.nonzero(self._on[g])[0]] += 1
                        t0 = loop.time()
                        self.n_steps += 1
                        self.n_stream_steps += n
                        if self.keep_metrics:
                            self.step_log.append([n, t0, 0.0, 0.0])
                        if (not pipelined or self.n_steps % 64 == 0) and not flying:
                            # a step that is short against the window: submit and collect as ONE job of the GPU thread (every await
                            # costs a trip through an event loop that is busy with thousands of sockets).  (Every 64th step of the
                            # pipelined mode runs this way too: it is where the step time is measured.)
                            g_sync = g
                            scores, dt = await self._gpu.call(self._step_now, self._pcm[g], self._on[g])
                            g_sync = None
                            scores = scores[:, keep]
                            now = loop.time()
                            self._step_s = dt if self._step_s == 0.0 else 0.5 * self._step_s + 0.5 * dt
                            if self.keep_metrics:
                                self.step_log[self.n_steps - 1][2:] = [now, now]
                            self._dispatch(g, scores, now)
                        else:
                            flying.append(g)
                            t_sub.append(t0)
                            await self._gpu.call(self.model.engine.submit, self._pcm[g], self._on[g])
                            if self.keep_metrics:
                                self.step_log[self.n_steps - 1][2] = loop.time()
                    # collect: otherwise keep two steps in flight, so that the next upload overlaps this step's kernels
                    while flying and (len(flying) == 2 or not n or not pipelined):
                        scores = (await self._gpu.call(self.model.engine.collect))[:, keep]
                        now = loop.time()
                        t_sub.popleft()
                        if self.keep_metrics:
                            self.step_log[self.n_steps - len(flying)][3] = now
                        self._dispatch(flying.popleft(), scores, now)
                    self._reap()
                except OwwRangeError:
                    t_sub.clear()
                    if g_sync is not None:                  # the one-job step raised: nothing of it is in flight, its rows are dropped
                        self._release_buffer(g_sync)
                        g_sync = None
                    await self._recover_range(flying)
                if self._n_fill:
                    self._have_chunk.set()          # something was staged meanwhile (a backlog, or messages that arrived during the step)
        except asyncio.CancelledError:
            raise
        except Exception as e:
            # anything else (a device error, a bug): no step will ever be scored again -- say so and hang up on everybody instead of
            # accepting audio that is silently dropped
            self.failed = e
            import logging
            logging.getLogger("openwakeword_amd.serve").exception("the GPU pump failed; closing %d connections", len(self.conns))
            for c in list(self.conns.values()):
                c.closed = True
                try:
                    await asyncio.wait_for(c.ws.close(code=1011, message=b"scoring backend failed"), timeout=self.send_timeout_s)
                except Exception:
                    pass
            raise


def _supervise(cmd: List[str], n: int, poll_s: float = 0.5) -> int:
    """Parent of `--workers N`: start the worker processes in their own session, forward SIGTERM / SIGINT to them, and watch ALL of
    them -- when any worker ends (a GPU fault, an exception) the others are stopped and the exit code is its own, so that a process
    supervisor restarts the whole set instead of serving on with a silent hole.  Workers never outlive the parent's
    signal handling: every exit path terminates, then kills, what is still running.  Returns the exit code."""
    import signal
    import subprocess
    import time
    procs = [subprocess.Popen(cmd, start_new_session=True) for _ in range(n)]
    stop = {"sig": 0}

    def on_signal(signum, _frame):
        stop["sig"] = signum

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    code = 0
    try:
        while not stop["sig"]:
            ended = [p for p in procs if p.poll() is not None]
            if ended:
                code = ended[0].returncode or 1 if any(p.returncode for p in ended) else 0
                import logging
                logging.getLogger(__name__).error("worker %d ended with code %s: stopping the other %d", ended[0].pid, ended[0].returncode, n - len(ended))
                break
            time.sleep(poll_s)
        else:
            code = 128 + stop["sig"]
    finally:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        deadline = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for sg, h in old.items():
            signal.signal(sg, h)
    return code


def main(argv=None) -> None:
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--streams", type=int, default=1024, help="client slots (= streams of the batched model) per worker process")
    ap.add_argument("--models", nargs="+", default=["alexa"])
    ap.add_argument("--weights", default=None, help='"synthetic" for random-init weights; default: the .onnx files next to the package')
    ap.add_argument("--threshold", type=float, default=0.5)
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=9000)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--use-mfma", type=int, default=None, choices=(3, 1),
                    help="kernel family (default: the fp16-split family 3, and the exact-fp32 family 1 for weights family 3 refuses at "
                         "commit); the fan-in server steps through oww_submit_masked, which both provide")
    ap.add_argument("--workers", type=int, default=1,
                    help="server processes sharing the port (SO_REUSEPORT; the kernel spreads new connections over them), each with its own "
                         "event loop and its own handle on the GPU: one Python event loop carries about two thousand real-time clients "
                         "(tools/serve_load.py), the GPU a thousand times that -- the edge scales over host cores, not over GPUs")
    ap.add_argument("--reuse-port", action="store_true", help="(set for the worker processes of --workers)")
    a = ap.parse_args(argv)
    if a.workers > 1:
        import sys
        cmd = [sys.executable, "-m", "openwakeword_amd.serve", "--streams", str(a.streams), "--models", *a.models, "--threshold", str(a.threshold),
               "--host", a.host, "--port", str(a.port), "--device", str(a.device), "--workers", "1", "--reuse-port"]
        cmd += ["--weights", a.weights] if a.weights else []
        cmd += ["--use-mfma", str(a.use_mfma)] if a.use_mfma is not None else []
        raise SystemExit(_supervise(cmd, a.workers))
    from .model import BatchedModel
    model = BatchedModel(a.streams, a.models, weights=a.weights, device=a.device, use_mfma=a.use_mfma)
    web.run_app(FanInServer(model, threshold=a.threshold).app(), host=a.host, port=a.port, reuse_port=True if a.reuse_port else None)


if __name__ == "__main__":
    main()
