#!/usr/bin/env python3
"""The serving edge under REAL sockets at scale: one FanInServer (aiohttp, 127.0.0.1) in this process, N websocket
clients in separate client processes speaking the reference example's wire protocol (examples/web/streaming_server.py:32-70: the
sample rate as TEXT, int16 PCM as BINARY, {"activations": [...]} back).  Clients have mixed message periods (40 / 80 / 160 ms of
audio per message), random phases and a 50 % duty cycle (talk spurts and pauses of 0.8 .. 2.4 s), paced in real time.

Reported next to bench.py's simulated `masked_step.packed` numbers:
  * the pump: batched steps, participants per step, submit -> scores-delivered time per step, pump period;
  * per stream-step latency on the server (arrival of the message that completed the chunk -> scores dispatched), p50 / p99;
  * end-to-end latency at the clients (message sent -> activation message received; threshold 0, so every stream-step answers);
  * whether the event loop kept up (backlog at the end, messages dropped);
  * bit-exactness: the scores tapped for a sample of clients equal a private single-stream engine run on the audio they sent.

    python tools/serve_load.py --clients 8192 --seconds 12 --procs 8 [--out profiles/r05_serve_load.json]"""
import argparse
import asyncio
import json
import multiprocessing as mp
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

HEADS = ["alexa", "hey_mycroft", "hey_jarvis"]
POOL_SEED, POOL_LEN = 777, 1 << 20
PERIODS = [0.040, 0.080, 0.160]


def noise_pool():
    return (np.random.default_rng(POOL_SEED).standard_normal(POOL_LEN) * 3000).astype(np.int16)


def client_plan(g: int, seconds: float):
    """Deterministic schedule of client g: (period, list of message send offsets in s).  Talk spurts and pauses alternate."""
    r = np.random.default_rng(10_000 + g)
    period = PERIODS[g % len(PERIODS)]
    t = float(r.random() * period)                       # phase
    talking = bool(r.random() < 0.5)
    until = t + float(r.uniform(0.8, 2.4))
    sends = []
    while t < seconds:
        if t >= until:
            talking = not talking
            until = t + float(r.uniform(0.8, 2.4))
        if talking:
            sends.append(t)
        t += period
    return period, sends


def client_audio(pool, g: int, n_msgs: int, n_per: int):
    off = (g * 7919) % (POOL_LEN - n_msgs * n_per - 4)
    x = pool[off:off + n_msgs * n_per].copy()
    x[0], x[1] = (g & 0x7FFF), (g >> 15)                 # the first two samples name the client (the server maps connection -> client)
    return x


async def one_client(session, url, g, pool, seconds, t0, res):
    import aiohttp
    period, sends = client_plan(g, seconds)
    n_per = int(round(period * 16000))
    audio = client_audio(pool, g, len(sends), n_per)
    sent_at = []                                          # completion time of every chunk this client has delivered
    lat = []
    try:
        async with session.ws_connect(url, max_msg_size=0, heartbeat=None) as ws:
            await ws.receive()                            # {"loaded_models": ...}
            await ws.send_str("16000")

            async def reader():
                k = 0
                async for msg in ws:
                    if msg.type == aiohttp.WSMsgType.TEXT:
                        now = time.perf_counter()
                        if k < len(sent_at):
                            lat.append(now - sent_at[k])
                        k += 1
                    elif msg.type in (aiohttp.WSMsgType.CLOSED, aiohttp.WSMsgType.ERROR):
                        break
            rd = asyncio.ensure_future(reader())
            total = 0
            late = 0.0
            for i, off in enumerate(sends):
                delay = t0 + off - time.perf_counter()
                if delay > 0:
                    await asyncio.sleep(delay)
                else:
                    late = max(late, -delay)
                await ws.send_bytes(audio[i * n_per:(i + 1) * n_per].tobytes())
                now = time.perf_counter()
                done = (total + n_per) // 1280 - total // 1280
                total += n_per
                sent_at.extend([now] * done)
            await asyncio.sleep(max(0.0, t0 + seconds + 1.5 - time.perf_counter()))      # let the last answers arrive
            rd.cancel()
            res["lat"].extend(lat)
            res["chunks"] += len(sent_at)
            res["answers"] += len(lat)
            res["late"] = max(res["late"], late)
    except Exception as e:                                # noqa: BLE001
        res["errors"].append(f"client {g}: {type(e).__name__}: {e}")


def client_proc(ids, port, seconds, t0_wall, q, periods):
    PERIODS[:] = periods
    import aiohttp
    try:
        soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
        resource.setrlimit(resource.RLIMIT_NOFILE, (hard, hard))
    except Exception:
        pass
    pool = noise_pool()
    res = {"lat": [], "chunks": 0, "answers": 0, "late": 0.0, "errors": []}

    async def main():
        conn = aiohttp.TCPConnector(limit=0)
        async with aiohttp.ClientSession(connector=conn) as session:
            t0 = time.perf_counter() + (t0_wall - time.time())       # common start on the wall clock
            await asyncio.gather(*[one_client(session, f"http://127.0.0.1:{port}/ws", g, pool, seconds, t0, res) for g in ids])
    asyncio.run(main())
    q.put({"lat": np.asarray(res["lat"], np.float32), "chunks": res["chunks"], "answers": res["answers"], "late": res["late"],
           "errors": res["errors"][:5], "n_errors": len(res["errors"])})


class _FakeEngine:
    has_vad = False

    def __init__(self, S):
        self.S, self.q = S, []

    def pinned_empty(self, shape, dtype=np.int16):
        return np.zeros(shape, dtype)

    def submit(self, pcm, on):
        self.q.append(np.tile((np.abs(pcm[:, :8].astype(np.float32)).sum(1, keepdims=True) % 97) / 97.0, (1, 3)))

    def collect(self):
        return self.q.pop(0)


class _FakeModel:
    def __init__(self, S):
        self.n_streams, self.labels, self._keep, self.engine = S, list(HEADS), [0, 1, 2], _FakeEngine(S)

    def reset(self, ids=None, reset_vad=False):
        pass

    def close(self):
        pass


def pct(a, q):
    return float(np.percentile(a, q)) if len(a) else None


def serve_until(a, slots, t_end_wall, reuse_port):
    """Run one FanInServer in this process until `t_end_wall` (or until `a._done` is set by the in-process driver); returns its metrics."""
    import socket
    from aiohttp import web
    from openwakeword_amd import serve, weights as W
    from openwakeword_amd.engine import StreamEngine
    from openwakeword_amd.model import BatchedModel
    heads = {n: W.synthetic_head(n, seed=11 + i) for i, n in enumerate(HEADS)}
    emb = W.synthetic_embedding(seed=3)
    model = _FakeModel(slots) if a.dry_run else BatchedModel(slots, HEADS, weights={"heads": heads, "embedding": emb})
    step = max(1, a.clients // a.sample)
    tap, who = {}, {}                                     # connection id -> rows; connection id -> client number
    orig_push = serve._Client.push

    def push(self, x, now=0.0):                           # (tool-level hook: learn which client a connection is from its first samples)
        if self.cid not in who and x.size >= 2:
            who[self.cid] = int(x[0]) | (int(x[1]) << 15)
        return orig_push(self, x, now)
    serve._Client.push = push

    def on_scores(cid, k, row):
        g = who.get(cid)
        if g is not None and g % step == 0:
            tap.setdefault(cid, []).append(row.copy())
    srv = serve.FanInServer(model, threshold=a.threshold, window_s=a.window_ms * 1e-3, on_scores=on_scores)
    srv.keep_metrics = True

    async def run():
        runner = web.AppRunner(srv.app())
        await runner.setup()
        site = web.TCPSite(runner, "127.0.0.1", a.port, backlog=65535, reuse_port=reuse_port or None)
        await site.start()
        if a.ready_file:
            open(a.ready_file, "w").write("ready\n")
        a._ready = True
        while time.time() < t_end_wall and not getattr(a, "_done", False) and not (a.stop_file and os.path.exists(a.stop_file)):
            await asyncio.sleep(0.25)
        backlog = sum(c.n_pending // 1280 for c in srv.conns.values())
        n_conn = srv._next_cid
        await runner.cleanup()
        return backlog, n_conn
    backlog, n_conn = asyncio.run(run())
    lat = np.concatenate(srv.latencies_s) if srv.latencies_s else np.zeros(0)
    steps = np.asarray(srv.step_log, np.float64).reshape(-1, 4)
    pool = noise_pool()
    checked, worst, equal = 0, 0.0, True
    for cid, rows in ({} if a.dry_run else tap).items():
        g = who[cid]
        period, sends = client_plan(g, a.seconds)
        n_per = int(round(period * 16000))
        x = client_audio(pool, g, len(sends), n_per)
        e = StreamEngine(1, heads, emb)
        want = np.stack([e.step(x[None, k * 1280:(k + 1) * 1280])[0] for k in range(min(len(rows), x.size // 1280))]) if rows else np.zeros((0, 3))
        e.close()
        got = np.stack(rows)[:len(want)] if rows else want
        checked += 1
        if len(want):
            worst = max(worst, float(np.abs(got - want[:, model._keep]).max()))
            equal = equal and bool(np.array_equal(got, want[:, model._keep]))
    model.close()
    return {"lat": lat.astype(np.float32), "steps": steps, "n_steps": int(srv.n_steps), "n_stream_steps": int(srv.n_stream_steps),
            "backlog": int(backlog), "n_conn": int(n_conn), "dropped": int(srv.n_dropped_messages), "recoveries": int(srv.n_range_recoveries),
            "checked": checked, "equal": equal, "worst": worst, "slots": slots}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clients", type=int, default=8192)
    ap.add_argument("--slots", type=int, default=0, help="stream slots per server process (default: clients, or 1.5 x its share + 64)")
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--procs", type=int, default=8, help="client processes")
    ap.add_argument("--server-procs", type=int, default=1, help="server processes sharing the port (SO_REUSEPORT), each with its own handle on "
                    "the GPU: how `python -m openwakeword_amd.serve --workers N` scales the Python edge over host cores")
    ap.add_argument("--port", type=int, default=0)
    ap.add_argument("--sample", type=int, default=24, help="clients whose tapped scores are held to a private engine")
    ap.add_argument("--window-ms", type=float, default=10.0)
    ap.add_argument("--out", default="")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: a stand-in model (development aid for the socket plumbing of this tool)")
    ap.add_argument("--periods-ms", default="40,80,160", help="audio per client message (clients cycle through the list)")
    ap.add_argument("--threshold", type=float, default=0.0, help="activation threshold; 0 = every stream-step answers (needed for the "
                    "client-side latency: the k-th answer belongs to the k-th chunk)")
    ap.add_argument("--role", default="driver", choices=("driver", "server"), help="(internal) server = one worker process of --server-procs")
    ap.add_argument("--t-end", type=float, default=0.0, help="(internal) wall-clock time at which a server worker stops")
    ap.add_argument("--ready-file", default="")
    ap.add_argument("--stop-file", default="", help="(internal) a server worker stops when this file appears")
    ap.add_argument("--metrics-out", default="")
    a = ap.parse_args()
    PERIODS[:] = [float(v) * 1e-3 for v in a.periods_ms.split(",")]
    try:
        soft, hard = resource.getrlimit(resource.RLIMIT_NOFILE)
        resource.setrlimit(resource.RLIMIT_NOFILE, (hard, hard))
    except Exception:
        hard = -1
    if a.role == "server":
        m = serve_until(a, a.slots, a.t_end, True)
        np.savez(a.metrics_out, **{k: np.asarray(v) for k, v in m.items()})
        return
    import socket
    import subprocess
    import tempfile
    import threading
    if not a.port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            a.port = sk.getsockname()[1]
    n_srv = max(1, a.server_procs)
    slots = a.slots or (a.clients if n_srv == 1 else int(1.5 * a.clients / n_srv) + 64)
    t_start = time.time()
    t_give_up = t_start + 900.0
    tmp = tempfile.mkdtemp(prefix="serve_load_")
    a.stop_file = os.path.join(tmp, "stop")
    workers, metrics = [], []
    if n_srv > 1:
        for w in range(n_srv):
            cmd = [sys.executable, os.path.abspath(__file__), "--role", "server", "--clients", str(a.clients), "--slots", str(slots),
                   "--seconds", str(a.seconds), "--port", str(a.port), "--sample", str(a.sample), "--window-ms", str(a.window_ms),
                   "--periods-ms", a.periods_ms, "--threshold", str(a.threshold), "--t-end", str(t_give_up), "--stop-file", a.stop_file,
                   "--ready-file", os.path.join(tmp, f"ready{w}"), "--metrics-out", os.path.join(tmp, f"m{w}.npz")] + (["--dry-run"] if a.dry_run else [])
            workers.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=open(os.path.join(tmp, f"w{w}.err"), "w")))
        while not all(os.path.exists(os.path.join(tmp, f"ready{w}")) for w in range(n_srv)):
            if any(p.poll() is not None for p in workers) or time.time() > t_start + 300:
                errs = "".join(open(os.path.join(tmp, f"w{w}.err")).read()[-600:] for w in range(n_srv))
                raise SystemExit(f"server workers did not come up in time:\n{errs}")
            time.sleep(0.2)
    else:
        a.ready_file = ""
        box = {}
        th = threading.Thread(target=lambda: box.update(m=serve_until(a, slots, t_give_up, False)), daemon=True)
        th.start()
        while not getattr(a, "_ready", False):          # (the first `import torch` of a fresh box and the handle's creation take their time)
            if not th.is_alive() or time.time() > t_start + 300:
                raise SystemExit("the server did not come up")
            time.sleep(0.2)
    t0_wall = time.time() + 6.0 + a.clients / 3000.0      # the clients connect first and start sending together
    t_end_wall = t0_wall + a.seconds + 3.0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ids = [list(range(p, a.clients, a.procs)) for p in range(a.procs)]
    procs = [ctx.Process(target=client_proc, args=(ids[p], a.port, a.seconds, t0_wall, q, list(PERIODS)), daemon=True) for p in range(a.procs)]
    for p in procs:
        p.start()
    results = []
    while len(results) < a.procs and time.time() < t_end_wall + 60:
        try:
            results.append(q.get(True, 1.0))
        except Exception:
            pass
    time.sleep(max(0.0, t_end_wall - time.time()))
    open(a.stop_file, "w").write("stop\n")
    if n_srv > 1:
        for w, p in enumerate(workers):
            p.wait(timeout=300)
            z = np.load(os.path.join(tmp, f"m{w}.npz"), allow_pickle=True)
            metrics.append({k: z[k] for k in z.files})
    else:
        a._done = True
        th.join(timeout=300)
        metrics.append({k: np.asarray(v) for k, v in box["m"].items()})
    wall = time.time() - t_start
    lat_srv = np.concatenate([m["lat"] for m in metrics]).astype(np.float64)
    steps = np.concatenate([m["steps"].reshape(-1, 4) for m in metrics])
    done = steps[steps[:, 3] > 0]
    e2e = np.concatenate([r["lat"] for r in results]) if results else np.zeros(0, np.float32)
    n_stream_steps = int(sum(int(m["n_stream_steps"]) for m in metrics))
    period = np.concatenate([np.zeros(0)] + [np.diff(m["steps"].reshape(-1, 4)[:, 1]) for m in metrics if m["steps"].size > 4])
    out = {
        "what": "FanInServer under real websocket clients (tools/serve_load.py): reference wire protocol, 127.0.0.1, clients in separate processes",
        "clients": a.clients, "connections_accepted": int(sum(int(m["n_conn"]) for m in metrics)), "connections_per_server": [int(m["n_conn"]) for m in metrics],
        "server_processes": n_srv, "slots_per_server": slots,
        "client_processes": a.procs, "seconds": a.seconds,
        "message_periods_ms": [int(p * 1e3) for p in PERIODS], "duty": "talk spurts / pauses of 0.8-2.4 s, ~50 %", "window_ms": a.window_ms,
        "cpu_quota": __import__("oracle.parity_sample", fromlist=["effective_cpus"]).effective_cpus() if os.path.isdir(os.path.join(ROOT, "oracle")) else None,
        "nofile_limit": hard, "wall_s": round(wall, 1),
        "pump": {"batched_steps": int(sum(int(m["n_steps"]) for m in metrics)), "stream_steps": n_stream_steps,
                 "participants_per_step_mean": round(float(steps[:, 0].mean()), 1) if len(steps) else None,
                 "participation_mean": round(float(steps[:, 0].mean()) / slots, 4) if len(steps) else None,
                 "step_ms_p50": round(1e3 * pct(done[:, 3] - done[:, 1], 50), 3) if len(done) else None,
                 "step_ms_p99": round(1e3 * pct(done[:, 3] - done[:, 1], 99), 3) if len(done) else None,
                 "step_what": "submit -> scores dispatched, as the event loop sees it (one GPU-thread job per step while steps are short)",
                 "pump_period_ms_p50": round(1e3 * pct(period, 50), 3) if len(period) else None,
                 "stream_steps_per_s": round(n_stream_steps / a.seconds, 1)},
        "server_latency_ms": {"what": "arrival of the message that completed a chunk -> its scores dispatched", "n": int(lat_srv.size),
                              "p50": round(1e3 * pct(lat_srv, 50), 2) if lat_srv.size else None,
                              "p99": round(1e3 * pct(lat_srv, 99), 2) if lat_srv.size else None,
                              "max": round(1e3 * float(lat_srv.max()), 2) if lat_srv.size else None},
        "threshold": a.threshold,
        "client_latency_ms": ({"what": "message sent -> {'activations': ...} received (threshold 0: every stream-step answers)", "n": int(e2e.size),
                               "p50": round(1e3 * pct(e2e, 50), 2) if e2e.size else None, "p99": round(1e3 * pct(e2e, 99), 2) if e2e.size else None,
                               "max": round(1e3 * float(e2e.max()), 2) if e2e.size else None} if a.threshold <= 0 else
                              "not measured: with a threshold only some stream-steps answer, and the wire protocol does not say which"),
        "chunks_sent": int(sum(r["chunks"] for r in results)), "answers_received": int(sum(r["answers"] for r in results)),
        "client_send_lateness_max_ms": round(1e3 * max([r["late"] for r in results] or [0.0]), 1),
        "client_errors": int(sum(r["n_errors"] for r in results)), "client_error_samples": [e for r in results for e in r["errors"]][:5],
        "client_processes_reporting": len(results),
        "backlog_chunks_at_end": int(sum(int(m["backlog"]) for m in metrics)), "dropped_messages": int(sum(int(m["dropped"]) for m in metrics)),
        "range_recoveries": int(sum(int(m["recoveries"]) for m in metrics)),
        "bit_exact_sample": {"clients_checked": int(sum(int(m["checked"]) for m in metrics)), "equal": bool(all(bool(m["equal"]) for m in metrics)),
                             "max_abs_diff": float(max(float(m["worst"]) for m in metrics))},
    }
    line = json.dumps(out)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
