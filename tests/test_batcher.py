"""CPU: the query batcher and the QueryQueue mirror (reference queue_manager.py)."""
import asyncio
import time

import pytest

from ragroute_amd.queue_manager import QueryBatcher, QueryQueue


def test_query_queue_mirror():
    async def go():
        q = QueryQueue(max_size=3)
        assert q.empty() and q.qsize() == 0
        await q.enqueue({"id": 1})
        await q.enqueue({"id": 2})
        assert q.qsize() == 2 and not q.empty()
        assert (await q.dequeue())["id"] == 1
        q.task_done()
        assert (await q.dequeue())["id"] == 2
        q.task_done()
        await q.join()
    asyncio.run(go())


def test_batcher_coalesces_and_preserves_per_request_results():
    calls = []

    def run(items):
        calls.append(len(items))
        time.sleep(0.01)
        return [x * 10 for x in items]

    async def go():
        b = QueryBatcher(run, max_batch=256, max_wait_ms=20)
        res = await asyncio.gather(*[b.submit(i) for i in range(300)])
        await b.close()
        return res, b
    res, b = asyncio.run(go())
    assert res == [i * 10 for i in range(300)]
    assert max(calls) <= 256 and sum(calls) == 300
    assert len(calls) <= 3          # 300 concurrent requests -> 2 (or 3) windows, not 300 calls
    assert b.items_run == 300


def test_batcher_single_request_is_not_held_longer_than_max_wait():
    async def go():
        b = QueryBatcher(lambda items: items, max_batch=64, max_wait_ms=5)
        t0 = time.perf_counter()
        r = await b.submit("x")
        dt = time.perf_counter() - t0
        await b.close()
        return r, dt
    r, dt = asyncio.run(go())
    assert r == "x" and dt < 0.5


def test_batcher_failure_reaches_every_request_of_the_window_only():
    state = {"n": 0}

    def run(items):
        state["n"] += 1
        if state["n"] == 1:
            raise ValueError("boom")
        return items

    async def go():
        b = QueryBatcher(run, max_batch=4, max_wait_ms=50)
        first = await asyncio.gather(*[b.submit(i) for i in range(4)], return_exceptions=True)
        second = await asyncio.gather(*[b.submit(i) for i in range(3)])
        await b.close()
        return first, second
    first, second = asyncio.run(go())
    assert all(isinstance(e, ValueError) for e in first)
    assert second == [0, 1, 2]


def test_batcher_rejects_wrong_result_count():
    async def go():
        b = QueryBatcher(lambda items: items[:-1], max_batch=8, max_wait_ms=10)
        with pytest.raises(RuntimeError):
            await asyncio.gather(*[b.submit(i) for i in range(3)])
        await b.close()
    asyncio.run(go())


# ---- two-stage form: window i's reply building overlaps window i+1's search ------------------------------------------
def test_pipelined_batcher_overlaps_search_and_finish_and_keeps_results_per_request():
    log = []

    def search(items):
        log.append(("s0", time.perf_counter()))
        time.sleep(0.05)
        log.append(("s1", time.perf_counter()))
        return list(items)

    def finish(handle):
        log.append(("f0", time.perf_counter()))
        time.sleep(0.05)
        log.append(("f1", time.perf_counter()))
        return [x * 10 for x in handle]

    async def go():
        b = QueryBatcher(search=search, finish=finish, max_batch=4, max_wait_ms=1)
        res = await asyncio.gather(*[b.submit(i) for i in range(12)])
        await b.close()
        return res, b
    res, b = asyncio.run(go())
    assert res == [i * 10 for i in range(12)]
    assert b.batches_run == 3 and b.items_run == 12
    s0 = [t for k, t in log if k == "s0"]
    f1 = [t for k, t in log if k == "f1"]
    assert s0[1] < f1[0] and s0[2] < f1[1], "the next window's search must start before the previous window's replies are built"
    # one search and one finish at a time, windows complete in order
    spans = sorted((t, k) for k, t in log)
    assert [k for _, k in spans if k[0] == "s"] == ["s0", "s1"] * 3 and [k for _, k in spans if k[0] == "f"] == ["f0", "f1"] * 3
    assert b.search_seconds >= 0.14 and b.finish_seconds >= 0.14


def test_pipelined_batcher_failures_stay_in_their_window():
    state = {"s": 0, "f": 0}

    def search(items):
        state["s"] += 1
        if state["s"] == 1:
            raise ValueError("search failed")
        return list(items)

    def finish(handle):
        state["f"] += 1
        if state["f"] == 1:
            raise KeyError("finish failed")
        return handle

    async def go():
        b = QueryBatcher(search=search, finish=finish, max_batch=2, max_wait_ms=20)
        a = await asyncio.gather(*[b.submit(i) for i in range(2)], return_exceptions=True)
        c = await asyncio.gather(*[b.submit(i) for i in range(2)], return_exceptions=True)
        d = await asyncio.gather(*[b.submit(i) for i in range(2)])
        short = QueryBatcher(search=lambda items: items, finish=lambda h: h[:-1], max_batch=8, max_wait_ms=5)
        with pytest.raises(RuntimeError):
            await asyncio.gather(*[short.submit(i) for i in range(3)])
        await b.close()
        await short.close()
        return a, c, d
    a, c, d = asyncio.run(go())
    assert all(isinstance(e, ValueError) for e in a) and all(isinstance(e, KeyError) for e in c) and d == [0, 1]
    with pytest.raises(ValueError):
        QueryBatcher(search=lambda i: i)
    with pytest.raises(ValueError):
        QueryBatcher()


def test_pipelined_batcher_survives_cancelled_and_timed_out_requests():
    """Clients that give up (asyncio.wait_for timeouts, cancelled tasks) must not wedge the batcher, leak `_open` counts (the window
    cap is derived from them) or lose other requests' results."""
    import random

    def search(items):
        time.sleep(0.002)
        return list(items)

    def finish(handle):
        time.sleep(0.001)
        return [x + 1 for x in handle]

    async def go():
        b = QueryBatcher(search=search, finish=finish, max_batch=64, max_wait_ms=0.5)
        rnd = random.Random(3)
        ok = lost = 0

        async def client(i):
            nonlocal ok, lost
            try:
                r = await asyncio.wait_for(b.submit(i), timeout=rnd.choice([0.0005, 0.002, 0.01, 1.0]))
                assert r == i + 1
                ok += 1
            except asyncio.TimeoutError:
                lost += 1

        for wave in range(10):
            tasks = [asyncio.ensure_future(client(wave * 300 + i)) for i in range(300)]
            for t in rnd.sample(tasks, 20):
                t.cancel()
            await asyncio.gather(*tasks, return_exceptions=True)
        assert b._open == 0 and not b._pending
        assert await asyncio.wait_for(b.submit(41), timeout=2.0) == 42          # still serving
        await b.close()
        return ok, lost
    ok, lost = asyncio.run(go())
    assert ok > 500 and lost > 0
