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
