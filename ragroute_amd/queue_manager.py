"""Query queueing for the services.

`QueryQueue` mirrors reference ragroute/queue_manager.py:4-32 (same methods).  `QueryBatcher` is what the GPU path adds
(SURVEY §8f rank 1): the reference router and data source handle strictly one query per loop iteration
(router.py:207-219, data_source.py:99-132, nq = 1 at data_source.py:114), which would leave the scan kernel — built
for 256 resident queries — 255/256 idle.  The batcher coalesces concurrent requests into windows of at most
`max_batch` queries (or whatever arrived within `max_wait_ms` of the first one), runs ONE batched call, and resolves
every request with its own row of the result, so each caller still gets a per-query reply."""
import asyncio
from typing import Any, Callable, List, Sequence


class QueryQueue:
    """Queue for managing incoming queries to the router (queue_manager.py:4-32)."""

    def __init__(self, max_size=100):
        self.queue = asyncio.Queue(maxsize=max_size)

    async def enqueue(self, query_data):
        await self.queue.put(query_data)

    async def dequeue(self):
        return await self.queue.get()

    def task_done(self):
        self.queue.task_done()

    async def join(self):
        await self.queue.join()

    def empty(self):
        return self.queue.empty()

    def qsize(self):
        return self.queue.qsize()


class QueryBatcher:
    """Coalesce awaitable single requests into batched calls.

    run_batch(items: list) -> sequence of len(items) results (blocking; executed in the default executor so the
    event loop keeps accepting requests while the GPU works).  An exception in run_batch fails every request of that
    window with the same exception (the caller logs and drops, as data_source.py:137-138 does)."""

    def __init__(self, run_batch: Callable[[List[Any]], Sequence[Any]], max_batch: int = 256, max_wait_ms: float = 2.0,
                 in_executor: bool = True):
        if max_batch < 1:
            raise ValueError("max_batch must be >= 1")
        self.run_batch = run_batch
        self.max_batch = int(max_batch)
        self.max_wait = float(max_wait_ms) / 1e3
        self.in_executor = in_executor
        self._queue: asyncio.Queue = None
        self._worker = None
        self.batches_run = 0
        self.items_run = 0

    def _ensure_started(self):
        if self._worker is None or self._worker.done():
            self._queue = self._queue or asyncio.Queue()
            self._worker = asyncio.get_running_loop().create_task(self._run())

    async def submit(self, item):
        """Enqueue one request and wait for its own result."""
        self._ensure_started()
        fut = asyncio.get_running_loop().create_future()
        await self._queue.put((item, fut))
        return await fut

    async def _run(self):
        loop = asyncio.get_running_loop()
        while True:
            item, fut = await self._queue.get()
            items, futs = [item], [fut]
            deadline = loop.time() + self.max_wait
            while len(items) < self.max_batch:
                timeout = deadline - loop.time()
                if timeout <= 0 and self._queue.empty():
                    break
                try:
                    nxt = self._queue.get_nowait() if timeout <= 0 else await asyncio.wait_for(self._queue.get(), timeout)
                except (asyncio.TimeoutError, asyncio.QueueEmpty):
                    break
                items.append(nxt[0])
                futs.append(nxt[1])
            try:
                if self.in_executor:
                    results = await loop.run_in_executor(None, self.run_batch, items)
                else:
                    results = self.run_batch(items)
                if len(results) != len(items):
                    raise RuntimeError(f"run_batch returned {len(results)} results for {len(items)} requests")
                for f, r in zip(futs, results):
                    if not f.done():
                        f.set_result(r)
            except Exception as e:  # noqa: BLE001 - every request of the window sees the failure
                for f in futs:
                    if not f.done():
                        f.set_exception(e)
            self.batches_run += 1
            self.items_run += len(items)

    async def close(self):
        if self._worker is not None:
            self._worker.cancel()
            try:
                await self._worker
            except asyncio.CancelledError:
                pass
            self._worker = None
