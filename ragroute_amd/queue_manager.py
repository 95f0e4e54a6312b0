"""Query queueing for the services.

`QueryQueue` mirrors reference ragroute/queue_manager.py:4-32 (same methods).  `QueryBatcher` is what the GPU path adds
(SURVEY §8f rank 1): the reference router and data source handle strictly one query per loop iteration
(router.py:207-219, data_source.py:99-132, nq = 1 at data_source.py:114), which would leave the scan kernel — built
for 256 resident queries — 255/256 idle.  The batcher coalesces concurrent requests into windows of at most
`max_batch` queries (or whatever arrived within `max_wait_ms` of the first one), runs ONE batched call, and resolves
every request with its own row of the result, so each caller still gets a per-query reply."""
import asyncio
import collections
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

    One-stage form: run_batch(items: list) -> sequence of len(items) results (blocking; executed in a worker thread so the
    event loop keeps accepting requests while the GPU works).
    Two-stage form (search=, finish=): search(items) -> handle runs on the SEARCH thread, finish(handle) -> results on the
    REPLY thread, and the two overlap: while the reply thread builds window i's per-query tuples (pure Python), the search
    thread already runs window i+1's scan, so the GPU does not idle through the reply building.  At most one search and one
    finish are in flight; windows complete in order.
    Closed-loop callers (each client sends its next request when its reply is in) would defeat that overlap when all of them fit
    ONE window: the whole population then sits in a single search / reply cycle.  A window is therefore capped at half of the
    requests currently in flight, but not below `split_above` (128: below that a scan takes as long as a full one, so halving
    only adds passes): 256 clients run as two alternating windows of 128, one in the scan while the other's replies are built.
    An exception in either stage fails every request of that window with the same exception (the caller logs and drops, as
    data_source.py:137-138 does)."""

    def __init__(self, run_batch: Callable[[List[Any]], Sequence[Any]] = None, max_batch: int = 256, max_wait_ms: float = 2.0,
                 in_executor: bool = True, search: Callable[[List[Any]], Any] = None, finish: Callable[[Any], Sequence[Any]] = None,
                 split_above: int = 128):
        if max_batch < 1:
            raise ValueError("max_batch must be >= 1")
        if (run_batch is None) == (search is None) or (search is None) != (finish is None):
            raise ValueError("give run_batch, or search and finish")
        self.run_batch, self.search, self.finish = run_batch, search, finish
        self.max_batch = int(max_batch)
        self.max_wait = float(max_wait_ms) / 1e3
        self.in_executor = in_executor
        self._pending = collections.deque()   # (item, future) in arrival order
        self._waiter = None                   # the collector's wake-up future while the deque is empty
        self._worker = None
        self._pools = None
        self._inflight = set()
        self.split_above = int(split_above)
        self._open = 0                 # requests submitted and not yet resolved
        self.batches_run = 0
        self.items_run = 0
        self.search_seconds = 0.0      # time spent inside search / run_batch, and inside finish (worker threads' clocks)
        self.finish_seconds = 0.0

    def _ensure_started(self):
        if self._worker is None or self._worker.done():
            self._worker = asyncio.get_running_loop().create_task(self._run())

    def enqueue(self, item):
        """Enqueue one request; returns the future that will hold its own result (await it)."""
        if self._worker is None or self._worker.done():
            self._ensure_started()
        fut = asyncio.get_running_loop().create_future()
        self._open += 1
        fut.add_done_callback(self._closed)
        self._pending.append((item, fut))
        w = self._waiter
        if w is not None and not w.done():
            w.set_result(None)
        return fut

    def _closed(self, _fut):
        self._open -= 1

    async def submit(self, item):
        """Enqueue one request and wait for its own result."""
        return await self.enqueue(item)

    async def _next(self, loop, timeout=None):
        """Wait until a request is pending (or `timeout` seconds passed); True if one is."""
        if self._pending:
            return True
        self._waiter = loop.create_future()
        try:
            if timeout is None:
                await self._waiter
            else:
                await asyncio.wait_for(self._waiter, timeout)
        except asyncio.TimeoutError:
            pass
        finally:
            self._waiter = None
        return bool(self._pending)

    async def _collect(self, loop):
        """One window: the first request (waited for), then whatever is queued, then - up to max_wait after the first - whatever
        still arrives, at most max_batch."""
        await self._next(loop)
        item, fut = self._pending.popleft()
        items, futs = [item], [fut]
        deadline = loop.time() + self.max_wait
        q = self._pending
        cap = self.max_batch
        if self.search is not None and self._open > self.split_above:   # two-stage form: leave half of the population for the next window
            cap = min(cap, max(self.split_above, (self._open + 1) // 2))
        while len(items) < cap:
            if not q:
                timeout = deadline - loop.time()
                if timeout <= 0 or not await self._next(loop, timeout):
                    break
            nxt = q.popleft()
            items.append(nxt[0])
            futs.append(nxt[1])
        return items, futs

    def _timed(self, fn, which):
        import time

        def call(arg):
            t0 = time.perf_counter()
            try:
                return fn(arg)
            finally:
                setattr(self, which, getattr(self, which) + time.perf_counter() - t0)
        return call

    @staticmethod
    def _resolve(futs, results=None, error=None):
        for i, f in enumerate(futs):
            if not f.done():
                if error is None:
                    f.set_result(results[i])
                else:
                    f.set_exception(error)

    async def _run(self):
        loop = asyncio.get_running_loop()
        if self.search is not None:
            return await self._run_pipelined(loop)
        run = self._timed(self.run_batch, "search_seconds")
        while True:
            items, futs = await self._collect(loop)
            try:
                if self.in_executor:
                    results = await loop.run_in_executor(None, run, items)
                else:
                    results = run(items)
                if len(results) != len(items):
                    raise RuntimeError(f"run_batch returned {len(results)} results for {len(items)} requests")
                self._resolve(futs, results)
            except Exception as e:  # noqa: BLE001 - every request of the window sees the failure
                self._resolve(futs, error=e)
            self.batches_run += 1
            self.items_run += len(items)

    async def _run_pipelined(self, loop):
        from concurrent.futures import ThreadPoolExecutor
        if self._pools is None:
            self._pools = (ThreadPoolExecutor(1, thread_name_prefix="rr-search"), ThreadPoolExecutor(1, thread_name_prefix="rr-reply"))
        search_pool, finish_pool = self._pools
        search, finish = self._timed(self.search, "search_seconds"), self._timed(self.finish, "finish_seconds")
        prev = None                     # the previous window's finish stage (windows complete in order)

        async def finish_stage(handle, items, futs, after):
            try:
                if after is not None:
                    await asyncio.shield(after)
                results = await loop.run_in_executor(finish_pool, finish, handle)
                if len(results) != len(items):
                    raise RuntimeError(f"finish returned {len(results)} results for {len(items)} requests")
                self._resolve(futs, results)
            except Exception as e:  # noqa: BLE001
                self._resolve(futs, error=e)

        while True:
            items, futs = await self._collect(loop)
            self.batches_run += 1
            self.items_run += len(items)
            try:
                handle = await loop.run_in_executor(search_pool, search, items)
            except Exception as e:  # noqa: BLE001
                self._resolve(futs, error=e)
                continue
            prev = loop.create_task(finish_stage(handle, items, futs, prev))   # ... and straight on to the next window's search
            self._inflight.add(prev)
            prev.add_done_callback(self._inflight.discard)

    def shutdown_threads(self):
        """End the search / reply worker threads (two-stage form); safe from any thread, idempotent.  Requests still in flight are
        failed by the cancellation of the service loop that owns them."""
        pools, self._pools = self._pools, None
        if pools is not None:
            for p in pools:
                p.shutdown(wait=False)

    async def close(self):
        if self._worker is not None:
            self._worker.cancel()
            try:
                await self._worker
            except asyncio.CancelledError:
                pass
            self._worker = None
        for t in list(self._inflight):
            t.cancel()
        self.shutdown_threads()
