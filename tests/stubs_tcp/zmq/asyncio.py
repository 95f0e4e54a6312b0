import asyncio as _aio
import json
import struct

from . import PULL, PUSH


class Socket:
    def __init__(self, kind):
        self.kind, self.port = kind, None
        self._server = self._writer = None
        self._inbox = None
        self._closed = False
        self._readers = set()

    # PULL side ---------------------------------------------------------------------------------------------------------
    def bind(self, addr):
        assert self.kind == PULL and addr.startswith("tcp://*:"), addr
        self.port = int(addr.rsplit(":", 1)[1])
        try:                                    # like zmq, listen from bind() on (the callers bind inside a running loop)
            self._starting = _aio.get_running_loop().create_task(self._ensure_server())
        except RuntimeError:
            self._starting = None

    async def _ensure_server(self):
        if self._server is None and self._inbox is None:
            self._inbox = _aio.Queue()
            self._server = await _aio.start_server(self._serve, "127.0.0.1", self.port, reuse_address=True)
        while self._server is None:             # another task is starting it
            await _aio.sleep(0.01)

    async def _serve(self, reader, writer):
        task = _aio.current_task()
        self._readers.add(task)
        try:
            while True:
                head = await reader.readexactly(4)
                body = await reader.readexactly(struct.unpack("<I", head)[0])
                await self._inbox.put(body)
        except (_aio.IncompleteReadError, ConnectionError, _aio.CancelledError):
            pass
        finally:
            self._readers.discard(task)
            writer.close()

    async def recv(self):
        """One raw message (the front-end measures its size, http_server.py:233-234)."""
        assert self.kind == PULL
        await self._ensure_server()
        return await self._inbox.get()

    async def recv_json(self):
        return json.loads(await self.recv())

    # PUSH side ---------------------------------------------------------------------------------------------------------
    def connect(self, addr):
        assert self.kind == PUSH and addr.startswith("tcp://localhost:"), addr
        self.port = int(addr.rsplit(":", 1)[1])

    async def send_json(self, obj):
        assert self.kind == PUSH
        data = json.dumps(obj).encode()
        for _ in range(600):                      # like zmq: the peer may bind later
            if self._closed:
                return
            if self._writer is None:
                try:
                    _, self._writer = await _aio.open_connection("127.0.0.1", self.port)
                except OSError:
                    await _aio.sleep(0.05)
                    continue
            try:
                self._writer.write(struct.pack("<I", len(data)) + data)
                await self._writer.drain()
                return
            except (ConnectionError, OSError):
                self._writer = None
                await _aio.sleep(0.05)
        raise ConnectionError(f"no PULL peer on port {self.port}")

    def close(self):
        self._closed = True
        if self._writer is not None:
            self._writer.close()
        if self._server is not None:
            self._server.close()
        for t in list(self._readers):
            t.cancel()


class Context:
    def socket(self, kind):
        return Socket(kind)

    def term(self):
        pass
