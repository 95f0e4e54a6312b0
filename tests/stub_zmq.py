"""In-test stand-in for pyzmq (not installed in this image, and transport is out of scope): PUSH/PULL sockets over in-process
queues keyed by TCP port, JSON-encoding every message like the real send_json/recv_json so that only JSON-serialisable
replies pass.  Installed by `install()` as `zmq` and `zmq.asyncio` before the service loops import them."""
import asyncio
import json
import sys
import types

PULL, PUSH = 7, 8


class Hub:
    def __init__(self):
        self.queues = {}
        self.sockets = []
        self.contexts = []

    def queue(self, port):
        if port not in self.queues:
            self.queues[port] = asyncio.Queue()
        return self.queues[port]

    async def send(self, port, obj):
        """What http_server.py's PUSH socket to `port` does."""
        await self.queue(port).put(json.dumps(obj))

    async def recv(self, port, timeout=30.0):
        """What http_server.py's PULL socket bound at `port` receives."""
        return json.loads(await asyncio.wait_for(self.queue(port).get(), timeout))


HUB = Hub()


class Socket:
    def __init__(self, kind):
        self.kind, self.port, self.closed = kind, None, False
        HUB.sockets.append(self)

    def bind(self, addr):
        assert self.kind == PULL and addr.startswith("tcp://*:"), addr
        self.port = int(addr.rsplit(":", 1)[1])

    def connect(self, addr):
        assert self.kind == PUSH and addr.startswith("tcp://localhost:"), addr
        self.port = int(addr.rsplit(":", 1)[1])

    async def send_json(self, obj):
        assert self.kind == PUSH and not self.closed
        await HUB.queue(self.port).put(json.dumps(obj))

    async def recv_json(self):
        assert self.kind == PULL and not self.closed
        return json.loads(await HUB.queue(self.port).get())

    def close(self):
        self.closed = True


class Context:
    def __init__(self):
        self.terminated = False
        HUB.contexts.append(self)

    def socket(self, kind):
        return Socket(kind)

    def term(self):
        self.terminated = True


def install():
    """Fresh hub + stub modules in sys.modules; returns the hub."""
    global HUB
    HUB = Hub()
    zmq = types.ModuleType("zmq")
    zmq.PULL, zmq.PUSH = PULL, PUSH
    zmq_asyncio = types.ModuleType("zmq.asyncio")
    zmq_asyncio.Context = Context
    zmq.asyncio = zmq_asyncio
    sys.modules["zmq"] = zmq
    sys.modules["zmq.asyncio"] = zmq_asyncio
    return HUB


def uninstall():
    sys.modules.pop("zmq", None)
    sys.modules.pop("zmq.asyncio", None)
