"""Cross-process stand-in for pyzmq's PUSH/PULL sockets (pyzmq is not installed in this image and transport is out of scope):
length-prefixed JSON frames over loopback TCP.  Only what the reference's front-end and the service loops use is provided:
Context().socket(PUSH|PULL), bind("tcp://*:P"), connect("tcp://localhost:P"), send_json, recv_json, close, term.  Test
infrastructure: lets the reference's own `main.py` start its processes with the ragroute_amd drop-ins and talk to them."""
PULL, PUSH = 7, 8

from . import asyncio  # noqa: E402,F401  (zmq.asyncio is used as an attribute by the reference)
