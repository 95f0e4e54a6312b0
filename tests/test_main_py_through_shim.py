"""The reference's UNMODIFIED `main.py` (BASELINE config 0's plumbing) started with the compat shim on PYTHONPATH: its orchestrator
(ragroute/ragroute.py:43-54) spawns `run_router` / `run_data_source` from ragroute_amd, its own aiohttp front-end
(ragroute/http_server.py) talks to them over the wire format, and a `GET /query` comes back with the merged documents.
`--simulate` (main.py:17) needs no models, indexes or GPU — and since round 3 neither does the front-end's merge: the list-shaped
`rerank_medrag` is a host function as in the reference, so the query COMPLETES in the build container.  pyzmq / ollama / python-liquid are not in this image: transport is
a TCP stand-in (tests/stubs_tcp/zmq), the two generation-side imports are inert stubs.  Runs only where the reference checkout
is mounted (never on the GPU box)."""
import json
import os
import signal
import socket
import subprocess
import sys
import time
import urllib.parse
import urllib.request

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RAGROUTE_REFERENCE_DIR", "/root/reference")
PORTS = [8000, 5555, 5556] + [6000 + i for i in range(4)] + [7500 + i for i in range(4)]   # ragroute/config.py:3-10


def _port_free(p):
    with socket.socket() as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)   # a closed listener in TIME_WAIT is not "in use"
        try:
            s.bind(("127.0.0.1", p))
            return True
        except OSError:
            return False


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "main.py")), reason="reference checkout not mounted")
def test_reference_main_py_runs_unchanged_with_the_drop_ins(tmp_path):
    pytest.importorskip("aiohttp")
    if not all(_port_free(p) for p in PORTS):
        pytest.skip("the reference's fixed ports are in use")
    env = dict(os.environ, RAGROUTE_REFERENCE_DIR=REF, PYTHONUNBUFFERED="1", PYTHONPATH=os.path.join(ROOT, "tests", "stubs_tcp"))
    log = open(tmp_path / "main.log", "w")
    proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "compat", "run_main.py"), "--dataset", "medrag", "--routing", "all",
                             "--disable-llm", "--simulate"], cwd=str(tmp_path), env=env, stdout=log, stderr=subprocess.STDOUT,
                            start_new_session=True)
    try:
        url = "http://127.0.0.1:8000/query?" + urllib.parse.urlencode({"q": "what is aspirin?", "choices": json.dumps(["a", "b"]), "qid": "q1"})
        body, deadline = None, time.time() + 90
        while time.time() < deadline and proc.poll() is None:
            try:
                with urllib.request.urlopen(url, timeout=60) as r:
                    body = json.loads(r.read())
                break
            except OSError as e:
                refused = isinstance(e, ConnectionError) or isinstance(getattr(e, "reason", None), ConnectionError)
                if refused:                       # the front-end is not up yet
                    time.sleep(0.5)
                    continue
                break                             # timed out
        logtxt = open(tmp_path / "main.log").read()
        # the reference's front-end gathered one reply per data source, each produced by ragroute_amd.data_source (http_server.py:233-257)
        for name in ("pubmed", "statpearls", "textbooks", "wikipedia"):
            assert f"Received results from data source {name}" in logtxt, logtxt[-3000:]
        # ... and its _complete_query (267-341) merged them with ragroute_amd.rerank.rerank_medrag (a host function, as in the
        # reference: no HIP device in this process) and answered
        assert body is not None and "Document 1 content" in json.dumps(body), (body, logtxt[-3000:])
        assert "No HIP GPUs are available" not in logtxt
    finally:
        if proc.poll() is None:
            os.killpg(proc.pid, signal.SIGINT)
            try:
                proc.wait(20)
            except subprocess.TimeoutExpired:
                os.killpg(proc.pid, signal.SIGKILL)
        log.close()
