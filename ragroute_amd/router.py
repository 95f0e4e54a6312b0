"""Router of the retrieval hot path — host-side mirror of reference ragroute/router.py.

Same names and argument meaning as the reference (`CorpusRoutingNN`, `Router.load_router`,
`Router.select_relevant_sources`, `run_router`), but the numeric part — feature build (router.py:245-267),
StandardScaler (269-270), MLP forward (50-55), sigmoid and threshold (275-280) — runs as ONE fused HIP
kernel (csrc/router.hip, C ABI `rr_router_mlp`) for a whole batch of queries.  The zero-padding,
centroid concat, one-hot and scaler are folded into fc1 on the host in float64 at load time:

    fc1(scale(x_c)) = (W1[:, :Dmax]/s_q) q  +  [ b1 + (W1[:, Dmax:2Dmax]/s_c) cen_c + (W1[:, 2Dmax:]/s_o) e_c - (W1/s) mean ]
                    =  W1q' q + c1[c]

Query encoders (router.py:85-104, 285-303) are out of scope: embeddings are inputs.
"""
import ctypes
import json
import logging
import os
import pickle
import random
import time
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, config
from ._lib import RouterWeightsStruct, check, lib

logger = logging.getLogger("router")


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


class CorpusRoutingNN:
    """Parameter container with the reference module's state_dict layout (router.py:37-48):
    fc1.weight[256,in] fc1.bias[256] ln1.weight/bias[256] fc2.weight[128,256] fc2.bias[128]
    ln2.weight/bias[128] fc3.weight[1,128] fc3.bias[1].  Dropout(0.4) is inert at inference (router.py:119)."""

    KEYS = ("fc1.weight", "fc1.bias", "ln1.weight", "ln1.bias", "fc2.weight", "fc2.bias", "ln2.weight", "ln2.bias",
            "fc3.weight", "fc3.bias")

    def __init__(self, input_dim: int, seed: Optional[int] = None):
        self.input_dim = int(input_dim)
        g = torch.Generator().manual_seed(0 if seed is None else seed)

        def linear(o, i):  # torch.nn.Linear default init: U(-1/sqrt(i), 1/sqrt(i))
            b = 1.0 / np.sqrt(i)
            return ((torch.rand((o, i), generator=g) * 2 - 1) * b).numpy(), ((torch.rand((o,), generator=g) * 2 - 1) * b).numpy()

        w1, b1 = linear(256, self.input_dim)
        w2, b2 = linear(128, 256)
        w3, b3 = linear(1, 128)
        self._sd = {"fc1.weight": w1, "fc1.bias": b1, "ln1.weight": np.ones(256, np.float32), "ln1.bias": np.zeros(256, np.float32),
                    "fc2.weight": w2, "fc2.bias": b2, "ln2.weight": np.ones(128, np.float32), "ln2.bias": np.zeros(128, np.float32),
                    "fc3.weight": w3, "fc3.bias": b3}
        self._plain = None

    def state_dict(self):
        return {k: v.copy() for k, v in self._sd.items()}

    def load_state_dict(self, sd):
        new = {}
        for k in self.KEYS:
            if k not in sd:
                raise KeyError(f"missing key {k!r} in router state_dict")
            v = sd[k]
            v = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if v.shape != self._sd[k].shape:
                raise ValueError(f"{k}: expected shape {self._sd[k].shape}, got {v.shape}")
            new[k] = v.astype(np.float32)
        self._sd = new
        self._plain = None

    def eval(self):
        return self

    def to(self, device):
        return self

    def forward(self, x):
        """Logits [R,1] of raw feature rows x [R,input_dim] (float32 tensor or array), on the GPU kernel
        with nothing folded (fc1 applied to the whole row)."""
        if self._plain is None:
            self._plain = FoldedRouter.plain(self._sd)
        xt = torch.as_tensor(np.asarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x).to("cuda", torch.float32)
        logits, _ = self._plain.run(xt.reshape(xt.shape[0], 1, self.input_dim))
        return logits.reshape(-1, 1)

    __call__ = forward


def fold_weights(sd, centroids, onehot_ids, n_onehot, d_max, scaler_mean=None, scaler_scale=None):
    """Fold zero-padding, centroid concat, one-hot and StandardScaler (router.py:245-270) into fc1, in float64.
    centroids [C][d_max] (already zero padded, router.py:149-151); onehot_ids [C].
    Returns (w1q [d_max][256], c1 [C][256]) with  fc1(scaler(features_c(q))) = q @ w1q + c1[c]."""
    W = np.asarray(sd["fc1.weight"], np.float64)
    n_in = 2 * d_max + n_onehot
    if W.shape != (256, n_in):
        raise ValueError(f"fc1.weight is {W.shape}, expected (256, {n_in})")
    scale = np.ones(n_in) if scaler_scale is None else np.asarray(scaler_scale, np.float64)
    mean = np.zeros(n_in) if scaler_mean is None else np.asarray(scaler_mean, np.float64)
    Ws = W / scale[None, :]
    base = np.asarray(sd["fc1.bias"], np.float64) - Ws @ mean
    cen = np.asarray(centroids, np.float64)
    c1 = base[None, :] + cen @ Ws[:, d_max:2 * d_max].T + Ws[:, 2 * d_max + np.asarray(onehot_ids)].T
    return Ws[:, :d_max].T.copy(), c1


class FoldedRouter:
    """Device-resident folded weights + launcher for `rr_router_mlp`."""

    def __init__(self, w1q, c1, sd, model_of_source, prob_threshold, device="cuda"):
        dev = torch.device(device)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
        self.n_sources, self.d_max = c1.shape[0], w1q.shape[0]
        self.n_models = int(max(model_of_source)) + 1
        self.device = dev
        self._ws = None   # fc1 partial sums of the batched (matrix-core) form, grown on demand
        self.t = {"w1q": f(w1q), "c1": f(c1), "ln1_g": f(sd["ln1.weight"]), "ln1_b": f(sd["ln1.bias"]),
                  "w2": f(sd["fc2.weight"].T), "b2": f(sd["fc2.bias"]), "ln2_g": f(sd["ln2.weight"]), "ln2_b": f(sd["ln2.bias"]),
                  "w3": f(sd["fc3.weight"].reshape(-1)),
                  "model_of_source": torch.tensor(list(model_of_source), dtype=torch.int32, device=dev)}
        self.struct = RouterWeightsStruct(
            n_sources=self.n_sources, d_max=self.d_max, n_models=self.n_models, reserved=0,
            model_of_source=self.t["model_of_source"].data_ptr(), w1q=self.t["w1q"].data_ptr(), c1=self.t["c1"].data_ptr(),
            ln1_g=self.t["ln1_g"].data_ptr(), ln1_b=self.t["ln1_b"].data_ptr(), w2=self.t["w2"].data_ptr(),
            b2=self.t["b2"].data_ptr(), ln2_g=self.t["ln2_g"].data_ptr(), ln2_b=self.t["ln2_b"].data_ptr(),
            w3=self.t["w3"].data_ptr(), b3=float(np.asarray(sd["fc3.bias"]).reshape(-1)[0]),
            prob_threshold=float(prob_threshold), ln_eps=1e-5, reserved2=0.0)

    @classmethod
    def plain(cls, sd, prob_threshold=0.5):
        """No folding: one 'source', the whole feature row is the 'query'."""
        return cls(np.asarray(sd["fc1.weight"], np.float64).T, np.asarray(sd["fc1.bias"], np.float64)[None, :], sd, [0], prob_threshold)

    @classmethod
    def fold(cls, sd, centroids, onehot_ids, n_onehot, d_max, model_of_source, prob_threshold, scaler_mean=None, scaler_scale=None,
             device="cuda"):
        w1q, c1 = fold_weights(sd, centroids, onehot_ids, n_onehot, d_max, scaler_mean, scaler_scale)
        return cls(w1q, c1, sd, model_of_source, prob_threshold, device)

    def run(self, xq):
        """xq: float32 CUDA tensor [nq, n_models, d_max] -> (logits f32 [nq,C], mask bool [nq,C]) on device."""
        if xq.dim() != 3 or xq.shape[1] != self.n_models or xq.shape[2] != self.d_max:
            raise ValueError(f"router input must be [nq,{self.n_models},{self.d_max}], got {tuple(xq.shape)}")
        xq = xq.to(self.device, torch.float32).contiguous()
        nq = xq.shape[0]
        logits = torch.empty((nq, self.n_sources), dtype=torch.float32, device=self.device)
        mask = torch.empty((nq, self.n_sources), dtype=torch.uint8, device=self.device)
        need = lib().rr_router_workspace_bytes(ctypes.byref(self.struct), nq)   # 0: small batch, latency-oriented kernel
        if need > (self._ws.numel() if self._ws is not None else 0):
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        check(lib().rr_router_mlp_ws(ctypes.byref(self.struct), xq.data_ptr(), nq, logits.data_ptr(), mask.data_ptr(),
                                     self._ws.data_ptr() if need else None, self._ws.numel() if need else 0, _stream_ptr()),
              "rr_router_mlp_ws")
        return logits, mask.view(torch.bool)


class Router:
    """Mirror of reference `Router` (router.py:58-346) for the selection path.

    `encode_query` needs a query encoder, which is out of scope: pass `encoder=`
    (callable: str -> Dict[model_name, np.ndarray]) or run with `simulate=True` (random embeddings,
    router.py:286-288)."""

    def __init__(self, dataset: str, data_sources: List[str], routing_strategy: str, simulate: bool = False, encoder=None):
        if dataset not in config.DATA_SOURCES:
            raise ValueError(f"Unknown dataset: {dataset}")
        self.dataset = dataset
        self.data_sources = list(data_sources)
        self.routing_strategy = routing_strategy
        self.simulate = simulate
        self.running = False
        self.encoder = encoder
        self.device = "cuda"
        self.model_names: List[str] = []
        for ds in self.data_sources:
            name = config.EMBEDDING_MODELS_PER_DATA_SOURCE[dataset][ds][0]
            if name not in self.model_names:
                self.model_names.append(name)
        self.embedding_models = {m: None for m in self.model_names}
        self.router: Optional[CorpusRoutingNN] = None
        self.scaler = None
        self.centroids: Dict[str, np.ndarray] = {}
        self._folded: Optional[FoldedRouter] = None
        self._batcher = None
        self._serve_task = None
        self.batch_window_ms = float(os.environ.get("RAGROUTE_BATCH_WINDOW_MS", 0.2))

    # -- loading (router.py:106-151) -------------------------------------------------------------
    def load_router(self, model_path=None, scaler_path=None, stats_files=None):
        ds = self.dataset
        M = config.MODELS_USR_DIR
        if model_path is None:
            model_path = {"medrag": os.path.join(M, "MedRAG/routing/best_model.pth"),
                          "feb4rag": os.path.join(M, "FeB4RAG/dataset_creation/2_search/router_best_model.pt"),
                          "wikipedia": os.path.join(M, "Retrieval-QA-Benchmark_backup", "euromlsys", "new_submission",
                                                    "cluster_router_output", "best_model.pth")}[ds]
        self.router = CorpusRoutingNN(config.ROUTER_INPUT_DIMENSION[ds])
        self.router.load_state_dict(torch.load(model_path, map_location="cpu"))
        self.scaler = None
        if ds == "medrag":
            path = scaler_path or os.path.join(M, "MedRAG/routing/preprocessed_data.pkl")
            with open(path, "rb") as f:
                _, _, _, self.scaler, _ = pickle.load(f)
        elif ds == "wikipedia":
            path = scaler_path or os.path.join(M, "Retrieval-QA-Benchmark_backup", "euromlsys", "new_submission",
                                               "cluster_router_output", "scaler.pkl")
            with open(path, "rb") as f:
                self.scaler = pickle.load(f)
        self.centroids = {}
        for corpus in self.data_sources:
            if stats_files is not None:
                stats_file = stats_files[corpus]
            elif ds == "medrag":
                stats_file = os.path.join(config.USR_DIR, "MedRAG/routing/", f"{corpus}_stats.json")
            elif ds == "feb4rag":
                stats_file = os.path.join(config.USR_DIR, "FeB4RAG/dataset_creation/2_search/embeddings",
                                          corpus + "_" + config.EMBEDDING_MODELS_PER_DATA_SOURCE[ds][corpus][0] + "_stats.json")
            else:
                stats_file = os.path.join(config.USR_DIR, "wiki_dataset", "dpr_wiki_index", "faiss_clusters", "cluster_stats.json")
            with open(stats_file, "r") as f:
                stats = json.load(f)
            if ds == "wikipedia" and isinstance(stats, list):
                stats = stats[int(corpus)]
            centroid = np.array(stats["centroid"], dtype=np.float32)
            self.centroids[corpus] = np.pad(centroid, (0, config.EMBEDDING_MAX_LENGTH[ds] - len(centroid)))
        self._folded = None  # folded onto the device lazily, at the first routing call

    def set_router(self, state_dict, centroids: Dict[str, np.ndarray], scaler_mean=None, scaler_scale=None):
        """Install weights directly (synthetic corpora, tests): same objects load_router() would build."""
        ds = self.dataset
        self.router = CorpusRoutingNN(config.ROUTER_INPUT_DIMENSION[ds])
        self.router.load_state_dict(state_dict)
        d_max = config.EMBEDDING_MAX_LENGTH[ds]
        self.centroids = {c: np.pad(np.asarray(v, np.float32), (0, d_max - len(v))) for c, v in centroids.items()}
        self.scaler = None if scaler_mean is None else _ScalerLike(scaler_mean, scaler_scale)
        self._folded = None

    def _fold(self):
        ds = self.dataset
        d_max = config.EMBEDDING_MAX_LENGTH[ds]
        if ds == "feb4rag":
            ids, n1 = [config.FEB4RAG_SOURCE_TO_ID[c] for c in self.data_sources], len(config.FEB4RAG_SOURCE_TO_ID)
        elif ds == "medrag":
            ids, n1 = [config.MEDRAG_SOURCE_TO_ID[c] for c in self.data_sources], len(config.MEDRAG_SOURCE_TO_ID)
        else:
            ids, n1 = [int(c) for c in self.data_sources], len(self.data_sources)
        mos = [self.model_names.index(config.EMBEDDING_MODELS_PER_DATA_SOURCE[ds][c][0]) for c in self.data_sources]
        mean = scale = None
        if self.scaler is not None:
            mean, scale = self.scaler.mean_, self.scaler.scale_
        self._folded = FoldedRouter.fold(self.router._sd, np.stack([self.centroids[c] for c in self.data_sources]), ids, n1, d_max,
                                         mos, config.ROUTER_THRESHOLD[ds], mean, scale)

    # -- selection (router.py:221-283) -------------------------------------------------------------
    def select_relevant_sources(self, query_embeddings: Dict[str, np.ndarray]) -> List[str]:
        if self.simulate:
            return self.data_sources
        if self.routing_strategy == "ragroute":
            return self.select_relevant_sources_ragroute(query_embeddings)
        if self.routing_strategy == "all":
            return self.data_sources
        if self.routing_strategy == "random":
            return random.sample(self.data_sources, config.RANDOM_ROUTING_SAMPLE[self.dataset])
        if self.routing_strategy == "none":
            return []
        raise ValueError(f"Unknown routing strategy: {self.routing_strategy}")

    def pack_queries(self, query_embeddings: Dict[str, np.ndarray]) -> torch.Tensor:
        """{model: [d_model] or [B,d_model]} -> float32 [B, n_models, d_max], zero padded (router.py:245-249)."""
        d_max = config.EMBEDDING_MAX_LENGTH[self.dataset]
        first = np.asarray(next(iter(query_embeddings.values())))
        B = 1 if first.ndim == 1 else first.shape[0]
        x = np.zeros((B, len(self.model_names), d_max), np.float32)
        for m, name in enumerate(self.model_names):
            e = np.asarray(query_embeddings[name], np.float32).reshape(B, -1)
            x[:, m, : e.shape[1]] = e
        return torch.from_numpy(x)

    def route_batch(self, xq):
        """xq: tensor [B, n_models, d_max] (host or device) -> (logits [B,C], mask bool [B,C]) device tensors."""
        if self._folded is None:
            if self.router is None:
                raise RuntimeError("router weights not loaded: call load_router() or set_router() first")
            self._fold()
        return self._folded.run(xq if isinstance(xq, torch.Tensor) else torch.as_tensor(xq))

    def select_relevant_sources_ragroute(self, query_embeddings: Dict[str, np.ndarray]) -> List[str]:
        _, mask = self.route_batch(self.pack_queries(query_embeddings))
        keep = mask[0].cpu().numpy()
        return [c for p, c in zip(keep, self.data_sources) if p]

    # -- encoding (out of scope) ---------------------------------------------------------------------
    def encode_query(self, query):
        if self.simulate:
            return {m: np.random.rand(config.EMBEDDING_MAX_LENGTH[self.dataset]) for m in self.embedding_models}
        if self.encoder is None:
            raise NotImplementedError("query encoders are outside the hot path; pass encoder= or use simulate=True")
        return self.encoder(query)

    # -- batched serving -------------------------------------------------------------
    def route_window(self, embeddings_list):
        """ONE fused-kernel launch for a window of requests: list of {model: embedding} -> list of source-name lists
        (what select_relevant_sources_ragroute returns for each of them on its own, router.py:241-283)."""
        d_max = config.EMBEDDING_MAX_LENGTH[self.dataset]
        x = np.zeros((len(embeddings_list), len(self.model_names), d_max), np.float32)
        for b, emb in enumerate(embeddings_list):
            for m, name in enumerate(self.model_names):
                e = np.asarray(emb[name], np.float32).reshape(-1)
                x[b, m, : e.shape[0]] = e
        _, mask = self.route_batch(torch.from_numpy(x))
        keep = mask.cpu().numpy()
        return [[c for p, c in zip(row, self.data_sources) if p] for row in keep]

    async def handle_query(self, query_data):
        """Reply message for one request, in the reference's wire format (router.py:305-332).  With the `ragroute`
        strategy concurrent requests are coalesced by the batcher into one router-MLP launch of up to 256 queries; the
        reference serves them strictly one at a time (router.py:207-219)."""
        import asyncio
        start_time = time.time()
        query_embeddings = self.encode_query(query_data["query"])
        embed_time = time.time() - start_time
        start_time = time.time()
        if self.simulate or self.routing_strategy != "ragroute":
            sources_corpora = self.select_relevant_sources(query_embeddings)
        else:
            if self._batcher is None:
                from .queue_manager import QueryBatcher
                self._batcher = QueryBatcher(self.route_window, max_batch=256, max_wait_ms=self.batch_window_ms)
            sources_corpora = await self._batcher.submit(query_embeddings)
        select_time = time.time() - start_time
        serialized = {m: (e.tolist() if isinstance(e, np.ndarray) else e) for m, e in query_embeddings.items()}
        if self.simulate:
            await asyncio.sleep(config.ROUTER_DELAY)  # router.py:321-322
        return {"query_id": query_data["id"], "data_sources": sources_corpora, "embeddings": serialized,
                "embedding_time": embed_time, "selection_time": select_time}

    # -- service loop (transport glue; needs pyzmq, which the reference also needs) -----------------
    async def start(self):
        """router.py:153-219: bind PULL :5555, connect PUSH :5556, load the weights, then serve.  Every request becomes its
        own task so that requests arriving together meet in the batcher."""
        import asyncio
        import zmq
        import zmq.asyncio
        self.context = zmq.asyncio.Context()
        self.running = True
        self.receiver = self.context.socket(zmq.PULL)
        self.receiver.bind(f"tcp://*:{config.SERVER_ROUTER_PORT}")
        self.sender = self.context.socket(zmq.PUSH)
        self.sender.connect(f"tcp://localhost:{config.ROUTER_SERVER_PORT}")
        self._serve_task = asyncio.current_task()
        pending, failure = set(), []
        try:
            if not self.simulate:
                if self.router is None:  # weights installed through set_router() are kept
                    self.load_router()
                if self.routing_strategy == "ragroute":  # warm-up forward, as router.py:173-175
                    self.route_batch(torch.zeros((1, len(self.model_names), config.EMBEDDING_MAX_LENGTH[self.dataset])))

            async def reply(query_data):
                await self.sender.send_json(await self.handle_query(query_data))

            def done(task):  # the reference lets a failing query end the router (router.py:213, no try/except): so do we
                pending.discard(task)
                if not task.cancelled() and task.exception() is not None and not failure:
                    failure.append(task.exception())
                    if self._serve_task is not None:
                        self._serve_task.cancel()

            while self.running:
                query_data = await self.receiver.recv_json()
                task = asyncio.ensure_future(reply(query_data))
                pending.add(task)
                task.add_done_callback(done)
        except asyncio.CancelledError:
            logger.info("Router shutdown requested")
        finally:
            for task in list(pending):
                task.cancel()
            self.stop()
        if failure:
            raise failure[0]

    def stop(self):
        """router.py:335-341: stop serving, close both sockets, terminate the context."""
        self.running = False
        task, self._serve_task = self._serve_task, None
        for name in ("receiver", "sender"):
            sock = getattr(self, name, None)
            if sock is not None:
                sock.close()
                setattr(self, name, None)
        ctx = getattr(self, "context", None)
        if ctx is not None:
            ctx.term()
            self.context = None
        if task is not None and not task.done():
            import asyncio
            try:
                current = asyncio.current_task()
            except RuntimeError:
                current = None
            if task is not current:
                task.cancel()  # the loop is parked in recv_json: wake it so it can leave


class _ScalerLike:
    def __init__(self, mean, scale):
        self.mean_ = np.asarray(mean, np.float64)
        self.scale_ = np.asarray(scale, np.float64)


_ENCODER_FACTORY = None
CURRENT = None   # the Router this process serves with (set by run_router; introspection / tests)


def set_encoder_factory(factory):
    """Plug the query encoders in: factory(dataset) -> callable(str) -> {model_name: np.ndarray}.  The encoders themselves
    (router.py:85-104, 285-303: MedCPT, DPR, the FeB4RAG zoo) are outside the hot path; `run_router` keeps the reference's
    signature, so this module-level hook is how a deployment (or a test) supplies them."""
    global _ENCODER_FACTORY
    _ENCODER_FACTORY = factory


async def run_router(dataset: str, data_sources: List[str], routing_strategy: str, simulate: bool = False):
    """Process entry with the reference's signature (router.py:343-346)."""
    encoder = _ENCODER_FACTORY(dataset) if (_ENCODER_FACTORY is not None and not simulate) else None
    global CURRENT
    CURRENT = router = Router(dataset, data_sources, routing_strategy, simulate=simulate, encoder=encoder)
    await router.start()
