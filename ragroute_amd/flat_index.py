"""faiss-shaped flat index whose corpus lives in HBM and whose search is the HIP scan kernel.

Mirrors the operator surface the reference uses from `faiss` (ragroute/data_source.py):
  index.search(xq: float32[nq,d], k) -> (D float32[nq,k], I int64[nq,k])      data_source.py:158,186,203
  index.ntotal, index.d
  normalize_L2(x: float32[n,d])  in place                                       data_source.py:198-199
PyTorch is used for device memory and streams only; all arithmetic is in libragroute_hip.so.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, lib

_TORCH_DTYPE = {"fp16": torch.float16, "bf16": torch.bfloat16}
_RR_DTYPE = {"fp16": _lib.RR_DTYPE_F16, "bf16": _lib.RR_DTYPE_BF16}


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _require_gpu():
    if not torch.cuda.is_available():
        raise _lib.RagrouteHipError("ragroute_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")


def normalize_L2(x):
    """In-place row-wise L2 normalisation (faiss.normalize_L2, data_source.py:198-199).
    Accepts a C-contiguous float32 numpy array [n,d] (normalised in place through the GPU) or a
    float32 CUDA tensor (normalised in place on device)."""
    if isinstance(x, torch.Tensor):
        if x.dtype != torch.float32 or not x.is_cuda or not x.is_contiguous() or x.dim() != 2:
            raise ValueError("normalize_L2 needs a contiguous float32 CUDA tensor [n,d]")
        check(lib().rr_l2_normalize_f32(x.data_ptr(), x.shape[0], x.shape[1], _stream_ptr()), "rr_l2_normalize_f32")
        return
    if not (isinstance(x, np.ndarray) and x.dtype == np.float32 and x.ndim == 2 and x.flags.c_contiguous):
        raise ValueError("normalize_L2 needs a C-contiguous float32 array [n,d]")
    _require_gpu()
    t = torch.from_numpy(x).cuda()
    check(lib().rr_l2_normalize_f32(t.data_ptr(), x.shape[0], x.shape[1], _stream_ptr()), "rr_l2_normalize_f32")
    x[...] = t.cpu().numpy()


class FlatIndex:
    """Exact inner-product index (the role of faiss.IndexFlatIP in the reference).

    metric: "ip" (reference behaviour), "cosine" (rows and queries L2-normalised on ingest/search, i.e. normalize_L2 + IP
    as data_source.py:196-203 does for the wikipedia corpora) or "l2" (squared L2 distances, nearest first: the role of
    faiss.IndexFlatL2).
    dtype : "fp16" or "bf16" storage/MFMA input type; scores accumulate in f32.

    build_screen() adds an int8 screening copy (ip / cosine, d <= 1536): searches then stream half the bytes and return the
    same exact top-k, proven per query by a quantisation-error bound; a batch whose proof fails is repeated on the
    f16/bf16 rows (counted in `screen_fallbacks`).
    """

    def __init__(self, d, metric="ip", dtype="fp16", device=None):
        if metric not in ("ip", "cosine", "l2"):
            raise ValueError(f"unknown metric {metric!r}")
        if dtype not in _TORCH_DTYPE:
            raise ValueError(f"unknown dtype {dtype!r}")
        self.d = int(d)
        self.metric = metric
        self.dtype = dtype
        self.dim = lib().rr_padded_dim(self.d)
        if self.dim < 0:
            raise _lib.RagrouteHipError(f"embedding dimension {d} is not supported (max 8192)")
        _require_gpu()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.ntotal = 0
        self._xb = torch.empty((0, self.dim), dtype=_TORCH_DTYPE[dtype], device=self.device)
        self._hn = None  # |x|^2/2 per row (metric "l2")
        self._ws = {}
        self._x8 = self._x8_stats = None   # int8 screening copy (build_screen)
        self._ws_screen = {}
        self.screen_list = 512
        self.screen_fallbacks = 0

    # -- storage -----------------------------------------------------------------------------
    def reserve(self, n):
        if n > self._xb.shape[0]:
            new = torch.empty((n, self.dim), dtype=self._xb.dtype, device=self.device)
            new[: self.ntotal].copy_(self._xb[: self.ntotal])
            self._xb = new

    def add(self, x, chunk_rows=1 << 18):
        """Append rows (float32 numpy [n,d] or CUDA tensor) to the HBM-resident corpus."""
        n = x.shape[0]
        if x.shape[1] != self.d:
            raise ValueError(f"expected rows of dimension {self.d}, got {x.shape[1]}")
        if self.ntotal + n > self._xb.shape[0]:
            self.reserve(max(self.ntotal + n, int(self._xb.shape[0] * 1.5)))
        with torch.cuda.device(self.device):
            for s in range(0, n, chunk_rows):
                e = min(n, s + chunk_rows)
                part = x[s:e]
                if isinstance(part, np.ndarray):  # also np.memmap: only this chunk is paged in
                    part = np.ascontiguousarray(part, dtype=np.float32)
                    part = torch.from_numpy(part if part.flags.writeable else part.copy())
                part = part.to(self.device, dtype=torch.float32).contiguous()
                out = self._xb[self.ntotal + s : self.ntotal + e]
                check(lib().rr_rows_to_half(part.data_ptr(), e - s, self.d, self.d, out.data_ptr(), _RR_DTYPE[self.dtype],
                                            self.dim, int(self.metric == "cosine"), _stream_ptr()), "rr_rows_to_half")
                del part
        self.ntotal += n
        self._x8 = self._x8_stats = None  # one scale for the whole corpus: the screening copy is rebuilt, not appended to
        self._refresh_norms()

    def _refresh_norms(self):
        if self.metric != "l2":
            return
        self._hn = torch.empty(max(1, self.ntotal), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().rr_half_sqnorms(self._xb.data_ptr(), _RR_DTYPE[self.dtype], self.ntotal, self.dim, self._hn.data_ptr(),
                                        _stream_ptr()), "rr_half_sqnorms")

    def adopt(self, xb_dev, ntotal=None):
        """Use an existing device matrix [n, dim] of the index dtype as the corpus (no copy)."""
        if xb_dev.dtype != self._xb.dtype or xb_dev.dim() != 2 or xb_dev.shape[1] != self.dim or not xb_dev.is_contiguous():
            raise ValueError("adopt() needs a contiguous device matrix [n, rr_padded_dim(d)] of the index dtype")
        self._xb = xb_dev
        self.ntotal = int(xb_dev.shape[0] if ntotal is None else ntotal)
        self._x8 = self._x8_stats = None
        self._refresh_norms()

    def build_screen(self, list_len=None):
        """Build the int8 screening copy of the current corpus (+50 % HBM).  list_len: rows per query that are re-scored
        exactly (k <= list_len <= 1024; default max(512, 16 k) capped at 1024)."""
        if self.metric == "l2":
            raise _lib.RagrouteHipError("the int8 screening copy serves the ip / cosine metrics only")
        dim8 = lib().rr_screen_dim(self.dim)
        if dim8 < 0:
            raise _lib.RagrouteHipError("the int8 screening copy needs d <= 1536")
        if list_len is not None:
            self.screen_list = int(list_len)
        self._x8 = torch.empty((max(1, self.ntotal), dim8), dtype=torch.int8, device=self.device)
        self._x8_stats = torch.zeros(8, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().rr_screen_build(self._xb.data_ptr(), _RR_DTYPE[self.dtype], self.ntotal, self.dim, self._x8.data_ptr(),
                                        self._x8_stats.data_ptr(), _stream_ptr()), "rr_screen_build")
        return self

    def search_screened(self, xq_half, k, id_offset=0, out=None, route_mask=None, list_len=None):
        """Device-to-device search through the int8 screening copy: (D, I, exact) CUDA tensors; exact[q] == 1 where the result
        is proven to be the exact top-k.  No host sync; search_prepared() wraps this with the check and the fallback."""
        if self._x8 is None:
            raise _lib.RagrouteHipError("build_screen() has not been called (or the corpus changed since)")
        nq = xq_half.shape[0]
        L = int(list_len if list_len is not None else min(_lib.RR_MAX_K, max(self.screen_list, 16 * k)))
        L = max(L, k)
        key = (k, L, nq)
        ws = self._ws_screen.get(key)
        if ws is None:
            nbytes = lib().rr_flat_search_screened_workspace_bytes(k, L, nq, self.dim)
            if nbytes == 0:
                raise ValueError(f"need 1 <= k <= list_len <= {_lib.RR_MAX_K}, got k={k}, list_len={L}")
            self._ws_screen = {key: torch.empty(nbytes, dtype=torch.uint8, device=self.device)}
            ws = self._ws_screen[key]
        if out is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        else:
            D, I = out
        exact = torch.empty(nq, dtype=torch.uint8, device=self.device)
        mptr, mstride = self._mask_args(route_mask, nq)
        check(lib().rr_flat_search_screened(self._xb.data_ptr(), _RR_DTYPE[self.dtype], self._x8.data_ptr(), self._x8_stats.data_ptr(),
                                            self.ntotal, self.dim, xq_half.data_ptr(), nq, k, L, D.data_ptr(), I.data_ptr(), id_offset,
                                            exact.data_ptr(), ws.data_ptr(), ws.numel(), mptr, mstride, _stream_ptr()),
              "rr_flat_search_screened")
        return D, I, exact

    @staticmethod
    def _mask_args(route_mask, nq):
        if route_mask is None:
            return None, 0
        if route_mask.dtype not in (torch.bool, torch.uint8) or route_mask.dim() != 1 or route_mask.shape[0] != nq or not route_mask.is_cuda:
            raise ValueError("route_mask must be a bool/uint8 CUDA vector with one entry per query")
        return route_mask.data_ptr(), route_mask.stride(0)

    def centroid(self):
        """float32 CUDA vector [d]: mean of the stored rows (the router's centroid feature, router.py:147-151)."""
        out = torch.empty(self.dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(lib().rr_centroid(self._xb.data_ptr(), _RR_DTYPE[self.dtype], self.ntotal, self.dim, self.d, out.data_ptr(),
                                    _stream_ptr()), "rr_centroid")
        return out[: self.d]

    @property
    def xb(self):
        return self._xb[: self.ntotal]

    # -- search ------------------------------------------------------------------------------
    def _workspace(self, k):
        ws = self._ws.get(k)
        if ws is None:
            nbytes = lib().rr_flat_search_workspace_bytes_for(k, self.dim)
            if nbytes == 0:
                raise ValueError(f"k must be in [1, {_lib.RR_MAX_K}], got {k}")
            self._ws = {k: torch.empty(nbytes, dtype=torch.uint8, device=self.device)}
            ws = self._ws[k]
        return ws

    def prepare_queries(self, xq):
        """float32 [nq,d] (numpy or tensor) -> device [nq,dim] in the index dtype (normalised for cosine)."""
        if isinstance(xq, np.ndarray):
            xq = torch.from_numpy(np.ascontiguousarray(xq, dtype=np.float32))
        if xq.dim() != 2 or xq.shape[1] != self.d:
            raise ValueError(f"queries must be [nq,{self.d}]")
        xq = xq.to(self.device, dtype=torch.float32).contiguous()
        out = torch.empty((xq.shape[0], self.dim), dtype=self._xb.dtype, device=self.device)
        check(lib().rr_rows_to_half(xq.data_ptr(), xq.shape[0], self.d, self.d, out.data_ptr(), _RR_DTYPE[self.dtype],
                                    self.dim, int(self.metric == "cosine"), _stream_ptr()), "rr_rows_to_half")
        return out

    def search_prepared(self, xq_half, k, id_offset=0, out=None, route_mask=None):
        """Device-to-device search: xq_half [nq,dim] index dtype -> (D f32[nq,k], I i64[nq,k]) CUDA tensors.
        route_mask: optional bool/uint8 CUDA tensor [nq] (may be a strided column of the router's [nq,C] mask):
        queries with a zero entry get an all-padding result.  Enqueued on the current stream, no host sync."""
        if xq_half.dtype != self._xb.dtype or xq_half.dim() != 2 or xq_half.shape[1] != self.dim or not xq_half.is_contiguous():
            raise ValueError("search_prepared() needs contiguous [nq, dim] queries of the index dtype")
        nq = xq_half.shape[0]
        if self._x8 is not None and nq > 0:
            # screened search; the proof flags are the one host read of this path
            D, I, exact = self.search_screened(xq_half, k, id_offset, out, route_mask)
            if bool(exact.all()):
                return D, I
            self.screen_fallbacks += 1
            out = (D, I)
        ws = self._workspace(k)
        if out is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        else:
            D, I = out
        mptr, mstride = self._mask_args(route_mask, nq)
        if self.metric == "l2":
            check(lib().rr_flat_search_l2(self._xb.data_ptr(), self._hn.data_ptr() if self._hn is not None else None,
                                          _RR_DTYPE[self.dtype], self.ntotal, self.dim, xq_half.data_ptr(), nq, k, D.data_ptr(),
                                          I.data_ptr(), id_offset, ws.data_ptr(), ws.numel(), mptr, mstride, _stream_ptr()),
                  "rr_flat_search_l2")
        else:
            check(lib().rr_flat_search(self._xb.data_ptr(), _RR_DTYPE[self.dtype], self.ntotal, self.dim, xq_half.data_ptr(), nq, k,
                                       D.data_ptr(), I.data_ptr(), id_offset, ws.data_ptr(), ws.numel(), mptr, mstride, _stream_ptr()),
                  "rr_flat_search")
        return D, I

    def search(self, xq, k):
        """faiss call shape: float32 [nq,d] -> (D float32[nq,k], I int64[nq,k]) numpy arrays."""
        with torch.cuda.device(self.device):
            D, I = self.search_prepared(self.prepare_queries(xq), int(k))
            return D.cpu().numpy(), I.cpu().numpy()


class SegmentedIndex:
    """Several data sources that are searched with the SAME query embeddings, resident in ONE HBM matrix and searched in ONE
    pass (C ABI `rr_flat_search_segments`): MedRAG's four sources all use MedCPT, five FeB4RAG sources UAE-Large-V1
    (reference ragroute/config.py:37-71).  The reference asks every selected source separately (`index.search`,
    data_source.py:158, 186, 203) and its front-end concatenates the replies and keeps the k best (http_server.py:280-293,
    rerank.py:3-9); `search_prepared` returns exactly that merged list — per query, the k best rows of the union of the
    sources the router selected for it — with one query preparation, one bootstrap and one chunk schedule for all of them.

    Source s occupies rows [begin_s, begin_s + n_s) of the matrix, begin_s a multiple of 256; `source(s)` is a FlatIndex on that
    slice (no copy), so the per-source call surface (`DataSource.retrieve_docs_*`) keeps working on the same memory.
    Result ids are id_offsets[s] + local row; ties are broken by (source order, row) = ascending id when the id offsets ascend.
    Metrics: "ip" / "cosine"."""

    def __init__(self, d, rows_per_source, id_offsets=None, mask_cols=None, metric="ip", dtype="fp16", device=None):
        if metric not in ("ip", "cosine"):
            raise ValueError("SegmentedIndex serves the ip / cosine metrics")
        rows = [int(n) for n in rows_per_source]
        if not 1 <= len(rows) <= _lib.RR_MAX_SEGMENTS or min(rows) < 0:
            raise ValueError(f"need 1 .. {_lib.RR_MAX_SEGMENTS} sources with >= 0 rows each")
        self._proto = FlatIndex(d, metric=metric, dtype=dtype, device=device)   # query conversion, workspace, dims
        self.d, self.dim, self.metric, self.dtype, self.device = self._proto.d, self._proto.dim, metric, dtype, self._proto.device
        self.rows = rows
        self.id_offsets = [int(v) for v in (id_offsets if id_offsets is not None else [s << 40 for s in range(len(rows))])]
        self.mask_cols = [int(v) for v in (mask_cols if mask_cols is not None else range(len(rows)))]
        if len(self.id_offsets) != len(rows) or len(self.mask_cols) != len(rows):
            raise ValueError("one id offset and one mask column per source")
        A = _lib.RR_SEGMENT_ALIGN
        self.begins, off = [], 0
        for n in rows:
            self.begins.append(off)
            off += -(-n // A) * A
        self.n_rows_total = self.begins[-1] + rows[-1]            # nothing behind the last valid row is ever read
        if self.n_rows_total > 0xFFFFFFE0:
            raise ValueError("more than 2^32 - 32 rows in one segmented matrix")
        self._xb = torch.zeros((max(1, self.n_rows_total), self.dim), dtype=_TORCH_DTYPE[dtype], device=self.device)
        self._segs = (_lib.SegmentStruct * len(rows))(*[_lib.SegmentStruct(b, n, o, c, 0)
                                                        for b, n, o, c in zip(self.begins, rows, self.id_offsets, self.mask_cols)])
        self.ntotal = sum(rows)

    def rows_of(self, s):
        """Device view [n_s, dim] of source s (fill it with rows in the index dtype, zero padded to dim)."""
        return self._xb[self.begins[s]: self.begins[s] + self.rows[s]]

    def source(self, s):
        """FlatIndex over source s's slice of the matrix (no copy): the per-source search surface on the same memory."""
        idx = FlatIndex(self.d, metric=self.metric, dtype=self.dtype, device=self.device)
        idx.adopt(self.rows_of(s))
        return idx

    def fill(self, s, x, chunk_rows=1 << 18):
        """Ingest float32 rows [n_s, d] (numpy or tensor) into source s (normalised for cosine)."""
        if x.shape[0] != self.rows[s] or x.shape[1] != self.d:
            raise ValueError(f"source {s} holds [{self.rows[s]}, {self.d}] rows")
        out = self.rows_of(s)
        with torch.cuda.device(self.device):
            for b in range(0, x.shape[0], chunk_rows):
                part = x[b: b + chunk_rows]
                if isinstance(part, np.ndarray):
                    part = torch.from_numpy(np.ascontiguousarray(part, dtype=np.float32))
                part = part.to(self.device, dtype=torch.float32).contiguous()
                check(lib().rr_rows_to_half(part.data_ptr(), part.shape[0], self.d, self.d, out[b: b + part.shape[0]].data_ptr(),
                                            _RR_DTYPE[self.dtype], self.dim, int(self.metric == "cosine"), _stream_ptr()), "rr_rows_to_half")

    @classmethod
    def from_indexes(cls, indexes, id_offsets=None, mask_cols=None):
        """Pack existing FlatIndex objects (same d / metric / dtype / device) into one matrix and re-point each of them at its
        slice, so their own buffers can be released."""
        first = indexes[0]
        for i in indexes:
            if (i.d, i.metric, i.dtype, i.device) != (first.d, first.metric, first.dtype, first.device):
                raise ValueError("sources of one SegmentedIndex share dimension, metric, dtype and device")
        seg = cls(first.d, [i.ntotal for i in indexes], id_offsets, mask_cols, first.metric, first.dtype, first.device)
        for s, i in enumerate(indexes):
            seg.rows_of(s).copy_(i.xb)
            i.adopt(seg.rows_of(s))
        return seg

    def prepare_queries(self, xq):
        return self._proto.prepare_queries(xq)

    def search_prepared(self, xq_half, k, route_mask=None, out=None):
        """xq_half [nq, dim] index dtype -> (D f32 [nq,k], I i64 [nq,k]) CUDA tensors: the merged top-k over the selected sources.
        route_mask: optional bool/uint8 CUDA matrix [nq, C] (the router's mask; source s reads column mask_cols[s])."""
        if xq_half.dtype != self._xb.dtype or xq_half.dim() != 2 or xq_half.shape[1] != self.dim or not xq_half.is_contiguous():
            raise ValueError("search_prepared() needs contiguous [nq, dim] queries of the index dtype")
        nq = xq_half.shape[0]
        ws = self._proto._workspace(k)
        if out is None:
            D = torch.empty((nq, k), dtype=torch.float32, device=self.device)
            I = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        else:
            D, I = out
        mptr, mstride = None, 0
        if route_mask is not None:
            if (route_mask.dtype not in (torch.bool, torch.uint8) or route_mask.dim() != 2 or route_mask.shape[0] != nq
                    or not route_mask.is_cuda or route_mask.stride(1) != 1 or route_mask.shape[1] <= max(self.mask_cols)):
                raise ValueError("route_mask must be a bool/uint8 CUDA matrix [nq, C] with unit column stride covering every mask column")
            mptr, mstride = route_mask.data_ptr(), route_mask.stride(0)
        check(lib().rr_flat_search_segments(self._xb.data_ptr(), _RR_DTYPE[self.dtype], self.n_rows_total, self.dim, self._segs,
                                            len(self.rows), xq_half.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(), ws.data_ptr(),
                                            ws.numel(), mptr, mstride, _stream_ptr()), "rr_flat_search_segments")
        return D, I

    def search(self, xq, k, route_mask=None):
        """float32 [nq,d] -> (D, I) numpy arrays (merged over the selected sources)."""
        with torch.cuda.device(self.device):
            if route_mask is not None and not isinstance(route_mask, torch.Tensor):
                route_mask = torch.from_numpy(np.ascontiguousarray(route_mask, dtype=np.uint8)).to(self.device)
            D, I = self.search_prepared(self.prepare_queries(xq), int(k), route_mask)
            return D.cpu().numpy(), I.cpu().numpy()
