"""torch.library custom ops over the C ABI (`torch.ops.ragroute.*`), the operator surface proposed in SURVEY §8b:

    ragroute::flat_topk(xb, xq, k, id_offset=0, l2=False) -> (D f32[nq,k], I i64[nq,k])     index.search, data_source.py:158,186,203
    ragroute::l2_normalize_(x) -> x                                                         faiss.normalize_L2, data_source.py:199
    ragroute::flat_topk_segments(xb, row_begins, rows, id_offsets, mask_cols, xq, k, route_mask=None) -> (D, I)
                                                                                            one search over the sources of one encoder
                                                                                            (config.py:37-71) = their per-source
                                                                                            index.search + http_server.py:280-293 + rerank.py:3-9
    ragroute::merge_topk(D, I, k, descending=True) -> (D, I)                                rerank.py:3-9, 28-34
    ragroute::merge_gathered(buf, slots, nq, k_in, k, descending=True) -> (D, I)            the same on the all-gathered exchange buffer, in place
    ragroute::rows_to_half(x, dim, bf16=False, normalize=False) -> Tensor                   ingest / query conversion
    ragroute::router_mlp(xq, folded weights..., b3, prob_threshold) -> (logits, mask)       router.py:241-283, 50-55

All take and return CUDA tensors, enqueue on the current stream and never synchronise.  The faiss-shaped classes in
flat_index.py are thin conveniences over the same entry points (they additionally cache the workspace)."""
import torch

from . import _lib
from ._lib import check, lib

_WS = {}


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _workspace(device, k):
    key = (device.index, k)
    ws = _WS.get(key)
    if ws is None:
        nbytes = lib().rr_flat_search_workspace_bytes(k)
        if nbytes == 0:
            raise ValueError(f"k must be in [1, {_lib.RR_MAX_K}]")
        _WS.clear()
        ws = _WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws


def _dtype_code(t):
    if t.dtype == torch.float16:
        return _lib.RR_DTYPE_F16
    if t.dtype == torch.bfloat16:
        return _lib.RR_DTYPE_BF16
    raise ValueError("corpus / queries must be float16 or bfloat16")


@torch.library.custom_op("ragroute::flat_topk", mutates_args=())
def flat_topk(xb: torch.Tensor, xq: torch.Tensor, k: int, id_offset: int = 0, l2: bool = False) -> tuple[torch.Tensor, torch.Tensor]:
    if not (xb.is_cuda and xq.is_cuda and xb.dim() == 2 and xq.dim() == 2 and xb.shape[1] == xq.shape[1] and xb.dtype == xq.dtype):
        raise ValueError("flat_topk: xb [n,dim] and xq [nq,dim] must be CUDA tensors of the same half dtype and padded width")
    xb, xq = xb.contiguous(), xq.contiguous()
    n, dim = xb.shape
    nq = xq.shape[0]
    D = torch.empty((nq, k), dtype=torch.float32, device=xb.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=xb.device)
    ws = _workspace(xb.device, k)
    with torch.cuda.device(xb.device):
        if l2:
            hn = torch.empty(max(1, n), dtype=torch.float32, device=xb.device)
            check(lib().rr_half_sqnorms(xb.data_ptr(), _dtype_code(xb), n, dim, hn.data_ptr(), _stream()), "rr_half_sqnorms")
            check(lib().rr_flat_search_l2(xb.data_ptr(), hn.data_ptr(), _dtype_code(xb), n, dim, xq.data_ptr(), nq, k, D.data_ptr(),
                                          I.data_ptr(), id_offset, ws.data_ptr(), ws.numel(), None, 0, _stream()), "rr_flat_search_l2")
        else:
            check(lib().rr_flat_search(xb.data_ptr(), _dtype_code(xb), n, dim, xq.data_ptr(), nq, k, D.data_ptr(), I.data_ptr(),
                                       id_offset, ws.data_ptr(), ws.numel(), None, 0, _stream()), "rr_flat_search")
    return D, I


@flat_topk.register_fake
def _(xb, xq, k, id_offset=0, l2=False):
    return xq.new_empty((xq.shape[0], k), dtype=torch.float32), xq.new_empty((xq.shape[0], k), dtype=torch.int64)


@torch.library.custom_op("ragroute::flat_topk_segments", mutates_args=())
def flat_topk_segments(xb: torch.Tensor, row_begins: list[int], rows: list[int], id_offsets: list[int], mask_cols: list[int],
                       xq: torch.Tensor, k: int, route_mask: torch.Tensor | None = None) -> tuple[torch.Tensor, torch.Tensor]:
    """rr_flat_search_segments: xb ONE matrix [n, dim] holding source s at rows [row_begins[s], row_begins[s] + rows[s]) (begins
    multiples of 256); route_mask optional bool/uint8 [nq, C], source s reads column mask_cols[s] (-1: always selected)."""
    n_seg = len(rows)
    if not (len(row_begins) == len(id_offsets) == len(mask_cols) == n_seg and 1 <= n_seg <= _lib.RR_MAX_SEGMENTS):
        raise ValueError("flat_topk_segments: one begin / row count / id offset / mask column per segment (1 .. 32 segments)")
    if not (xb.is_cuda and xq.is_cuda and xb.dim() == 2 and xq.dim() == 2 and xb.shape[1] == xq.shape[1] and xb.dtype == xq.dtype):
        raise ValueError("flat_topk_segments: xb [n,dim] and xq [nq,dim] must be CUDA tensors of the same half dtype and padded width")
    xb, xq = xb.contiguous(), xq.contiguous()
    nq = xq.shape[0]
    segs = (_lib.SegmentStruct * n_seg)(*[_lib.SegmentStruct(int(b), int(r), int(o), int(c), 0)
                                          for b, r, o, c in zip(row_begins, rows, id_offsets, mask_cols)])
    D = torch.empty((nq, k), dtype=torch.float32, device=xb.device)
    I = torch.empty((nq, k), dtype=torch.int64, device=xb.device)
    ws = _workspace(xb.device, k)
    mptr, mstride = None, 0
    if route_mask is not None:
        if route_mask.dtype not in (torch.bool, torch.uint8) or route_mask.dim() != 2 or route_mask.shape[0] != nq or not route_mask.is_cuda:
            raise ValueError("flat_topk_segments: route_mask must be a bool/uint8 CUDA matrix [nq, C]")
        route_mask = route_mask.contiguous()
        mptr, mstride = route_mask.data_ptr(), route_mask.stride(0)
    with torch.cuda.device(xb.device):
        check(lib().rr_flat_search_segments(xb.data_ptr(), _dtype_code(xb), xb.shape[0], xb.shape[1], segs, n_seg, xq.data_ptr(), nq, k,
                                            D.data_ptr(), I.data_ptr(), ws.data_ptr(), ws.numel(), mptr, mstride, _stream()),
              "rr_flat_search_segments")
    return D, I


@flat_topk_segments.register_fake
def _(xb, row_begins, rows, id_offsets, mask_cols, xq, k, route_mask=None):
    return xq.new_empty((xq.shape[0], k), dtype=torch.float32), xq.new_empty((xq.shape[0], k), dtype=torch.int64)


@torch.library.custom_op("ragroute::l2_normalize_", mutates_args=("x",))
def l2_normalize_(x: torch.Tensor) -> None:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()):
        raise ValueError("l2_normalize_: contiguous float32 CUDA matrix expected")
    with torch.cuda.device(x.device):
        check(lib().rr_l2_normalize_f32(x.data_ptr(), x.shape[0], x.shape[1], _stream()), "rr_l2_normalize_f32")


@torch.library.custom_op("ragroute::merge_topk", mutates_args=())
def merge_topk(D: torch.Tensor, I: torch.Tensor, k: int, descending: bool = True) -> tuple[torch.Tensor, torch.Tensor]:
    from .rerank import merge_topk as _merge
    return _merge(D, I, k, descending)


@merge_topk.register_fake
def _(D, I, k, descending=True):
    return D.new_empty((D.shape[0], k), dtype=torch.float32), I.new_empty((I.shape[0], k), dtype=torch.int64)


@torch.library.custom_op("ragroute::merge_gathered", mutates_args=())
def merge_gathered(buf: torch.Tensor, slots: int, nq: int, k_in: int, k: int, descending: bool = True) -> tuple[torch.Tensor, torch.Tensor]:
    """rr_merge_topk_gathered: buf uint8 [world, bytes per rank] as all_gather_into_tensor leaves the packed candidate buffers."""
    from .sharded import merge_gathered as _mg
    return _mg(buf, nq, k_in, slots, k, descending)


@merge_gathered.register_fake
def _(buf, slots, nq, k_in, k, descending=True):
    return buf.new_empty((nq, k), dtype=torch.float32), buf.new_empty((nq, k), dtype=torch.int64)


@torch.library.custom_op("ragroute::rows_to_half", mutates_args=())
def rows_to_half(x: torch.Tensor, dim: int, bf16: bool = False, normalize: bool = False) -> torch.Tensor:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2):
        raise ValueError("rows_to_half: float32 CUDA matrix expected")
    x = x.contiguous()
    out = torch.empty((x.shape[0], dim), dtype=torch.bfloat16 if bf16 else torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        check(lib().rr_rows_to_half(x.data_ptr(), x.shape[0], x.shape[1], x.shape[1], out.data_ptr(),
                                    _lib.RR_DTYPE_BF16 if bf16 else _lib.RR_DTYPE_F16, dim, int(normalize), _stream()), "rr_rows_to_half")
    return out


@rows_to_half.register_fake
def _(x, dim, bf16=False, normalize=False):
    return x.new_empty((x.shape[0], dim), dtype=torch.bfloat16 if bf16 else torch.float16)


@torch.library.custom_op("ragroute::router_mlp", mutates_args=())
def router_mlp(xq: torch.Tensor, w1q: torch.Tensor, c1: torch.Tensor, ln1_g: torch.Tensor, ln1_b: torch.Tensor, w2: torch.Tensor,
               b2: torch.Tensor, ln2_g: torch.Tensor, ln2_b: torch.Tensor, w3: torch.Tensor, model_of_source: torch.Tensor,
               b3: float, prob_threshold: float) -> tuple[torch.Tensor, torch.Tensor]:
    """CorpusRoutingNN forward with the folded fc1 (router.py:241-283, 50-55): xq f32 [nq, n_models, d_max], the folded weights
    of `ragroute_amd.router.fold_weights` (w1q [d_max,256], c1 [C,256], w2 [256,128] = fc2.weight.T, ...), model_of_source
    int32 [C] -> (logits f32 [nq,C], mask bool [nq,C])."""
    import ctypes
    f32 = [xq, w1q, c1, ln1_g, ln1_b, w2, b2, ln2_g, ln2_b, w3]
    if not all(t.is_cuda and t.dtype == torch.float32 for t in f32) or model_of_source.dtype != torch.int32 or not model_of_source.is_cuda:
        raise ValueError("router_mlp: float32 CUDA tensors (and an int32 CUDA model_of_source) expected")
    xq, w1q, c1, ln1_g, ln1_b, w2, b2, ln2_g, ln2_b, w3 = [t.contiguous() for t in f32]
    mos = model_of_source.contiguous()
    nq, n_models, d_max = xq.shape
    C = c1.shape[0]
    if w1q.shape != (d_max, 256) or c1.shape[1] != 256 or w2.shape != (256, 128) or mos.numel() != C:
        raise ValueError("router_mlp: weight shapes do not match the folded CorpusRoutingNN layout")
    st = _lib.RouterWeightsStruct(
        n_sources=C, d_max=d_max, n_models=n_models, reserved=0, model_of_source=mos.data_ptr(), w1q=w1q.data_ptr(), c1=c1.data_ptr(),
        ln1_g=ln1_g.data_ptr(), ln1_b=ln1_b.data_ptr(), w2=w2.data_ptr(), b2=b2.data_ptr(), ln2_g=ln2_g.data_ptr(), ln2_b=ln2_b.data_ptr(),
        w3=w3.data_ptr(), b3=float(b3), prob_threshold=float(prob_threshold), ln_eps=1e-5, reserved2=0.0)
    logits = torch.empty((nq, C), dtype=torch.float32, device=xq.device)
    mask = torch.empty((nq, C), dtype=torch.uint8, device=xq.device)
    with torch.cuda.device(xq.device):
        check(lib().rr_router_mlp(ctypes.byref(st), xq.data_ptr(), nq, logits.data_ptr(), mask.data_ptr(), _stream()), "rr_router_mlp")
    return logits, mask.view(torch.bool)


@router_mlp.register_fake
def _(xq, w1q, c1, ln1_g, ln1_b, w2, b2, ln2_g, ln2_b, w3, model_of_source, b3, prob_threshold):
    return xq.new_empty((xq.shape[0], c1.shape[0]), dtype=torch.float32), xq.new_empty((xq.shape[0], c1.shape[0]), dtype=torch.bool)
