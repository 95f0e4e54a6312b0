"""Placement of a federation's data sources on G GPUs: balanced by predicted scan time, large sources cut into row slices.

The reference runs one process per data source (ragroute/ragroute.py:10-16; the 13 FeB4RAG / 4 MedRAG sources of
config.py:32-57), fans a query out to the selected ones (http_server.py:198-209) and merges the flat candidate lists
(http_server.py:280-293, rerank.py:3-9).  Nothing in that flow needs a source to be whole: the k best rows of a source are
the k best of the union of the k best rows of its row ranges, so a *row slice* `(source, row_begin, n_rows)` is a search unit
like any other — its candidates carry `id = (source << 40) + row_begin + local_row`, the route mask column is the source's,
and the merge's (score, ascending id) order makes the union exact ("each data source (or each slice of one
corpus) is an independent top-k problem").

Whole-source placement (source s -> GPU s mod G) cannot balance the real federations: FeB4RAG's msmarco alone is a quarter
of the bytes.  `plan()` pours the sources, ordered so that the sources of one query encoder are adjacent, into G bins of equal
predicted time and cuts a source on a 256-row boundary where a bin is full.  The pieces of one encoder that meet on a rank
form ONE search unit (a SegmentedIndex: one query conversion, one bootstrap, one chunk schedule — rr_flat_search_segments);
a lone piece is a plain FlatIndex search.  The plan is a pure function of (sources, G, cost model): every rank computes it
for itself, so nothing about the layout is communicated.

Host logic only (no device, no library): tests/test_placement.py runs on CPU."""
from dataclasses import dataclass, field

SHARD_SHIFT = 40
SLICE_ALIGN = 256          # rows: RR_SEGMENT_ALIGN (a segment starts on a multiple of 256 rows of its matrix)
MIN_SLICE_ROWS = 1 << 16   # a cut never leaves a piece smaller than this (its fixed cost would dominate)


def padded_dim(d):
    """Row width the scan kernels store for embedding dimension d — the rule of rr_padded_dim (include/ragroute_hip.h),
    restated so the planner needs no library; tests/test_placement.py checks it against the library."""
    d = int(d)
    if d <= 0 or d > 8192:
        raise ValueError(f"embedding dimension {d} is not supported (1 .. 8192)")
    if d <= 768:
        return -(-d // 128) * 128
    for p in (896, 1024, 1280, 1536):
        if d <= p:
            return p
    return -(-d // 128) * 128


@dataclass(frozen=True)
class Source:
    """One data source of the federation.  sid: global source id = column of the router mask = id >> 40;
    encoder: any hashable — sources with the same key receive the same query embeddings (config.py:37-71)."""
    sid: int
    rows: int
    dim: int
    encoder: object = None
    metric: str = "ip"
    dtype: str = "fp16"
    name: str = ""

    @property
    def group(self):
        """Sources that may share one segmented search: same encoder, width, metric and storage type."""
        return (self.encoder if self.encoder is not None else ("source", self.sid), padded_dim(self.dim), self.metric, self.dtype)

    @property
    def row_bytes(self):
        return padded_dim(self.dim) * 2


@dataclass(frozen=True)
class RowSlice:
    sid: int
    row_begin: int
    n_rows: int

    @property
    def id_offset(self):
        return (self.sid << SHARD_SHIFT) + self.row_begin


@dataclass(frozen=True)
class Unit:
    """What one scan call serves and one exchange slot carries: the local pieces of one encoder group."""
    group: tuple
    slices: tuple          # RowSlice, ascending (sid, row_begin) = ascending id offset

    @property
    def segmented(self):
        return len(self.slices) > 1


@dataclass(frozen=True)
class CostModel:
    """Predicted time of one 256-query search unit: fixed_ms (query conversion, bootstrap sample + select, compactions, launch
    edges: the ~0.08 ms intercept of profiles/r03/config4_feb4rag_per_source.json) + bytes / rate.  Rates in GB/s by padded
    width (same file: 4.3 TB/s at 1024 and 4096, 4.45 TB/s at 768); segment_ms per additional piece of a segmented unit."""
    fixed_ms: float = 0.08
    segment_ms: float = 0.01
    gbps_narrow: float = 4450.0     # padded width <= 768 (query-resident kernels)
    gbps_wide: float = 4300.0

    def row_ms(self, src):
        return src.row_bytes / ((self.gbps_narrow if padded_dim(src.dim) <= 768 else self.gbps_wide) * 1e6)


@dataclass
class Placement:
    ranks: list                      # per rank: list of Unit
    predicted_ms: list               # per rank
    sources: dict = field(default_factory=dict)   # sid -> Source

    @property
    def slots(self):
        """Exchange slots per rank = the largest local unit count (every rank knows it from the plan: no collective)."""
        return max(1, max(len(u) for u in self.ranks))

    @property
    def imbalance(self):
        """max / mean of the predicted per-rank times."""
        busy = [t for t in self.predicted_ms]
        return max(busy) / (sum(busy) / len(busy)) if sum(busy) > 0 else 1.0

    def slices_of(self, sid):
        return sorted((s for units in self.ranks for u in units for s in u.slices if s.sid == sid), key=lambda s: s.row_begin)

    def rank_of(self, sid, row):
        for r, units in enumerate(self.ranks):
            for u in units:
                for s in u.slices:
                    if s.sid == sid and s.row_begin <= row < s.row_begin + s.n_rows:
                        return r
        raise KeyError((sid, row))

    def describe(self):
        out = []
        for r, units in enumerate(self.ranks):
            out.append({"rank": r, "predicted_ms": round(self.predicted_ms[r], 4),
                        "units": [{"encoder": str(u.group[0]), "dim": u.group[1],
                                   "slices": [{"source": self.sources[s.sid].name or s.sid, "row_begin": s.row_begin, "n_rows": s.n_rows}
                                              for s in u.slices]} for u in units]})
        return out


def _ordered(sources):
    """Sources of one group adjacent (ascending sid inside a group); groups by descending bytes, so the cuts fall inside the
    large groups and the small single-source groups travel whole."""
    groups = {}
    for s in sources:
        groups.setdefault(s.group, []).append(s)
    order = sorted(groups.values(), key=lambda members: (-sum(m.rows * m.row_bytes for m in members), min(m.sid for m in members)))
    return [sorted(members, key=lambda m: m.sid) for members in order]


def _pour(groups, G, cap, cost, min_slice):
    """Fill ranks 0 .. G-1 in order up to `cap` ms each.  Returns (per-rank {group key: [RowSlice]}, per-rank ms), or None
    if the federation does not fit under this cap."""
    ranks = [dict() for _ in range(G)]
    load = [0.0] * G
    r = 0
    for members in groups:
        for src in members:
            begin, per_row = 0, cost.row_ms(src)
            while True:
                remaining = src.rows - begin
                # no segmented L2 search: every piece of an L2 source is a unit of its own
                key = src.group if src.metric != "l2" else src.group + (src.sid, begin)
                opening = cost.segment_ms if key in ranks[r] else cost.fixed_ms
                fit = int(max(0.0, cap - load[r] - opening) / per_row)
                if remaining == 0 or fit >= remaining:
                    take = remaining
                else:                                     # the rank fills up inside this source: cut on a 256-row boundary
                    take = fit // SLICE_ALIGN * SLICE_ALIGN
                    if remaining - take < min_slice:      # never leave a sliver behind the cut ...
                        take = (remaining - min_slice) // SLICE_ALIGN * SLICE_ALIGN
                    if take < min_slice:                  # ... nor in front of it
                        if load[r] == 0.0:
                            return None                   # an empty rank cannot take it either: the cap is too low
                        take = 0
                if take == 0 and remaining > 0:
                    r += 1
                    if r >= G:
                        return None
                    continue
                load[r] += opening + take * per_row
                ranks[r].setdefault(key, []).append(RowSlice(src.sid, begin, take))
                begin += take
                if begin >= src.rows:
                    break
                r += 1                                    # the rest of this source starts the next rank
                if r >= G:
                    return None
    return ranks, load


def plan(sources, G, cost=None, min_slice_rows=MIN_SLICE_ROWS):
    """sources: iterable of Source; G: number of GPUs.  Returns a Placement whose largest predicted per-rank time is minimal
    for this pouring order (bisection on the cap; at most G - 1 cuts, each on a 256-row boundary)."""
    sources = list(sources)
    if G < 1:
        raise ValueError("need at least one GPU")
    if len({s.sid for s in sources}) != len(sources):
        raise ValueError("source ids must be distinct")
    for s in sources:
        if s.rows < 0 or s.rows >= (1 << SHARD_SHIFT):
            raise ValueError(f"source {s.sid}: row count {s.rows} outside [0, 2^40)")
    cost = cost or CostModel()
    groups = _ordered(sources)
    total = sum(cost.row_ms(s) * s.rows for s in sources) + cost.fixed_ms * len(groups)
    lo, hi = total / G, total + cost.fixed_ms * len(sources) + 1e-9
    best = _pour(groups, G, hi, cost, min_slice_rows)
    if best is None:
        raise RuntimeError("placement: the one-rank cap does not fit (cost model inconsistent)")
    for _ in range(48):
        if hi - lo < 1e-5 * hi:
            break
        mid = 0.5 * (lo + hi)
        got = _pour(groups, G, mid, cost, min_slice_rows)
        if got is None:
            lo = mid
        else:
            best, hi = got, mid
    ranks, load = best
    out = []
    for per_group in ranks:
        units = [Unit(key[:4], tuple(sorted(sl, key=lambda s: (s.sid, s.row_begin)))) for key, sl in per_group.items()]
        units.sort(key=lambda u: (u.slices[0].sid, u.slices[0].row_begin))
        out.append(units)
    return Placement(out, load, {s.sid: s for s in sources})


def whole_source_plan(sources, G, cost=None):
    """The round-1..3 layout (source s -> GPU s mod G), as a Placement: the baseline `plan()` is measured against."""
    sources = list(sources)
    cost = cost or CostModel()
    ranks, load = [], []
    for r in range(G):
        mine = [s for i, s in enumerate(sorted(sources, key=lambda s: s.sid)) if i % G == r]
        ranks.append([Unit(s.group, (RowSlice(s.sid, 0, s.rows),)) for s in mine])
        load.append(sum(cost.fixed_ms + cost.row_ms(s) * s.rows for s in mine))
    return Placement(ranks, load, {s.sid: s for s in sources})


def federation(dataset, rows=None):
    """The reference's federations as Source lists: names / order / encoders from config.py:32-71, row counts of the public
    corpora (MedRAG snippet counts, Xiong et al. 2024 table 1; BEIR document counts, Thakur et al. 2021 table 1) unless
    `rows` overrides them, encoder widths of the public model cards."""
    from . import config as C
    if dataset not in ROWS:
        raise ValueError(f"no public row counts for dataset {dataset!r}")
    names = C.DATA_SOURCES[dataset]
    rows = dict(ROWS[dataset], **(rows or {}))
    out = []
    for sid, name in enumerate(names):
        enc = C.EMBEDDING_MODELS_PER_DATA_SOURCE[dataset][name][0]
        out.append(Source(sid, int(rows[name]), ENCODER_WIDTH[enc], enc, "ip", "fp16", name))
    return out


ROWS = {
    "medrag": {"pubmed": 23_900_000, "statpearls": 301_200, "textbooks": 125_800, "wikipedia": 29_900_000},
    "feb4rag": {"msmarco": 8_841_823, "trec-covid": 171_332, "nfcorpus": 3_633, "scidocs": 25_657, "nq": 2_681_468,
                "hotpotqa": 5_233_329, "fiqa": 57_638, "arguana": 8_674, "webis-touche2020": 382_545, "dbpedia-entity": 4_635_922,
                "fever": 5_416_568, "climate-fever": 5_416_593, "scifact": 5_183},
}
ENCODER_WIDTH = {"e5-large": 1024, "SGPT-5.8B-weightedmean-msmarco-specb-bitfit": 4096, "UAE-Large-V1": 1024, "all-mpnet-base-v2": 768,
                 "multilingual-e5-large": 1024, "ember-v1": 1024, "e5-base": 768, "gte-base": 768, "ncbi/MedCPT-Query-Encoder": 768,
                 "facebook/dpr-question_encoder-single-nq-base": 768}
