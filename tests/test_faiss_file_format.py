"""CPU: the FAISS IndexFlat file reader against byte-level fixtures written field by field from FAISS 1.7.4's
serialisation (impl/index_write.cpp: `write_index_header` + the IndexFlat branch), NOT by
`ragroute_amd.data_source.write_faiss_flat_index` — a wrong offset or metric code in the reader cannot hide behind a
writer that shares the mistake.  Replaces `faiss.read_index` (reference ragroute/data_source.py:71).

    uint32  fourcc         "IxFI" IndexFlatIP | "IxF2" IndexFlatL2 | "IxFl" IndexFlat, other metric
    int     d
    int64   ntotal
    int64   dummy = 1 << 20, int64 dummy = 1 << 20
    bool    is_trained
    int     metric_type    0 = METRIC_INNER_PRODUCT, 1 = METRIC_L2, 2 = METRIC_L1, ...
    [float  metric_arg     only if metric_type > 1]
    uint64  size           codes.size() / 4  (WRITEXBVECTOR: 4-byte units)
    float   codes[size]    row-major vectors
"""
import struct

import numpy as np
import pytest

from ragroute_amd.data_source import read_faiss_flat_index, write_faiss_flat_index

PAYLOAD = [1.0, -2.0, 0.5, 4.0,
           -8.0, 16.0, 0.25, -0.125]                 # 2 x 4
PAYLOAD_HEX = ("0000803f" "000000c0" "0000003f" "00008040"
               "000000c1" "00008041" "0000803e" "000000be")

IXFI_HEX = ("49784649"                                # 'I' 'x' 'F' 'I'
            "04000000"                                # d = 4
            "0200000000000000"                        # ntotal = 2
            "0000100000000000" "0000100000000000"     # dummy = 1 << 20, twice
            "01"                                      # is_trained = true
            "00000000"                                # metric_type = 0 (inner product)
            "0800000000000000"                        # size = 8 floats
            + PAYLOAD_HEX)
IXF2_HEX = ("49784632" "04000000" "0200000000000000" "0000100000000000" "0000100000000000" "01"
            "01000000"                                # metric_type = 1 (L2)
            "0800000000000000" + PAYLOAD_HEX)
IXFL_L1_HEX = ("4978466c" "04000000" "0200000000000000" "0000100000000000" "0000100000000000" "01"
               "02000000"                             # metric_type = 2 (L1) ...
               "00000000"                             # ... followed by metric_arg = 0.0f
               "0800000000000000" + PAYLOAD_HEX)
HNSW_HEX = "49484e66" + "04000000" + "00" * 64      # "IHNf" (IndexHNSWFlat): a non-flat index


def _write(tmp_path, name, hexstr):
    p = tmp_path / name
    p.write_bytes(bytes.fromhex(hexstr))
    return str(p)


def test_payload_hex_is_what_it_says():
    assert bytes.fromhex(PAYLOAD_HEX) == struct.pack("<8f", *PAYLOAD)


@pytest.mark.parametrize("mmap", [True, False])
def test_reads_index_flat_ip_bytes(tmp_path, mmap):
    xb, metric = read_faiss_flat_index(_write(tmp_path, "ip.index", IXFI_HEX), mmap=mmap)
    assert metric == "ip" and xb.dtype == np.float32 and xb.shape == (2, 4)
    assert np.array_equal(np.asarray(xb), np.array(PAYLOAD, np.float32).reshape(2, 4))


def test_reads_index_flat_l2_bytes(tmp_path):
    xb, metric = read_faiss_flat_index(_write(tmp_path, "l2.index", IXF2_HEX))
    assert metric == "l2" and np.array_equal(np.asarray(xb), np.array(PAYLOAD, np.float32).reshape(2, 4))


def test_other_metrics_are_refused_not_read_as_l2(tmp_path):
    with pytest.raises(ValueError, match="metric_type 2"):
        read_faiss_flat_index(_write(tmp_path, "l1.index", IXFL_L1_HEX))


def test_non_flat_index_raises_a_clean_error(tmp_path):
    """The service loop logs the exception and carries on (data_source.py:137-138): it must be an Exception, not a crash."""
    with pytest.raises(ValueError, match="not a flat FAISS index"):
        read_faiss_flat_index(_write(tmp_path, "hnsw.index", HNSW_HEX))
    with pytest.raises(ValueError):
        read_faiss_flat_index(_write(tmp_path, "empty.index", ""))


def test_inconsistent_headers_are_refused(tmp_path):
    with pytest.raises(ValueError, match="contradicts"):                     # IxFI fourcc with metric_type 1
        read_faiss_flat_index(_write(tmp_path, "a.index", IXFI_HEX.replace("01" "00000000" "08", "01" "01000000" "08", 1)))
    with pytest.raises(ValueError, match="does not match"):                  # size field != ntotal * d
        read_faiss_flat_index(_write(tmp_path, "b.index", IXFI_HEX.replace("0800000000000000", "0700000000000000", 1)))
    with pytest.raises(ValueError, match="payload bytes"):                   # truncated payload
        read_faiss_flat_index(_write(tmp_path, "c.index", IXFI_HEX[:-8]))


def test_writer_emits_exactly_these_bytes(tmp_path):
    """The fixture writer used by other tests must produce the documented layout byte for byte."""
    xb = np.array(PAYLOAD, np.float32).reshape(2, 4)
    for metric, want in (("ip", IXFI_HEX), ("l2", IXF2_HEX)):
        p = tmp_path / f"w_{metric}.index"
        write_faiss_flat_index(str(p), xb, metric)
        assert p.read_bytes().hex() == want
