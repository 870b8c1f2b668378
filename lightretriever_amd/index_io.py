"""On-disk format of a flat inner-product index shard.

The reference persists `faiss.write_index(IndexFlatIP)` as `{prefix}.{ext}.faiss` next to a `{prefix}.{ext}.tsv` id map
(retriever/faiss_search.py:99-123, :478-488, :506-507; retriever/faiss_index.py:42-43).  The byte layout written here is Faiss's
own serialisation of IndexFlat as published in faiss/impl/index_write.cpp (v1.7.x - 1.8):

    u32   fourcc 'IxFI'                       (IndexFlatIP)
    i32   d
    i64   ntotal
    i64   dummy = 1 << 20,  i64 dummy = 1 << 20
    u8    is_trained = 1
    i32   metric_type = 0                     (METRIC_INNER_PRODUCT)
    u64   n_floats = ntotal * d               (WRITEXBVECTOR of the codes: size in 4-byte units)
    f32   x[ntotal * d]                       row-major

so a shard written here is a regular Faiss flat index file and a `{prefix}.flat.faiss` written by the reference loads here.
Faiss itself is not installed in this image: the layout is **unverified against a Faiss build** (round trip and header bytes
are tested).  Multi-rank: every rank writes `{prefix}.rank{r}-of-{R}.{ext}.faiss/.tsv` holding its own rows, so a 10 M-document
index reloads per rank without re-encoding and without one rank ever holding the whole matrix."""
from __future__ import annotations

import csv
import os
import struct
from typing import Iterable, Optional

import numpy as np

FOURCC_FLAT_IP = b"IxFI"
_HEADER = struct.Struct("<4siqqqBi")     # fourcc, d, ntotal, dummy, dummy, is_trained, metric_type  (37 bytes, packed)
HEADER_BYTES = _HEADER.size + 8          # + u64 vector size
MAPPING_TSV_KEYS = ["beir-docid", "faiss-docid"]


def shard_prefix(prefix: str, rank: int = 0, world: int = 1) -> str:
    return prefix if world == 1 else f"{prefix}.rank{rank}-of-{world}"


def write_flat_ip(fname: str, blocks: Iterable[np.ndarray], d: int, ntotal: int) -> None:
    """blocks: fp32 [n_i, d] arrays in row order (streamed: the shard comes off the GPU in chunks), sum n_i == ntotal."""
    tmp = fname + ".tmp"
    with open(tmp, "wb") as f:
        f.write(_HEADER.pack(FOURCC_FLAT_IP, d, ntotal, 1 << 20, 1 << 20, 1, 0))
        f.write(struct.pack("<Q", ntotal * d))
        rows = 0
        for b in blocks:
            b = np.ascontiguousarray(b, dtype="<f4")
            if b.ndim != 2 or b.shape[1] != d:
                raise ValueError(f"write_flat_ip: block {b.shape} does not match d={d}")
            f.write(b.tobytes())
            rows += b.shape[0]
        if rows != ntotal:
            raise ValueError(f"write_flat_ip: wrote {rows} rows, header says {ntotal}")
    os.replace(tmp, fname)


def read_flat_ip(fname: str) -> np.memmap:
    """-> read-only memmap fp32 [ntotal, d] over the file (no copy; the caller streams it to the GPU)."""
    size = os.path.getsize(fname)
    if size < HEADER_BYTES:
        raise ValueError(f"{fname}: too short for a flat index header")
    with open(fname, "rb") as f:
        fourcc, d, ntotal, _, _, trained, metric = _HEADER.unpack(f.read(_HEADER.size))
        (n_floats,) = struct.unpack("<Q", f.read(8))
    if fourcc != FOURCC_FLAT_IP:
        raise ValueError(f"{fname}: fourcc {fourcc!r} is not an inner-product flat index ('IxFI')")
    if metric != 0 or d <= 0 or ntotal < 0 or n_floats != ntotal * d or size != HEADER_BYTES + 4 * n_floats:
        raise ValueError(f"{fname}: inconsistent flat index header (d={d}, ntotal={ntotal}, floats={n_floats}, metric={metric}, bytes={size})")
    return np.memmap(fname, dtype="<f4", mode="r", offset=HEADER_BYTES, shape=(ntotal, d))


def save_dict_to_tsv(mapping: dict, output_path: str, keys: Optional[list] = None) -> None:
    """retriever/faiss_search.py:28-33."""
    with open(output_path, "w", newline="") as f:
        w = csv.writer(f, delimiter="\t", quoting=csv.QUOTE_MINIMAL)
        if keys:
            w.writerow(keys)
        for k, v in mapping.items():
            w.writerow([k, v])


def load_tsv_to_dict(input_path: str, header: bool = True) -> dict:
    """retriever/faiss_search.py:35-43."""
    out = {}
    with open(input_path, encoding="utf-8", newline="") as f:
        r = csv.reader(f, delimiter="\t", quoting=csv.QUOTE_MINIMAL)
        if header:
            next(r)
        for row in r:
            out[row[0]] = int(row[1])
    return out
