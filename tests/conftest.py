import csv
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _read_csv(path):
    rows = []
    with open(path) as f:
        r = csv.reader(f)
        next(r)
        for line in r:
            if line:
                rows.append([line[0], int(line[1]), int(line[2])])
    return rows


class Golden:
    """The reference's known-answer tables (tests/golden/ranges_golden.json)."""

    def __init__(self):
        with open(os.path.join(GOLDEN, "ranges_golden.json")) as f:
            self.doc = json.load(f)
        self.tables = {k: _read_csv(os.path.join(GOLDEN, v)) for k, v in self.doc["tables"].items()}

    def rows(self, spec):
        return self.tables[spec] if isinstance(spec, str) else spec

    def cases(self, op):
        return [c for c in self.doc["cases"] + self.doc["unit"] if c["op"] == op]


@pytest.fixture(scope="session")
def golden():
    return Golden()


def encode_keys(*tables):
    """Map contig strings of several row lists to dense ids in byte-lexicographic
    order (the order the reference emits groups in, grouped_stream.rs:105-113).
    Returns (names, [ (key u32, start i64, end i64) per table ])."""
    names = sorted({r[0] for t in tables for r in t}, key=lambda s: s.encode())
    ids = {n: i for i, n in enumerate(names)}
    out = []
    for t in tables:
        out.append((np.array([ids[r[0]] for r in t], np.uint32),
                    np.array([r[1] for r in t], np.int64),
                    np.array([r[2] for r in t], np.int64)))
    return names, out


def synth(n, seed, nkeys=1, mean_len=150, span=1_000_000, dtype=np.int32):
    """Small seeded uniform intervals for oracle-vs-device comparisons."""
    rng = np.random.default_rng(seed)
    key = rng.integers(0, nkeys, n, dtype=np.uint32)
    length = 1 + rng.integers(0, 2 * mean_len - 1, n)
    start = rng.integers(0, max(span - 2 * mean_len, 1), n)
    end = start + length - 1
    return key, start.astype(dtype), end.astype(dtype)


def pair_set(b, p):
    """Order-free canonical form of a pair list (parity is on the row multiset)."""
    v = (b.astype(np.uint64) << np.uint64(32)) | p.astype(np.uint64)
    return np.sort(v)
