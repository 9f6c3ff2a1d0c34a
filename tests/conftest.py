import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def make_synth_csr(nrows, dim, mean_nnz, seed, binary=True, label01=True, zipf=1.1):
    """Small host-side synthetic sample set (raw, without the bias column): power-law feature ids
    scattered over [0, dim-1), de-duplicated and sorted inside each row."""
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    nfeat = dim - 1
    wstar = rng.normal(0, 0.7, nfeat).astype(np.float32) * (rng.random(nfeat) < 0.3)
    perm = rng.permutation(nfeat)
    rowptr = [0]
    cols, vals, labels = [], [], []
    for _ in range(nrows):
        ln = max(0, int(rng.poisson(mean_nnz)))
        u = rng.random(ln)
        a = 1.0 - zipf
        ranks = np.floor((u * (nfeat ** a - 1.0) + 1.0) ** (1.0 / a)).astype(np.int64) - 1
        ids = np.unique(perm[np.clip(ranks, 0, nfeat - 1)])
        v = np.ones(ids.shape[0], np.float32) if binary else \
            rng.uniform(0.25, 2.0, ids.shape[0]).astype(np.float32)
        m = float(np.dot(wstar[ids], v)) - 0.3
        y = 1.0 if rng.random() < 1.0 / (1.0 + np.exp(-m)) else 0.0
        if not label01:
            y = 1.0 if y > 0 else -1.0
        cols.append(ids)
        vals.append(v)
        labels.append(y)
        rowptr.append(rowptr[-1] + ids.shape[0])
    col = np.concatenate(cols) if cols else np.zeros(0, np.int64)
    val = np.concatenate(vals) if vals else np.zeros(0, np.float32)
    return orc.Csr(np.array(rowptr, np.uint64), col.astype(np.uint32), val.astype(np.float32),
                   np.array(labels, np.float32), dim)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def write_libsvm(path, raw):
    """raw Csr (no bias column) -> libsvm text"""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        for r in range(raw.nrows):
            b, e = int(raw.rowptr[r]), int(raw.rowptr[r + 1])
            feats = " ".join("%d:%.9g" % (int(raw.col[k]), float(raw.val[k])) for k in range(b, e))
            f.write("%g %s\n" % (float(raw.label[r]), feats))
