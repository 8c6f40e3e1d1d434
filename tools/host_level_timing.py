#!/usr/bin/env python3
"""End-to-end time of the HOST-level C ABI (what the JNI shim calls) on BASELINE config 3: pageable host CSR in ->
H2D -> model build -> D2H of the indicator CSR -> host buffers out.  This is the PCIe-inclusive rate DESIGN.md quotes
next to bench.py's HBM-resident number."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from universal_recommender_amd import _lib, synth  # noqa: E402
from universal_recommender_amd import similarity_analysis as SA  # noqa: E402
from universal_recommender_amd.indexed_dataset import IndexedDataset  # noqa: E402


class _Dict:  # identity dictionaries: only sizes are needed here
    def __init__(self, n):
        self.size = n


cfg = synth.config3(float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
data = synth.generate(cfg)
ids = []
for (_, nc, rp, ci) in data:
    m = IndexedDataset.__new__(IndexedDataset)   # skip building 1.2M-entry string dictionaries
    m.row_ptr, m.col_idx, m.values, m.rowIDs, m.columnIDs = rp, ci, None, _Dict(cfg.n_users), _Dict(nc)
    m.create = lambda rp_, ci_, r_, c_, v_=None: type("Ind", (), {"row_ptr": rp_, "col_idx": ci_, "values": v_, "nnz": int(rp_[-1])})()
    ids.append(m)
lib = _lib.load(_lib.DEFAULT_PATH)
for it in range(3):
    t0 = time.perf_counter()
    res = SA.cooccurrencesIDSs(ids, randomSeed=42, library=lib)
    dt = time.perf_counter() - t0
    pairs = sum(s.pairs for s in SA.last_stats)
    print(f"run {it}: {dt * 1e3:.1f} ms end to end, {pairs} pairs -> {pairs / dt / 1e9:.2f} G pairs/s; device ms per event "
          f"{[round(s.ms_total, 2) for s in SA.last_stats]}; in {sum(d[2].nbytes + d[3].nbytes for d in data) / 1e6:.0f} MB, "
          f"out {sum(r.nnz for r in res) * 12 / 1e6:.0f} MB")
