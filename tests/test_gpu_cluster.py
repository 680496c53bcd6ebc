"""Row clustering on the device (bsmr_cluster_rows) against the host implementation, the
plain dense restatement and the reference's published logs - identical row order and cluster
count everywhere."""
import json
import time
from pathlib import Path

import numpy as np
import pytest

import synth
from test_clustering_exact import CASES, clustered_pattern

pytestmark = pytest.mark.gpu

GOLDEN = json.loads((Path(__file__).parent / "golden" / "reference_logs.json").read_text())["matrices"]


def logged_clusters(name, alpha):
    return next(r["bsmr_numClusters"] for r in GOLDEN[name]["runs"] if abs(r["alpha"] - alpha) < 1e-6)


@pytest.mark.parametrize("case", CASES, ids=[f"bins{-(-c[1] // c[2])}" for c in CASES])
@pytest.mark.parametrize("alpha", [0.1, 0.3, 0.6, 0.9])
def test_device_order_equals_host_and_oracle(engine, oracle, case, alpha):
    rows, cols, bw, groups, per_row, seed = case
    rows, cols, ro, ci = clustered_pattern(rows, cols, groups, per_row, seed)
    st, perm, clusters, stats = engine.cluster_rows_device(rows, cols, ro, ci, bw, alpha)
    assert st == engine.OK
    want, want_clusters = oracle.bsa_row_reordering(rows, cols, ro, ci, bw, alpha)
    assert np.array_equal(perm, want) and clusters == want_clusters
    pipe = engine.Pipeline(engine.CSR.from_arrays(rows, cols, ro, ci), alpha=alpha, delta=0.3, block_size=bw,
                           device=-1)
    assert np.array_equal(pipe.array("reorderedRows"), perm)
    assert stats["threads_per_pair"] == oracle.lib.oracle_cluster_threads(-(-cols // bw))


@pytest.mark.parametrize("alpha", [-0.5, 0.0, 0.3, 1.0, 1.5])
def test_device_edge_cases(engine, oracle, alpha):
    """empty rows, rows hidden in skipped bins, alphas outside (0, 1), a single row, no rows"""
    rng = np.random.default_rng(11)
    cols, bw = 4800, 16
    hidden_bins = [b for b in range(300) if (b % 96) >= 64]
    per = []
    for r in range(90):
        if r % 3 == 0:
            bins = rng.choice(hidden_bins, size=4, replace=False)
        elif r % 3 == 1:
            bins = np.concatenate([rng.choice(hidden_bins, size=2, replace=False), rng.integers(0, 64, 3)])
        else:
            bins = rng.integers(0, 64, 5)
        per.append(np.unique(bins * bw + rng.integers(0, bw, bins.size)))
    per[7] = np.zeros(0, np.int64)
    per[8] = np.zeros(0, np.int64)
    ro = np.zeros(len(per) + 1, dtype=np.uint32)
    ro[1:] = np.cumsum([c.size for c in per])
    ci = np.concatenate(per).astype(np.uint32)
    st, perm, clusters, _ = engine.cluster_rows_device(len(per), cols, ro, ci, bw, alpha)
    want, want_clusters = oracle.bsa_row_reordering(len(per), cols, ro, ci, bw, alpha)
    assert st == engine.OK and np.array_equal(perm, want) and clusters == want_clusters
    # one row / all rows empty
    st, perm, clusters, _ = engine.cluster_rows_device(1, 40, np.array([0, 2], np.uint32), np.array([3, 9], np.uint32), 16, alpha)
    assert st == engine.OK and perm.tolist() == [0] and clusters == 1
    st, perm, clusters, _ = engine.cluster_rows_device(3, 40, np.zeros(4, np.uint32), np.zeros(0, np.uint32), 16, alpha)
    assert st == engine.OK and perm.size == 0


@pytest.mark.parametrize("name,pattern,alphas", [
    ("mycielskian14", lambda: synth.mycielskian_pattern(14), (0.1, 0.3, 0.5, 0.7, 0.9)),
    ("Trefethen_20000", lambda: synth.trefethen_pattern(20000), (0.1, 0.3, 0.5, 0.7, 0.9)),
    ("wathen100", lambda: synth.wathen_pattern(100, 100), (0.3, 0.9)),
    ("mycielskian15", lambda: synth.mycielskian_pattern(15), (0.3,)),
])
def test_device_clustering_reproduces_reference_logs(engine, name, pattern, alphas):
    rows, cols, ro, ci = pattern()
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    for alpha in alphas:
        t0 = time.perf_counter()
        st, perm, clusters, stats = engine.cluster_rows_device(rows, cols, ro, ci, 16, alpha)
        wall = time.perf_counter() - t0
        assert st == engine.OK
        assert clusters == logged_clusters(name, alpha), (name, alpha)
        print(f"{name} alpha={alpha}: {clusters} clusters, device {stats['elapsed_ms']:.1f} ms (wall {wall * 1e3:.0f} ms), "
              f"{stats['passes']} passes, {stats['similarities']} similarities")
        if alpha == 0.3:
            pipe = engine.Pipeline(csr, alpha=alpha, delta=0.3, block_size=16, device=-1)
            assert np.array_equal(pipe.array("reorderedRows"), perm)


def test_clusters_running_ahead_give_the_host_order(engine):
    """A reddit-like row range (power-law degrees, most rows end in clusters of their own): the passes run ahead of the older
    clusters' decisions - tentative seeds, parked hits, drops - and the row order and cluster count are the host
    implementation's all the same."""
    rows, cols, ro, ci = synth.reddit_shard_like(rows=6000, seed=5)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    bw = csr.calculate_block_size(200 << 30)
    for alpha in (0.3, 0.6):
        st, perm, clusters, stats = engine.cluster_rows_device(rows, cols, ro, ci, bw, alpha)
        assert st == engine.OK
        pipe = engine.Pipeline(csr, alpha=alpha, delta=0.3, block_size=bw, device=-1)
        assert np.array_equal(pipe.array("reorderedRows"), perm) and pipe.num_clusters == clusters
        assert stats["passes_ahead"] > 0, stats
        print(f"reddit-like 6000 rows alpha={alpha}: {clusters} clusters, {stats['passes']} passes ({stats['passes_ahead']} ahead), "
              f"{stats['dropped_seeds']} dropped seeds, device {stats['elapsed_ms']:.1f} ms")


def test_a_grid_larger_than_the_pass_keeps_the_host_order(engine, monkeypatch):
    """with many more workgroups than a pass has items (BSMR_CLUSTER_GRID = 65535), workgroups without an
    item are still being scheduled when the closing workgroup has written the next pass's state.  The state is kept in
    two copies used in turn (a launch reads one, its closing workgroup writes the other), so such a workgroup sees its
    own pass's state and leaves: row order and cluster count stay the host implementation's."""
    monkeypatch.setenv("BSMR_CLUSTER_GRID", "65535")
    rows, cols, ro, ci = synth.reddit_shard_like(rows=6000, seed=5)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    bw = csr.calculate_block_size(200 << 30)
    for alpha in (0.3, 0.6):
        pipe = engine.Pipeline(csr, alpha=alpha, delta=0.3, block_size=bw, device=-1)
        for _ in range(3):
            st, perm, clusters, stats = engine.cluster_rows_device(rows, cols, ro, ci, bw, alpha)
            assert st == engine.OK
            assert np.array_equal(pipe.array("reorderedRows"), perm) and pipe.num_clusters == clusters, stats
    rows, cols, ro, ci = synth.mycielskian_pattern(12)
    for alpha in (0.3,):
        st, perm, clusters, stats = engine.cluster_rows_device(rows, cols, ro, ci, 16, alpha)
        pipe = engine.Pipeline(engine.CSR.from_arrays(rows, cols, ro, ci), alpha=alpha, delta=0.3, block_size=16, device=-1)
        assert st == engine.OK and np.array_equal(pipe.array("reorderedRows"), perm) and pipe.num_clusters == clusters


@pytest.mark.parametrize("rule", [1, 2])
def test_either_scheduling_rule_gives_the_host_order(engine, monkeypatch, rule):
    """BSMR_CLUSTER_RULE: 1 = clusters only judge what every older one has decided (round 2's rule, the kernel's cap on
    clusters that may run unconfirmed), 2 = the host switches between the two rules by the measured rows per millisecond,
    batch by batch.  What a pass may judge early changes, what the clusters are does not: row order and cluster count stay
    the host implementation's."""
    monkeypatch.setenv("BSMR_CLUSTER_RULE", str(rule))
    cases = [(synth.reddit_shard_like(rows=6000, seed=5), None), (synth.wathen_pattern(nx=30, ny=30), 16), (synth.nips_like(), None)]
    for (rows, cols, ro, ci), block in cases:
        csr = engine.CSR.from_arrays(rows, cols, ro, ci)
        bw = block or csr.calculate_block_size(200 << 30)
        for alpha in (0.3, 0.6):
            st, perm, clusters, stats = engine.cluster_rows_device(rows, cols, ro, ci, bw, alpha)
            assert st == engine.OK
            pipe = engine.Pipeline(csr, alpha=alpha, delta=0.3, block_size=bw, device=-1)
            assert np.array_equal(pipe.array("reorderedRows"), perm) and pipe.num_clusters == clusters, (rule, rows, alpha, stats)
            if rule == 1:
                assert stats["passes_ahead"] <= 1, stats
