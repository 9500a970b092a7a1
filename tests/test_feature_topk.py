"""Per-feature top activations (reference src/whisper_sae/analysis/feature_viz.py:59-250, :425-484,
tests/test_analysis.py:83-250).

CPU: the oracle restatement (oracle/feature_topk.py) against golden set G13 written by the reference's own
TopKTracker; the host-side bookkeeping of the drop-in (dataclass, counters, JSON schema).  GPU: the device-side
tracker (wsae_feature_topk_update through the C ABI) bit-exact against G13 and the oracle, the reference's own
behavioural tests restated under their names, and order / batching invariance at the benchmark's sizes."""

from __future__ import annotations

import json

import numpy as np
import pytest
import torch

from oracle import synth
from oracle.feature_topk import FeatureTopK
from whisper_sae.analysis import FeatureActivation, TopKTracker, collect_top_activations


@pytest.fixture(scope="module")
def g13(golden_dir):
    return np.load(golden_dir / "g13_feature_topk.npz")


def golden_lists(g):
    """{feature: [(value, sample, position), ...]} as the reference returned them."""
    return {f: [(float(g["vals"][f, j]), int(g["samples"][f, j]), int(g["positions"][f, j])) for j in range(int(g["counts"][f]))]
            for f in range(int(g["H"]))}


def ordinal_to_sample_pos(g):
    """ordinal -> (sample, position) over G13's three updates (batch rows, then positions)."""
    table = []
    for u in range(3):
        act = g[f"act{u}"]
        for b in range(act.shape[0]):
            for t in range(act.shape[1]):
                table.append((int(g[f"samples{u}"][b]), t))
    return table


class TestOracleAgainstReference:
    def test_dense_updates_match_g13(self, g13):
        H, keep = int(g13["H"]), int(g13["keep"])
        o = FeatureTopK(H, keep)
        for u in range(3):
            act = g13[f"act{u}"]
            o.update_dense(act.reshape(-1, H))
        v, ords, c = o.arrays()
        table = ordinal_to_sample_pos(g13)
        assert np.array_equal(c, g13["counts"])
        assert o.total_activations == int(g13["total_activations"])
        want = golden_lists(g13)
        for f in range(H):
            got = [(float(v[f, j]), *table[int(ords[f, j])]) for j in range(int(c[f]))]
            assert got == want[f], f

    def test_compact_form_is_the_same_tracker(self, g13):
        # the TopK code of a row holds the row's positive entries: feeding (values, indices) gives the same lists
        H, keep = int(g13["H"]), int(g13["keep"])
        dense, compact = FeatureTopK(H, keep), FeatureTopK(H, keep)
        for u in range(3):
            a = g13[f"act{u}"].reshape(-1, H)
            width = int((a > 0).sum(1).max())
            idx = np.argsort(-a, axis=1, kind="stable")[:, :width].astype(np.int32)
            vals = np.take_along_axis(a, idx, axis=1)
            dense.update_dense(a)
            compact.update_compact(vals, idx)
        for x, y in zip(dense.arrays(), compact.arrays()):
            assert np.array_equal(x, y)

    def test_strict_greater_keeps_the_earlier_of_equal_values(self):
        o = FeatureTopK(4, 2)
        a = np.zeros((5, 4), np.float32)
        a[:, 1] = [0.5, 0.7, 0.5, 0.7, 0.2]
        o.update_dense(a)
        assert o.lists[1] == [(np.float32(0.7), 1), (np.float32(0.7), 3)]
        o.update_dense(a[:1] * 0 + np.float32(0.7))
        assert [e[1] for e in o.lists[1]] == [1, 3]


class TestHostSide:
    """No GPU: the parts of the reference's tests/test_analysis.py that need no update."""

    def test_creation_and_dict_round_trip(self):
        a = FeatureActivation(feature_idx=42, activation_value=0.85, sample_idx=10, position_idx=150, timestamp_ms=1500.0,
                              transcription="hello world")
        d = a.to_dict()
        assert list(d) == ["feature_idx", "activation_value", "sample_idx", "position_idx", "timestamp_ms", "transcription",
                           "transcription_context", "audio_path", "metadata"]
        assert FeatureActivation.from_dict(d) == a
        assert FeatureActivation(0, 0.5, 0, 0).metadata == {}

    def test_initialization(self):
        t = TopKTracker(num_features=128, k=10)
        assert (t.num_features, t.k, t.total_activations, t.samples_processed) == (128, 10, 0, 0)
        assert t.get_top_examples(5) == [] and t.get_feature_stats()[5]["num_examples"] == 0

    def test_k_beyond_a_wavefront_is_refused(self):
        with pytest.raises(ValueError):
            TopKTracker(num_features=8, k=65)

    def test_prune_keeps_the_segment_being_recorded(self, monkeypatch):
        # the bookkeeping of _record/_prune with the device merge stubbed out: after every update the newest segment is
        # still there and ordinals bisect to the segment that produced them
        t = TopKTracker(4, 2)
        t._prune_at = 8
        hv, ho, hc = t._host
        t._vals = object()  # "device state exists": _sync_host is a no-op while _host_valid stays True
        for u in range(100):
            base = t._record(1, 1, [500 + u], None, None)
            assert t._segments[-1].base == base and t._bases[-1] == base
            f = u % 4  # pretend the merge put this update at the head of list f
            ho[f, 1], ho[f, 0] = ho[f, 0], base
            hc[f] = min(hc[f] + 1, 2)
        assert len(t._segments) <= 17
        for f in range(4):
            for j in range(2):
                seg = t._segments[__import__("bisect").bisect_right(t._bases, int(ho[f, j])) - 1]
                assert seg.base == int(ho[f, j]) and seg.samples == [500 + int(ho[f, j])]

    def test_load_reads_the_reference_schema(self, tmp_path):
        # the JSON a reference TopKTracker.save writes (feature_viz.py:209-229)
        ex = [FeatureActivation(10, 0.7, 1, 0, 0.0, "b").to_dict(), FeatureActivation(10, 0.5, 0, 0, 0.0, "a").to_dict()]
        doc = {"num_features": 64, "k": 5, "total_activations": 3, "samples_processed": 3,
               "features": {"10": ex, "20": [FeatureActivation(20, 0.9, 2, 0, 0.0, "c").to_dict()]}}
        p = tmp_path / "tracker.json"
        p.write_text(json.dumps(doc))
        t = TopKTracker.load(p)
        assert (t.num_features, t.k, t.samples_processed, t.total_activations) == (64, 5, 3, 3)
        got = t.get_top_examples(10)
        assert [e.activation_value for e in got] == [pytest.approx(0.7), pytest.approx(0.5)]
        assert got[0].transcription == "b" and t.get_top_examples(20)[0].sample_idx == 2
        t.save(tmp_path / "again.json")
        assert json.loads((tmp_path / "again.json").read_text())["features"].keys() == doc["features"].keys()


@pytest.mark.gpu
class TestDeviceTracker:
    def test_g13_bit_exact(self, g13, device):
        H, keep = int(g13["H"]), int(g13["keep"])
        t = TopKTracker(H, keep, device=device)
        for u in range(3):
            act = g13[f"act{u}"]
            samples = g13[f"samples{u}"].tolist()
            a = torch.from_numpy(act[:, 0] if act.shape[1] == 1 else act).to(device)
            t.update(a, samples, transcriptions=[f"utt{s}" for s in samples])
        assert t.total_activations == int(g13["total_activations"])
        assert t.samples_processed == int(g13["samples_processed"])
        want = golden_lists(g13)
        for f in range(H):
            ex = t.get_top_examples(f)
            assert [(np.float32(e.activation_value), e.sample_idx, e.position_idx) for e in ex] == \
                   [(np.float32(v), s, p) for v, s, p in want[f]], f
            assert all(e.transcription == f"utt{e.sample_idx}" and e.timestamp_ms == e.position_idx * 10.0 for e in ex)
        st = t.get_feature_stats()
        assert np.allclose([st[f]["mean_activation"] for f in range(H)], g13["stats_mean"], rtol=1e-6)

    @pytest.mark.parametrize("keep", [1, 20, 64])
    def test_compact_code_matches_oracle_with_ties(self, device, keep):
        # quantised values: plenty of equal activations, inside the lists and at their boundary
        H, K, rows = 200, 8, 700
        raw = synth.counter_u64(rows * K, 77, 3).reshape(rows, K)
        vals = (((raw >> np.uint64(20)) % np.uint64(13)).astype(np.float32) - 3.0) / 4.0  # some <= 0: not activations
        idx = np.stack([np.argsort(synth.counter_u64(H, 78, r))[:K] for r in range(rows)]).astype(np.int32)
        o = FeatureTopK(H, keep)
        t = TopKTracker(H, keep, device=device)
        for lo, hi in ((0, 300), (300, 301), (301, 700)):
            o.update_compact(vals[lo:hi], idx[lo:hi])
            t.update_compact(torch.from_numpy(vals[lo:hi]).to(device), torch.from_numpy(idx[lo:hi]).to(device),
                             list(range(lo, hi)))
        t._sync_host()
        hv, ho, hc = t._host
        v, ords, c = o.arrays()
        assert np.array_equal(hc, c) and t.total_activations == o.total_activations
        mask = np.arange(keep)[None, :] < c[:, None]
        assert np.array_equal(hv[mask], v[mask]) and np.array_equal(ho[mask], ords[mask])

    def test_more_than_512_updates_prune_and_keep_the_join(self, device):
        # _prune ran between the append and the merge of an update and dropped the segment just added;
        # every list read after the 513th update then raised IndexError or joined the wrong sample
        H, K, keep, per = 64, 4, 3, 2
        o = FeatureTopK(H, keep)
        t = TopKTracker(H, keep, device=device)
        n_up = 1300
        for u in range(n_up):
            raw = synth.counter_u64(per * K, 91, u).reshape(per, K)
            vals = ((raw % np.uint64(1000)).astype(np.float32) + 1.0 + u * 0.01)
            idx = np.stack([np.argsort(synth.counter_u64(H, 92, u * per + r))[:K] for r in range(per)]).astype(np.int32)
            o.update_compact(vals, idx)
            t.update_compact(torch.from_numpy(vals).to(device), torch.from_numpy(idx).to(device),
                             [1000 + u * per + r for r in range(per)])
        assert len(t._segments) < n_up  # pruning did happen
        v, ords, c = o.arrays()
        for f in range(H):
            ex = t.get_top_examples(f)
            assert len(ex) == c[f]
            for j, e in enumerate(ex):
                assert np.float32(e.activation_value) == v[f, j]
                assert e.sample_idx == 1000 + int(ords[f, j]) and e.position_idx == 0
        assert len(t.get_all_top_examples()) == int((c > 0).sum())

    def test_dense_and_compact_agree_at_bench_size(self, device):
        # 16384 rows x k = 32 of 3072 features: the code the benchmark's encoder emits, one call; the same entries
        # as a dense matrix, in four calls; the same entries with the rows' k slots permuted
        H, K, rows, keep = 3072, 32, 16384, 20
        g = torch.Generator(device="cpu").manual_seed(5)
        idx = torch.stack([torch.randperm(H, generator=g)[:K] for _ in range(rows)]).to(torch.int32)
        vals = torch.rand(rows, K, generator=g) - 0.1
        a, b, c = (TopKTracker(H, keep, device=device) for _ in range(3))
        a.update_compact(vals.to(device), idx.to(device), list(range(rows)))
        dense = torch.zeros(rows, H)
        dense.scatter_(1, idx.long(), vals.clamp_min(0))
        for lo in range(0, rows, 4096):
            b.update(dense[lo:lo + 4096].to(device), list(range(lo, lo + 4096)))
        perm = torch.randperm(K, generator=g)
        c.update_compact(vals[:, perm].to(device), idx[:, perm].to(device), list(range(rows)))
        for t in (a, b, c):
            t._sync_host()
        for x, y, z in zip(a._host, b._host, c._host):
            assert np.array_equal(x, y) and np.array_equal(x, z)
        assert a.total_activations == b.total_activations == int((vals > 0).sum())
        # sortedness + the kept minimum really is the k-th largest of the feature's column
        hv, ho, hc = a._host
        assert (hc == keep).all() and (np.diff(hv, axis=1) <= 0).all()
        col = dense[:, 123].numpy()
        assert np.array_equal(np.sort(col)[::-1][:keep], hv[123])

    # ---- the reference's own tests (tests/test_analysis.py:94-250), device-side -------------------
    def test_update_single_sample(self, device):
        t = TopKTracker(num_features=64, k=5, device=device)
        a = torch.zeros(1, 64)
        a[0, 10], a[0, 20] = 0.5, 0.8
        t.update(a, sample_indices=[0])
        assert t.samples_processed == 1 and t.total_activations == 2
        assert [e.activation_value for e in t.get_top_examples(10)] == [pytest.approx(0.5)]
        assert [e.activation_value for e in t.get_top_examples(20)] == [pytest.approx(0.8)]

    def test_top_k_limit(self, device):
        t = TopKTracker(num_features=64, k=3, device=device)
        for i in range(5):
            a = torch.zeros(1, 64)
            a[0, 0] = float(i) / 10
            t.update(a, sample_indices=[i])
        assert [e.activation_value for e in t.get_top_examples(0)] == [pytest.approx(0.4), pytest.approx(0.3), pytest.approx(0.2)]

    def test_update_with_sequence_and_timestamps(self, device):
        t = TopKTracker(num_features=64, k=10, device=device)
        a = torch.zeros(1, 100, 64)
        a[0, 0, 10], a[0, 2, 10], a[0, 50, 10] = 0.5, 0.8, 0.3
        t.update(a, sample_indices=[0], transcriptions=["hello world"])
        ex = t.get_top_examples(10)
        assert [(e.position_idx, e.timestamp_ms) for e in ex] == [(2, 20.0), (0, 0.0), (50, 500.0)]
        assert ex[0].transcription == "hello world"

    def test_save_and_load(self, device, tmp_path):
        t = TopKTracker(num_features=64, k=5, device=device)
        a = torch.zeros(3, 64)
        a[0, 10], a[1, 10], a[2, 20] = 0.5, 0.7, 0.9
        t.update(a, sample_indices=[0, 1, 2], transcriptions=["a", "b", "c"])
        t.save(tmp_path / "tracker.json")
        loaded = TopKTracker.load(tmp_path / "tracker.json", device=device)
        assert (loaded.num_features, loaded.k, loaded.samples_processed) == (64, 5, 3)
        assert [e.activation_value for e in loaded.get_top_examples(10)] == [pytest.approx(0.7), pytest.approx(0.5)]
        # a loaded tracker keeps tracking
        b = torch.zeros(1, 64)
        b[0, 10] = 0.6
        loaded.update(b, sample_indices=[3], transcriptions=["d"])
        assert [e.transcription for e in loaded.get_top_examples(10)] == ["b", "d", "a"]

    def test_collect_top_activations_uses_the_compact_code(self, device):
        from whisper_sae.sae.model import TopKSAE
        torch.manual_seed(0)
        model = TopKSAE(64, 256, k=8).to(device)
        data = [torch.randn(32, 64, device=device) for _ in range(3)]
        t = collect_top_activations(model, data, num_features=256, k=4, device=device)
        assert t.samples_processed == 96
        o = FeatureTopK(256, 4)
        for x in data:
            o.update_dense(model.encode(x).cpu().numpy())
        t._sync_host()
        for x, y in zip(t._host, o.arrays()):
            assert np.array_equal(x, y)
