"""HyperDB facade: host glue on CPU, end-to-end query() on the GPU against the oracle.

The reference's HyperDB cannot be constructed offline (model download), so the
expected values restate the brute-force tail of reference _execute_query (hyperdb.py:1541-1575) with the
oracle: first decay on the host (hyperdb.py:1344), then oracle.rank(..., timestamps=first, recency_bias=rb).
"""
import numpy as np
import pytest


def _docs(n):
    return [{"name": f"doc{i}", "info": {"type": "even" if i % 2 == 0 else "odd", "n": i},
             "timestamp": 1.7e9 + 3600.0 * i} for i in range(n)]


# ------------------------------------------------------------------------------------------------ CPU
def test_constructor_validation_and_cache_key():
    from hyperdb import HyperDB
    with pytest.raises(ValueError):
        HyperDB(fp_precision="float8")
    with pytest.raises(ValueError):
        HyperDB(ann_metric="chebyshev")
    k1 = HyperDB._hashable_key(np.array([1.0, 2.0]), 5, True, [("skip_doc", 2), ("metadata", {"a": 1})], 0, None, "dot_product", 5)
    k2 = HyperDB._hashable_key(np.array([1.0, 2.0]), 5, True, [("skip_doc", 2), ("metadata", {"a": 1})], 0, None, "dot_product", 5)
    assert k1 == k2 and hash(k1) == hash(k2)
    assert k1 != HyperDB._hashable_key(np.array([1.0, 2.5]), 5, True, None, 0, None, "dot_product", 5)


def test_empty_db_and_signature():
    import inspect
    from hyperdb import HyperDB
    db = HyperDB()
    with pytest.raises(Exception, match="database is empty"):
        db.query(np.ones(4))
    sig = inspect.signature(HyperDB.query)
    assert list(sig.parameters) == ["self", "query_input", "top_k", "return_similarities", "filters", "recency_bias",
                                    "timestamp_key", "metric", "ann_percent"]
    assert sig.parameters["top_k"].default == 5 and sig.parameters["metric"].default == "cosine_similarity"
    isig = inspect.signature(HyperDB.__init__)
    assert list(isig.parameters)[:11] == ["self", "documents", "vectors", "select_keys", "embedding_function", "fp_precision",
                                          "add_timestamp", "metadata_keys", "ann_metric", "n_trees", "cache_size"]


def test_host_helpers_without_gpu():
    from hyperdb import HyperDB
    db = HyperDB(metadata_keys=["timestamp", "info.type"])
    db.documents = _docs(6)
    first = db._handle_timestamps(0.5, "timestamp", db.documents)
    ts = np.array([d["timestamp"] for d in db.documents])
    assert np.allclose(first, 0.5 * np.exp(ts - ts.max()))
    assert db._handle_timestamps(0, None, db.documents) is None
    with pytest.raises(ValueError):
        db._handle_timestamps(0.5, "created", db.documents)
    m = db._row_mask([("skip_doc", 2), ("metadata", {"info.type": "even"})])
    assert m.tolist() == [False, False, True, False, True, False]
    assert db._row_mask([("skip_doc", -2)]).tolist() == [True] * 4 + [False] * 2
    with pytest.raises(Exception):
        db._row_mask([("skip_doc", 6)])
    with pytest.raises(NotImplementedError):
        db._row_mask([("key", "name")])
    with pytest.raises(ValueError):
        db._row_mask([("colour", "x")])


def test_dict_and_aliases_without_gpu():
    from hyperdb import HyperDB
    db = HyperDB(metadata_keys=["info.type"])
    db.documents = _docs(5)
    db.source_indices = list(range(5))
    db._chunks = [np.arange(10, dtype=np.float32).reshape(5, 2)]
    assert [d["name"] for d in db.dict()] == [f"doc{i}" for i in range(5)]
    odd = db.dict(metadata=("info.type", "odd"))
    assert [d["name"] for d in odd] == ["doc1", "doc3"]
    with_vec = db.dict(vectors=True, metadata={"info.type": "even"})
    assert [d["vector"] for d in with_vec] == [[0.0, 1.0], [4.0, 5.0], [8.0, 9.0]]
    with pytest.raises(ValueError):
        db.dict(metadata=["info.type"])
    assert db.commit_pending() is None
    with pytest.raises(NotImplementedError):
        db.add_document({"name": "x"}, vectors=np.ones((2, 2)), count=2)
    assert HyperDB().dict() == []


def test_sentence_filter_semantics():
    """Whole-word, punctuation-blind, case-blind, all tokens in ONE string, all filters must hit
    (reference hyperdb.py:1136-1176)."""
    from hyperdb import HyperDB
    db = HyperDB()
    db.documents = [{"name": "Pikachu", "info": {"description": "An electric mouse; it stores electricity!"}},
                    {"name": "Raichu", "info": {"description": "Its tail discharges ELECTRICITY into the ground.", "tags": ["mouse", "electric"]}},
                    "a plain string document about electricity",
                    {"name": "Bulbasaur", "info": {"description": "A strange seed was planted on its back."}},
                    42]
    assert db._row_mask([("sentence", "electricity")]).tolist() == [True, True, True, False, False]
    assert db._row_mask([("sentence", "Electric mouse")]).tolist() == [True, False, False, False, False]   # both words in one string
    assert db._row_mask([("sentence", ["electricity", "tail"])]).tolist() == [False, True, False, False, False]
    assert db._row_mask([("sentence", "electric")]).tolist() == [True, True, False, False, False]          # whole words only
    assert db._row_mask([("sentence", "seed, planted!")]).tolist() == [False, False, False, True, False]
    assert HyperDB.tokenize("Hello, World! hello") == {"hello", "world"}


def test_save_formats_are_the_reference_layout(tmp_path):
    """save() writes what reference load() reads (hyperdb.py:901-1005): checked on the files themselves, no GPU."""
    import gzip, json, pickle, sqlite3
    from hyperdb import HyperDB
    db = HyperDB(fp_precision="float32")
    db.documents = _docs(4)
    db.source_indices = [0, 1, 2, 3]
    db._chunks = [np.arange(12, dtype=np.float32).reshape(4, 3)]
    for name in ("db.pickle", "db.pickle.gz"):
        db.save(str(tmp_path / name))
        opener = gzip.open if name.endswith(".gz") else open
        with opener(tmp_path / name, "rb") as f:
            data = pickle.load(f)
        assert sorted(data) == ["documents", "metadata_index", "source_indices", "split_info", "vectors", "vectors_normalized"]
        assert data["vectors"] == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0], [6.0, 7.0, 8.0], [9.0, 10.0, 11.0]]
        assert data["documents"] == db.documents and data["source_indices"] == [0, 1, 2, 3]
    db.save(str(tmp_path / "db.json"), format="json")
    assert json.load(open(tmp_path / "db.json"))["vectors"][3] == [9.0, 10.0, 11.0]
    db.save(str(tmp_path / "db.sqlite"), format="sqlite")
    con = sqlite3.connect(tmp_path / "db.sqlite")
    assert [json.loads(r[0]) for r in con.execute("SELECT data FROM documents")] == db.documents
    assert [json.loads(r[0]) for r in con.execute("SELECT vector FROM vectors ORDER BY document_id")][1] == [3.0, 4.0, 5.0]
    assert [r[0] for r in con.execute("SELECT value FROM source_indices")] == [0, 1, 2, 3]
    assert json.loads(con.execute("SELECT value FROM settings WHERE name='vectors_normalized'").fetchone()[0]) is False
    con.close()
    with pytest.raises(ValueError):
        db.save(str(tmp_path / "x"), format="xml")
    empty = HyperDB()
    empty.save(str(tmp_path / "nothing.pickle"))
    assert not (tmp_path / "nothing.pickle").exists()


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_query_matches_reference_tail(capsys):
    from hyperdb import HyperDB
    from oracle import ranking_oracle as orc
    rng = np.random.default_rng(0)
    n, d = 151, 384                                            # config 1: pokemon-sized, MiniLM-sized
    V = rng.standard_normal((n, d)).astype(np.float32)
    docs = _docs(n)
    db = HyperDB(documents=docs, vectors=V, metadata_keys=["timestamp", "info.type"], ann_metric="dot")
    q = V[142] + 0.05 * rng.standard_normal(d).astype(np.float32)
    res = db.query(q, top_k=5)
    oi, osc = orc.rank(V, q, top_k=5, metric="cosine_similarity")
    assert [r[0]["name"] for r in res] == [f"doc{i}" for i in oi]
    assert np.allclose([r[1] for r in res], osc, atol=1e-5)
    assert [r[2] for r in res] == list(oi)                      # source_index
    assert res[0][0]["name"] == "doc142"
    # cached
    assert db.query(q, top_k=5) is res and db.cache_hits == 1
    # documents only
    only = db.query(q, top_k=3, return_similarities=False)
    assert only == [docs[i] for i in oi[:3]]
    # double recency (hyperdb.py:1344 then ranking_algorithm.py:183)
    res = db.query(q, top_k=5, recency_bias=0.5, timestamp_key="timestamp", metric="dot_product")
    ts = np.array([dd["timestamp"] for dd in docs])
    first = 0.5 * np.exp(ts - ts.max())
    oi, osc = orc.rank(V, q, top_k=5, metric="dot_product", timestamps=first, recency_bias=0.5)
    assert [r[2] for r in res] == list(oi) and np.allclose([r[1] for r in res], osc, rtol=1e-5, atol=1e-5)
    # filters -> row mask; indices stay global
    res = db.query(q, top_k=4, filters=[("skip_doc", 100), ("metadata", {"info.type": "even"})], metric="euclidean_metric")
    keep = np.array([i >= 100 and i % 2 == 0 for i in range(n)])
    oi, osc = orc.rank(V[keep], q, top_k=4, metric="euclidean_metric")
    assert [r[2] for r in res] == list(np.nonzero(keep)[0][oi])
    assert np.allclose([r[1] for r in res], osc, atol=1e-5)
    # top_k clamp + warning text
    capsys.readouterr()
    res = db.query(q, top_k=500, filters=[("skip_doc", 150)])
    assert len(res) == 1 and "Warning: top_k (500) is greater than the number of filtered documents (1)" in capsys.readouterr().out
    # errors
    with pytest.raises(ValueError, match="Invalid metric"):
        db.query(q, metric="nope")
    with pytest.raises(ValueError, match="must match the dimension"):
        db.query(np.ones(7))
    with pytest.raises(ValueError):
        db.query(np.full(d, np.nan))


@pytest.mark.gpu
def test_add_remove_and_batch():
    from hyperdb import HyperDB
    from oracle import ranking_oracle as orc
    rng = np.random.default_rng(1)
    V = rng.standard_normal((20_000, 128)).astype(np.float32)
    db = HyperDB(documents=[f"d{i}" for i in range(10_000)], vectors=V[:10_000], fp_precision="float16")
    assert db.vectors.dtype == np.float16
    db.add([f"d{i}" for i in range(10_000, 20_000)], vectors=V[10_000:])
    assert db.size() == 20_000
    Q = rng.standard_normal((9, 128)).astype(np.float16)
    out = db.query_batch(Q, top_k=7, metric="dot_product")
    V16 = V.astype(np.float16)
    for qi in (0, 8):
        oi, osc = orc.rank(V16, Q[qi], top_k=7, metric="dot_product")
        got_i, got_s = [r[2] for r in out[qi]], [r[1] for r in out[qi]]
        assert orc.same_result_modulo_ties(got_i, got_s, oi, osc, 1e-3)
        assert all(r[0] == f"d{r[2]}" for r in out[qi])
    best = out[0][0][2]
    db.remove_document(best)
    assert db.size() == 19_999
    again = db.query(Q[0], top_k=1, metric="dot_product")
    assert again[0][0] != f"d{best}"


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,name", [("pickle", "db.pickle.gz"), ("pickle", "db.pickle"), ("json", "db.json"), ("sqlite", "db.sqlite")])
def test_save_load_round_trip_and_sentence_filter(tmp_path, fmt, name):
    """A database written in the reference's layout comes back into HBM and answers like the original; the sentence
    filter restricts the rows on the device."""
    from hyperdb import HyperDB
    rng = np.random.default_rng(7)
    n, d = 300, 64
    V = rng.standard_normal((n, d)).astype(np.float32)
    docs = [{"name": f"doc{i}", "text": ("electric mouse" if i % 10 == 0 else "grass seed") + f" number {i}",
             "timestamp": 1.7e9 + i} for i in range(n)]
    db = HyperDB(documents=docs, vectors=V, metadata_keys=["timestamp"])
    q = V[40] + 0.01 * rng.standard_normal(d).astype(np.float32)
    want = db.query(q, top_k=7, metric="euclidean_metric")
    path = str(tmp_path / name)
    db.save(path, format=fmt)
    db2 = HyperDB(fp_precision="float32", metadata_keys=["timestamp"])
    db2.load(path, format=fmt)
    assert db2.size() == n and db2.vectors.dtype == np.float32 and np.array_equal(db2.vectors, V)
    got = db2.query(q, top_k=7, metric="euclidean_metric")
    assert [r[0]["name"] for r in got] == [r[0]["name"] for r in want]
    assert np.allclose([r[1] for r in got], [r[1] for r in want], atol=1e-6)
    assert [r[2] for r in got] == [r[2] for r in want]
    hits = db2.query(q, top_k=50, filters=[("sentence", "electric mouse")])
    assert len(hits) == 30 and all(r[0]["text"].startswith("electric mouse") for r in hits)
    assert hits[0][0]["name"] == "doc40"
    half = HyperDB(fp_precision="float16")
    half.load(path, format=fmt)                                    # lands in HBM in the requested precision
    assert half.vectors.dtype == np.float16 and half.query(q.astype(np.float16), top_k=1)[0][0]["name"] == "doc40"
