"""HyperDB facade: host glue on CPU, end-to-end query() on the GPU against the oracle.

The reference's HyperDB cannot be constructed offline (model download), so the
expected values restate the brute-force tail of reference _execute_query (hyperdb.py:1541-1575) with the
oracle: first decay on the host (hyperdb.py:1344), then oracle.rank(..., timestamps=first, recency_bias=rb).
"""
import numpy as np
import pytest


def _host_db(vectors=None, **kw):
    """HyperDB whose ``vectors`` come from a host array: lets the host-only glue (dict, save) run without a GPU."""
    from hyperdb import HyperDB

    class HostOnly(HyperDB):
        vectors = None
    db = HostOnly(**kw)
    HostOnly.vectors = vectors
    return db


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
    first = db._handle_timestamps(0.5, "timestamp")
    ts = np.array([d["timestamp"] for d in db.documents])
    assert np.allclose(first, 0.5 * np.exp(ts - ts.max()))
    # over the FILTERED documents (hyperdb.py:1555): the maximum is the newest KEPT document
    keep = np.array([True, True, True, False, False, False])
    assert np.allclose(db._handle_timestamps(0.5, "timestamp", keep), 0.5 * np.exp(ts[:3] - ts[2]))
    assert db._handle_timestamps(0, None) is None
    with pytest.raises(ValueError):
        db._handle_timestamps(0.5, "created")
    # a document without a timestamp only matters when it is kept
    del db.documents[5]["timestamp"]
    db._ts_cache.clear()
    with pytest.raises(ValueError, match="All timestamps must be populated"):
        db._handle_timestamps(0.5, "timestamp")
    assert db._handle_timestamps(0.5, "timestamp", keep) is not None
    db.documents[5]["timestamp"] = float(ts[5])
    db._ts_cache.clear()
    m = db._row_mask([("skip_doc", 2), ("metadata", {"info.type": "even"})])
    assert m.tolist() == [False, False, True, False, True, False]
    assert db._row_mask([("skip_doc", -2)]).tolist() == [True] * 4 + [False] * 2
    with pytest.raises(Exception):
        db._row_mask([("skip_doc", 6)])
    with pytest.raises(NotImplementedError):
        db._row_mask([("key", "name")])
    assert db._row_mask([("mask", [1, 0, 1, 1, 0, 0]), ("skip_doc", 1)]).tolist() == [False, False, True, True, False, False]
    with pytest.raises(ValueError):
        db._row_mask([("mask", [True, False])])
    with pytest.raises(ValueError):
        db._row_mask([("colour", "x")])


def test_sentence_filter_semantics():
    """Whole-word, punctuation-blind, case-blind, all words of a filter in ONE string of the document, all filters must hit
    (reference hyperdb.py:1136-1176); strings are tokenised once per document list."""
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
    assert db._row_mask([("sentence", "seed"), ("skip_doc", -1)]).tolist() == [False, False, False, True, False]
    assert HyperDB.tokenize("Hello, World! hello_world") == {"hello", "world", "helloworld"}
    deep = {"a": [{"b": [{"c": "needle in a haystack"}]}]}
    db.documents = [deep, {"x": 1}]
    db._invalidate_rows()
    assert db._row_mask([("sentence", "haystack needle")]).tolist() == [True, False]


def test_dict_and_aliases_without_gpu():
    from hyperdb import HyperDB
    db = _host_db(np.arange(10, dtype=np.float32).reshape(5, 2), metadata_keys=["info.type"])
    db.documents = _docs(5)
    db.source_indices = list(range(5))
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


def test_remove_document_host_bookkeeping():
    """Host lists after remove_document follow hyperdb.py:691-766: documents popped, source_indices renumbered to
    stay consecutive; device rows are tombstoned and row ids map back to the shifted document positions."""
    from hyperdb import HyperDB
    db = HyperDB()
    db.documents = [f"d{i}" for i in range(8)]
    db.source_indices = list(range(8))
    db.remove_document([5, 2])
    assert db.documents == ["d0", "d1", "d3", "d4", "d6", "d7"] and db.source_indices == [0, 1, 2, 3, 4, 5]
    db.remove_document(0)
    assert db.documents[0] == "d1" and db.source_indices == [0, 1, 2, 3, 4]
    with pytest.raises(IndexError):
        db.remove_document(5)
    # tombstone arithmetic (no index attached here, so set the state by hand)
    db._dead = np.array([0, 2, 5], dtype=np.int64)
    assert db._docs_of_rows([1, 3, 4, 6, 7]).tolist() == [0, 1, 2, 3, 4]


def test_add_timestamp_stamps_dict_documents():
    """add_timestamp (hyperdb.py:104-107, :582-588): 'timestamp' joins metadata_keys, dict documents get
    metadata.timestamp; checked on the host lists by stopping before the upload."""
    from hyperdb import HyperDB
    db = HyperDB(add_timestamp=True)
    assert "timestamp" in db.metadata_keys
    docs = [{"name": "a"}, "plain string", {"name": "b", "metadata": {"x": 1}}]
    with pytest.raises(Exception):                        # no GPU here: the upload raises after the host part ran
        db.add(docs, vectors=np.ones((3, 4), dtype=np.float32), add_timestamp=True)
    assert isinstance(docs[0]["metadata"]["timestamp"], float) and docs[2]["metadata"]["x"] == 1
    assert docs[2]["metadata"]["timestamp"] == docs[0]["metadata"]["timestamp"] and docs[1] == "plain string"


def test_save_formats_are_the_reference_layout(tmp_path):
    """save() writes what reference load() reads (hyperdb.py:901-1005): checked on the files themselves, no GPU."""
    import gzip, json, pickle, sqlite3
    from hyperdb import HyperDB
    db = _host_db(np.arange(12, dtype=np.float32).reshape(4, 3), fp_precision="float32")
    db.documents = _docs(4)
    db.source_indices = [0, 1, 2, 3]
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
    empty = _host_db(None)
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
    assert db.size() == 19_999 and db._dead.tolist() == [best]          # tombstoned on the device, not moved
    again = db.query(Q[0], top_k=1, metric="dot_product")
    assert again[0][0] != f"d{best}"
    # rows behind the tombstone map to shifted document positions; source_indices were renumbered (hyperdb.py:737-745)
    keep = np.ones(20_000, dtype=bool)
    keep[best] = False
    live = np.nonzero(keep)[0]
    oi, osc = orc.rank(V16[keep], Q[3], top_k=7, metric="dot_product")
    got = db.query(Q[3], top_k=7, metric="dot_product")
    assert orc.same_result_modulo_ties([r[2] for r in got], [r[1] for r in got], oi, osc, 1e-3)
    assert all(r[0] == f"d{live[r[2]]}" for r in got)
    # enough removals trigger the device-side compaction (hdb_index_gather): caches travel with the rows
    gone = rng.choice(19_999, size=6_000, replace=False)
    db.remove_document(gone.tolist())
    assert db._dead.size == 0 and db._index.n == db.size() == 13_999
    keep2 = np.ones(19_999, dtype=bool)
    keep2[gone] = False
    live2 = live[keep2]
    assert np.array_equal(db.vectors, V16[live2])
    for metric in ("cosine_similarity", "euclidean_metric", "hamming_distance"):
        oi, osc = orc.rank(V16[live2], Q[4].copy(), top_k=7, metric=metric)
        got = db.query(Q[4].copy(), top_k=7, metric=metric)
        if metric == "hamming_distance":
            assert [r[1] for r in got] == sorted(osc.tolist(), reverse=True)
        else:
            assert orc.same_result_modulo_ties([r[2] for r in got], [r[1] for r in got], oi, osc, 1e-3)
        assert all(r[0] == f"d{live2[r[2]]}" for r in got)
    db.add(["late"], vectors=V[:1])                                     # appending after a compaction still works
    assert db.size() == 14_000 and db.query(V16[0], top_k=2, metric="cosine_similarity")[0][1] > 0.999


@pytest.mark.gpu
def test_filters_with_recency_use_the_filtered_maximum():
    """the reference calls _handle_timestamps on the FILTERED documents (hyperdb.py:1555), so both decays are
    normalised by the newest KEPT document, and only kept documents need a timestamp.  The newest documents are
    filtered out here; with the global maximum exp(ts - max) would collapse to ~0 for unix timestamps."""
    from hyperdb import HyperDB
    from oracle import ranking_oracle as orc
    rng = np.random.default_rng(3)
    n, d = 400, 64
    V = rng.standard_normal((n, d)).astype(np.float32)
    docs = _docs(n)                                             # timestamps grow with i, one hour apart
    del docs[399]["timestamp"]                                  # filtered out below: must not raise
    db = HyperDB(documents=docs, vectors=V, metadata_keys=["timestamp", "info.type"], ann_metric="dot")
    q = rng.standard_normal(d).astype(np.float32)
    keep = np.array([i < 300 and i % 2 == 1 for i in range(n)])
    res = db.query(q, top_k=6, recency_bias=0.8, timestamp_key="timestamp", metric="cosine_similarity",
                   filters=[("skip_doc", -100), ("metadata", {"info.type": "odd"})])
    ts = np.array([dd["timestamp"] for dd in np.array(docs, dtype=object)[keep]])
    first = 0.8 * np.exp(ts - ts.max())
    oi, osc = orc.rank(V[keep], q, top_k=6, metric="cosine_similarity", timestamps=first, recency_bias=0.8)
    assert [r[2] for r in res] == list(np.nonzero(keep)[0][oi])
    assert np.allclose([r[1] for r in res], osc, atol=1e-5)
    assert max(r[1] for r in res) > 0.5                         # the recency term acts (it would be ~0 with the global max)
    with pytest.raises(ValueError, match="All timestamps must be populated"):
        db.query(q, top_k=6, recency_bias=0.8, timestamp_key="timestamp")          # unfiltered: document 399 counts
    # one document left: the reference's 2-D score and its Info line (ranking_algorithm.py:189-191)
    one = db.query(q, top_k=3, filters=[("mask", np.arange(n) == 17)], metric="dot_product")
    assert len(one) == 1 and one[0][2] == 17 and np.shape(one[0][1]) == (1,)
    assert np.allclose(one[0][1], V[17] @ q, atol=1e-4)


@pytest.mark.gpu
def test_config1_pokemon_shaped_documents(capsys):
    """BASELINE config 1 as written: 151 documents with the schema of demo/pokemon.jsonl (name, shortname, hp,
    info{id,type,weakness,description}, images{...}, moves[...]; values synthetic), seeded 151 x 384 vectors, cosine
    top-5 through HyperDB.query() -- against the oracle and the committed reference output (tests/golden/c1.npz)."""
    import os
    from hyperdb import HyperDB
    from oracle import ranking_oracle as orc
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1.npz"))
    V, q = g["V"], g["q"]
    types = ["psychic", "flying", "fire", "water", "grass", "electric", "rock"]
    docs = [{"name": f"Mon{i:03d}", "shortname": f"mon{i:03d}", "hp": 100 + 10 * (i % 20),
             "info": {"id": i + 1, "type": types[i % 7], "weakness": types[(i + 3) % 7], "description": f"Synthetic creature number {i}."},
             "images": {"photo": f"images/mon{i:03d}.jpg", "typeIcon": f"icons/{types[i % 7]}.jpg", "weaknessIcon": f"icons/{types[(i + 3) % 7]}.jpg"},
             "moves": [{"name": "Tackle", "dp": 40, "type": "normal"}, {"name": "Double Team", "type": "normal"}]} for i in range(151)]
    db = HyperDB(documents=docs, vectors=V, metadata_keys=["info.type"], ann_metric="dot")   # cosine != "dot": brute force
    res = db.query(q, top_k=5)
    assert [r[2] for r in res] == g["ref_idx"].tolist() == orc.rank(V, q, top_k=5)[0].tolist()
    assert np.allclose([r[1] for r in res], g["ref_scores"], atol=1e-5)
    assert res[0][0]["info"]["id"] == 143 and res[0][0]["name"] == "Mon142"
    fire = db.query(q, top_k=5, filters=[("metadata", {"info.type": "fire"})])
    keep = np.array([i % 7 == 2 for i in range(151)])
    oi, osc = orc.rank(V[keep], q, top_k=5)
    assert [r[2] for r in fire] == list(np.nonzero(keep)[0][oi]) and all(r[0]["info"]["type"] == "fire" for r in fire)


@pytest.mark.gpu
def test_recency_and_filters_stay_on_the_device():
    """The timestamps of a key, the row mask of a filter and the bias of both decays are resident: after the first query of
    a (filter, key, recency_bias) combination no O(N) host array is built (host_row_passes stands still), the answers equal
    the host restatement of the reference tail, and any change of the document list drops the caches."""
    from hyperdb import HyperDB
    from oracle import ranking_oracle as orc
    rng = np.random.default_rng(11)
    n, d = 20_000, 128
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    docs = [{"name": f"doc{i}", "info": {"type": "even" if i % 2 == 0 else "odd"}, "timestamp": 1.7e9 + 7.0 * i} for i in range(n)]
    db = HyperDB(documents=docs, vectors=V, fp_precision="float16", metadata_keys=["timestamp", "info.type"], cache_size=4)
    ts = np.array([x["timestamp"] for x in docs])
    combos = [dict(recency_bias=0.5, timestamp_key="timestamp"),
              dict(recency_bias=0.5, timestamp_key="timestamp", filters=[("skip_doc", -5000), ("metadata", {"info.type": "odd"})]),
              dict(recency_bias=-0.3, timestamp_key="timestamp", filters=[("skip_doc", 1000)]),
              dict(filters=[("metadata", {"info.type": "even"})])]
    keeps = [np.ones(n, bool), np.array([i < 15_000 and i % 2 == 1 for i in range(n)]), np.arange(n) >= 1000, np.arange(n) % 2 == 0]
    for kw in combos:
        db.query(rng.standard_normal(d).astype(np.float16), top_k=10, **kw)          # fills the caches
    passes = db.host_row_passes
    assert passes > 0
    for rnd in range(3):
        for kw, keep in zip(combos, keeps):
            q = rng.standard_normal(d).astype(np.float16)
            res = db.query(q, top_k=10, **kw)
            rb = kw.get("recency_bias", 0)
            first = rb * np.exp(ts[keep] - ts[keep].max()) if rb else None
            oi, osc = orc.rank(V[keep], q.copy(), top_k=10, metric="cosine_similarity", timestamps=first, recency_bias=rb)
            got_i, got_s = [r[2] for r in res], np.array([r[1] for r in res])
            assert orc.same_result_modulo_ties(np.array(got_i), got_s, np.nonzero(keep)[0][oi], osc, 1e-3), kw
    assert db.host_row_passes == passes, "cached combinations must not build O(N) host arrays"
    db.remove_document([3, 4, 5])                                 # tombstones: every per-row cache is rebuilt once
    res = db.query(V[10], top_k=3, recency_bias=0.5, timestamp_key="timestamp")
    assert db.host_row_passes > passes and res[0][0]["name"] == "doc10"
    p2 = db.host_row_passes
    db.query(V[11], top_k=3, recency_bias=0.5, timestamp_key="timestamp")
    assert db.host_row_passes == p2
    # a stored NaN document: queries raise like the reference (:150-151) until it is removed, then answer at once
    W = rng.standard_normal((200, 32)).astype(np.float32)
    W[17, 3] = np.nan
    db2 = HyperDB(documents=[f"d{i}" for i in range(200)], vectors=W)
    with pytest.raises(ValueError, match="NaN"):
        db2.query(W[0], top_k=3)
    db2.remove_document(17)
    assert db2.query(W[0], top_k=1, metric="dot_product")[0][0] == "d0"


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,name", [("pickle", "db.pickle.gz"), ("pickle", "db.pickle"), ("json", "db.json"), ("sqlite", "db.sqlite")])
def test_save_load_round_trip_and_mask_filter(tmp_path, fmt, name):
    """A database written in the reference's layout comes back into HBM and answers like the original; a caller-
    supplied mask filter restricts the rows on the device."""
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
    mouse = np.array([dd["text"].startswith("electric mouse") for dd in docs])      # the caller's own text predicate
    hits = db2.query(q, top_k=50, filters=[("mask", mouse)])
    assert len(hits) == 30 and all(r[0]["text"].startswith("electric mouse") for r in hits)
    assert hits[0][0]["name"] == "doc40"
    words = db2.query(q, top_k=50, filters=[("sentence", "electric mouse")])        # the reference's text filter: same rows
    assert [r[2] for r in words] == [r[2] for r in hits] and np.allclose([r[1] for r in words], [r[1] for r in hits])
    half = HyperDB(fp_precision="float16")
    half.load(path, format=fmt)                                    # lands in HBM in the requested precision
    assert half.vectors.dtype == np.float16 and half.query(q.astype(np.float16), top_k=1)[0][0]["name"] == "doc40"


@pytest.mark.gpu
def test_single_process_group_two_logical_shards_on_one_gpu():
    """hdb_group_* behind the drop-in API: HyperDB(devices=[0, 0]) and
    register_vectors(..., devices=[0, 0]) split the matrix into two row shards (both on cuda:0 here; one per GPU on a
    node), run them concurrently and merge their packed records -- the answers must equal the unsharded index bit for
    bit, including filters, recency with the GLOBAL newest timestamp, appends, removals and a forced exact re-run."""
    import hyperdb.ranking_algorithm as ranking
    from hyperdb import HyperDB
    from hyperdb._native import METRIC_IDS
    from oracle import ranking_oracle as orc
    rng = np.random.default_rng(8)
    n, d = 50_000, 384
    V = rng.standard_normal((n, d)).astype(np.float32).astype(np.float16)
    Q = rng.standard_normal((5, d)).astype(np.float16)
    one = ranking.register_vectors(V)
    two = ranking.register_vectors(V, devices=[0, 0])
    try:
        assert [s.n for s in two.index.shards] == [25_000, 25_000] and two.index.shards[1].row_base == 25_000
        ts = 1.7e9 + np.arange(n) * 0.5                          # newest rows in the LAST shard: per-shard maxima differ
        for metric in ("cosine_similarity", "dot_product", "euclidean_metric", "hamming_distance"):
            for kw in ({}, {"timestamps": ts, "recency_bias": 0.3}):
                a = ranking.rank_batch(one, Q.copy(), top_k=100, metric=metric, **kw)
                b = ranking.rank_batch(two, Q.copy(), top_k=100, metric=metric, **kw)
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (metric, kw.keys())
        i1, s1 = ranking.hyperDB_ranking_algorithm_sort(two, Q[0].copy(), top_k=10, metric="cosine_similarity")
        oi, osc = orc.rank(V, Q[0].copy(), top_k=10, metric="cosine_similarity")
        assert orc.same_result_modulo_ties(i1, s1, oi, osc, 1e-3)
        assert np.allclose(ranking.dot_product(two, Q[1]), ranking.dot_product(one, Q[1]))
        # a failed threshold on the shards (sample_target far too small) comes back through the exact selection
        two.index.set_option("sample_target", 16)
        bi, bs = two.index.topk(Q.astype(np.float32), 100, METRIC_IDS["dot_product"])
        two.index.set_option("sample_target", 0)
        ai, as_ = one.index.topk(Q.astype(np.float32), 100, METRIC_IDS["dot_product"])
        assert np.array_equal(ai, bi) and np.array_equal(as_, bs)
    finally:
        one.close()
        two.close()
    # eight logical shards, and a k beyond what the device merge kernels take (8 x 2048 > 8192: the group merges on the host)
    n8 = 200_000
    V8 = rng.standard_normal((n8, 128)).astype(np.float32).astype(np.float16)
    Q8 = rng.standard_normal((3, 128)).astype(np.float32)
    one = ranking.register_vectors(V8)
    eight = ranking.register_vectors(V8, devices=[0] * 8)
    try:
        assert len(eight.index.shards) == 8 and eight.index.shards[7].row_base == 175_000
        for metric in ("cosine_similarity", "euclidean_metric", "hamming_distance"):
            for k in (100, 2048, 3000):
                ai, as_ = one.index.topk(Q8, k, METRIC_IDS[metric])
                bi, bs = eight.index.topk(Q8, k, METRIC_IDS[metric])
                assert np.array_equal(ai, bi) and np.array_equal(as_, bs), (metric, k)
        assert all(s_.stat("fused") == 0 for s_ in eight.index.shards)          # shards that share a device: multi-kernel pipeline ...
        eight.index.shards[0].topk(Q8[:1], 10, METRIC_IDS["cosine_similarity"])
        assert eight.index.shards[0].stat("fused") != 0                          # ... for group calls only: the handle keeps its options
        for _ in range(50):                                                        # a query stream: workers stay hot, nothing leaks
            bi, bs = eight.index.topk(Q8[:1], 100, METRIC_IDS["dot_product"])
        ai, as_ = one.index.topk(Q8[:1], 100, METRIC_IDS["dot_product"])
        assert np.array_equal(ai, bi) and np.array_equal(as_, bs)
    finally:
        one.close()
        eight.close()
    # the facade on two shards: filters, double recency, append (goes behind the last shard), removal + compaction
    docs = _docs(2000)
    W = rng.standard_normal((2000, 64)).astype(np.float32)
    ref = HyperDB(documents=[dict(x) for x in docs], vectors=W, metadata_keys=["timestamp", "info.type"], ann_metric="dot")
    grp = HyperDB(documents=[dict(x) for x in docs], vectors=W, metadata_keys=["timestamp", "info.type"], ann_metric="dot", devices=[0, 0])
    q = rng.standard_normal(64).astype(np.float32)

    def same(**kw):
        ra, rb_ = ref.query(q, **kw), grp.query(q, **kw)
        assert [r[2] for r in ra] == [r[2] for r in rb_] and np.allclose([r[1] for r in ra], [r[1] for r in rb_], atol=1e-6), kw
        assert [r[0]["name"] for r in ra] == [r[0]["name"] for r in rb_]
    same(top_k=9)
    same(top_k=9, metric="euclidean_metric", filters=[("skip_doc", 700), ("metadata", {"info.type": "odd"})])
    same(top_k=9, metric="dot_product", recency_bias=0.7, timestamp_key="timestamp", filters=[("skip_doc", -300)])
    extra = rng.standard_normal((300, 64)).astype(np.float32)
    more_docs = [{"name": f"new{i}", "info": {"type": "odd"}, "timestamp": 1.8e9 + i} for i in range(300)]
    ref.add([dict(x) for x in more_docs], vectors=extra)
    grp.add([dict(x) for x in more_docs], vectors=extra)
    same(top_k=12, metric="cosine_similarity")
    gone = rng.choice(2300, size=900, replace=False).tolist()
    ref.remove_document(gone)
    grp.remove_document(gone)
    assert grp._dead.size == 0 and grp._index.n == 1400 and [s.row_base for s in grp._index.shards][0] == 0
    same(top_k=12, metric="cosine_similarity")
    same(top_k=5, metric="manhattan_distance", filters=[("metadata", {"info.type": "even"})])
    assert np.array_equal(ref.vectors, grp.vectors)
