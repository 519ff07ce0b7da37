"""``HyperDB`` facade over the MI355X ranking path (counterpart of reference hyperdb/hyperdb.py).

Scope: ONLY what sits either side of the hot path -- the constructor that
takes documents + precomputed vectors, ``query()`` with the reference's signature and return shapes,
and the ~30 lines of ``_execute_query`` around the ranking call (hyperdb.py:1461-1469, :1541-1575).
Also here, from the "next" rows of SURVEY.md section 8f: ``load`` (and its inverse ``save``) in the reference's own
pickle, json and sqlite layouts (hyperdb.py:769-1005) so an existing database file goes straight into HBM.  Everything
else the reference class does (sentence-transformer embedding, Annoy, the ``key`` filter that re-embeds sub-documents) is
out of scope and is NOT re-implemented here; a caller that has its own predicate passes its result as a ``mask`` filter.

Matrix residency: the N x d matrix lives ONLY in HBM (there is no host mirror; ``.vectors`` copies it back on demand).
``remove_document`` tombstones the rows (O(1) on the device: they join every query's row mask) and compacts the matrix
on the device, caches included, once a quarter of the rows are dead (``GpuIndex.compact`` -> hdb_index_gather).

Differences from the reference, all deliberate:
  * the N x d matrix is registered once on the GPU (``GpuIndex``) -- no per-query NaN scan, copy or
    re-normalisation (ranking_algorithm.py:150,:153,:37);
  * exact search always: the Annoy branch (hyperdb.py:1484-1487, :1546-1552) is bypassed, because a full
    HBM scan (~1.4 ms at N=10M) is faster than the reference's ANN candidate filtering and needs no
    O(N) Python index rebuild on every add/remove (hyperdb.py:218-220);
  * ranked rows map to documents by row id, O(k), instead of ``self.documents.index(document)``
    (hyperdb.py:1568, O(N) dict comparisons per hit);
  * ``filters``: ``skip_doc``, ``metadata`` equality and ``sentence`` (whole-word text search, hyperdb.py:1136-1176: a host
    predicate over token sets built once per document list) become a device row mask, cached on the device per filter, and so
    does the additive ``("mask", bool_array)`` filter; ``key`` (re-embeds sub-documents, hyperdb.py:1087) needs the embedding
    pipeline and raises NotImplementedError;
  * recency: the timestamps of a key are a resident float64 column; both decays of a query are one kernel
    (hdb_recency_bias_twice) whose result is kept per (key, filter, recency_bias) -- no O(N) host array per query;
  * strings can be queried only when an ``embedding_function`` is supplied (no model download);
  * additive ``devices=[...]``: the matrix is row-sharded over several GPUs and every query runs on all of them in one
    call of the same single-process API (``GpuGroup`` -> hdb_group_topk_host).
Kept from the reference: argument names and defaults, metric whitelist and messages, the top_k warning,
``(document, score, source_index)`` / ``document`` return shapes, the LRU result cache, and the DOUBLE
application of the recency decay when going through ``query()`` (hyperdb.py:1344 then
ranking_algorithm.py:183).
"""
from __future__ import annotations

import datetime
import gzip
import json
import pickle
import re
import sqlite3
import string
from collections import OrderedDict
from contextlib import closing

import math
from operator import itemgetter
import numpy as np
import torch

from . import ranking_algorithm as ranking
from ._native import GpuIndex, METRIC_IDS
from .group import GpuGroup

__all__ = ["HyperDB"]

_METRICS = ['dot_product', 'cosine_similarity', 'euclidean_metric', 'manhattan_distance', 'jaccard_similarity',
            'pearson_correlation', 'hamming_distance']


class HyperDB:
    def __init__(self, documents=None, vectors=None, select_keys=None, embedding_function=None, fp_precision="float32",
                 add_timestamp=False, metadata_keys=None, ann_metric="cosine", n_trees=10, cache_size=256, device=None,
                 devices=None):
        if fp_precision not in ["float16", "float32", "float64"]:
            raise ValueError("Unsupported floating-point precision.")                       # hyperdb.py:65-66
        accepted = ["angular", "euclidean", "manhattan", "hamming", "dot", "cosine"]
        if ann_metric not in accepted:
            raise ValueError(f"Unsupported ANN metric. Accepted values are: {', '.join(accepted)}")  # :69-71
        self.fp_precision = getattr(np, fp_precision)
        self.embedding_function = embedding_function
        self.select_keys = [select_keys] if isinstance(select_keys, str) else select_keys
        self.metadata_keys = [metadata_keys] if isinstance(metadata_keys, str) else (metadata_keys or [])
        self.add_timestamp = add_timestamp
        if self.add_timestamp and "timestamp" not in self.metadata_keys:      # hyperdb.py:104-107
            self.metadata_keys.append("timestamp")
        self.ann_metric, self.n_trees = ann_metric, n_trees          # accepted for compatibility; no ANN is built
        self.device = device
        self.devices = list(devices) if devices else None      # several GPUs: the matrix is row-sharded over them (GpuGroup)
        self.documents, self.source_indices = [], []
        self._index = None
        self._dead = np.zeros(0, dtype=np.int64)     # tombstoned device rows (ascending); see remove_document
        self._ts_cache = {}                           # timestamp_key -> float64 array over documents (NaN = missing)
        # resident per-row inputs of the ranking call, rebuilt only when the document list changes (_invalidate_rows):
        self._mask_cache = OrderedDict()              # filter key -> (per-shard uint8 device masks | None, documents kept, bool array over documents | None)
        self._ts_dev = {}                             # timestamp_key -> per-shard float64 device columns over the device rows
        self._bias_cache = OrderedDict()              # (timestamp_key, filter key, recency_bias) -> per-shard float32 device bias
        self._str_tokens = None                       # per document: token sets of the strings inside it (sentence filter)
        self.host_row_passes = 0                      # O(N) host arrays built so far (cache misses only; tests watch it)
        self._cache = OrderedDict()
        self._cache_size = cache_size
        self.cache_hits = self.cache_misses = 0
        if documents is not None or vectors is not None:
            self.add(documents, vectors, add_timestamp=self.add_timestamp)

    # ---------------------------------------------------------------- matrix lifecycle (hyperdb.py:496-766)
    def _embed(self, documents):
        if self.embedding_function is None:
            raise ValueError("vectors must be given: this build ships no embedding model "
                             "(pass vectors=... or an embedding_function).")
        return np.asarray(self.embedding_function(documents))

    def add(self, documents, vectors=None, add_timestamp=False):
        """Append documents with their vectors (hyperdb.py:548-566 without chunking/embedding).

        The device matrix grows in place (capacity doubling) and only the new rows' caches are computed:
        amortised O(new rows), where the reference re-concatenates the whole array (hyperdb.py:503-509) and
        rebuilds the Annoy index (:680) on every commit."""
        if documents is None:
            raise ValueError("documents are required")
        if not isinstance(documents, list):
            documents = [documents]
        if vectors is None:
            vectors = self._embed(documents)
        vectors = np.asarray(vectors)
        if vectors.ndim == 1:
            vectors = vectors.reshape(1, -1)
        if vectors.ndim != 2 or len(vectors) != len(documents):
            raise ValueError("All vectors must have the same dimension, one per document.")   # hyperdb.py:139-164
        if self._index is not None and vectors.shape[1] != self._index.d:
            raise ValueError("All vectors must have the same dimension, one per document.")
        vectors = vectors.astype(self.fp_precision, copy=False)
        if add_timestamp:                                   # hyperdb.py:582-588: dict documents get metadata.timestamp
            now = float(datetime.datetime.now().timestamp())
            for doc in documents:
                if isinstance(doc, dict):
                    doc.setdefault('metadata', {})['timestamp'] = now
        start = len(self.documents)
        self.documents = list(self.documents) + list(documents)
        self.source_indices = list(self.source_indices) + list(range(start, start + len(documents)))
        self._invalidate_rows()
        if self._index is None:
            self._index = GpuGroup(vectors, self.devices) if self.devices else GpuIndex(vectors, device=self.device)
        else:
            self._index.append(vectors)

    def add_document(self, document, vectors=None, count=1, add_timestamp=False):
        """One document and its vector (hyperdb.py:568-626); the chunked-text form (count > 1) belongs to the
        embedding pipeline and is not built."""
        if count != 1:
            raise NotImplementedError("count > 1 (several chunk vectors per document) needs the text pipeline")
        self.add([document], vectors, add_timestamp)

    def add_documents(self, documents, vectors=None, add_timestamp=False):
        """Several documents with one vector each (hyperdb.py:628-689)."""
        self.add(list(documents), vectors, add_timestamp)

    def commit_pending(self):
        """The reference batches additions until this call (hyperdb.py:496-546); here every add is already in HBM."""
        return None

    def dict(self, vectors=False, metadata=None):
        """The documents (optionally with their vectors under ``"vector"``), optionally restricted by a metadata
        ``{key: value}`` dict or ``(key, value)`` tuple (hyperdb.py:444-494)."""
        if not self.documents:
            return []
        keep = np.ones(len(self.documents), dtype=bool)
        if metadata:
            if isinstance(metadata, tuple) and len(metadata) == 2:
                metadata = {metadata[0]: metadata[1]}
            if not isinstance(metadata, dict):
                raise ValueError("metadata must be a dictionary of {key: value} pairs or a tuple of (key, value).")
            keep = self._row_mask([("metadata", metadata)])
        host = self.vectors if vectors else None
        out = []
        for i, doc in enumerate(self.documents):
            if not keep[i]:
                continue
            if host is not None and isinstance(doc, dict):
                doc["vector"] = host[i].tolist()           # in place, like the reference (:487)
            out.append(doc)
        return out

    def _invalidate_rows(self):
        """The document list or the device rows changed: every per-row cache is stale."""
        self._ts_cache.clear()
        self._mask_cache.clear()
        self._ts_dev.clear()
        self._bias_cache.clear()
        self._str_tokens = None
        self.clear_cache()

    def _shards(self):
        """[(GpuIndex, first device row, end device row)] -- one entry, or one per GPU of a GpuGroup."""
        ix = self._index
        if isinstance(ix, GpuGroup):
            return [(sh, lo, hi) for sh, (lo, hi) in zip(ix.shards, ix.bounds)]
        return [(ix, 0, ix.n)]

    # device row of every live document (identity while nothing is tombstoned)
    def _live_rows(self):
        if self._index is None:
            return np.zeros(0, dtype=np.int64)
        if self._dead.size == 0:
            return np.arange(self._index.n, dtype=np.int64)
        alive = np.ones(self._index.n, dtype=bool)
        alive[self._dead] = False
        return np.flatnonzero(alive)

    def _docs_of_rows(self, rows):
        """device rows -> positions in self.documents: every tombstone below a row shifts it down by one."""
        rows = np.asarray(rows, dtype=np.int64)
        return rows if self._dead.size == 0 else rows - np.searchsorted(self._dead, rows)

    def remove_document(self, index):
        """Drop documents by index or list of indices (hyperdb.py:691-766, one vector per document).

        Host lists shrink like the reference's (source_indices renumbered to stay consecutive, :737-745).  The device
        rows are only tombstoned -- they join the row mask of every query, O(1) on the GPU -- until a quarter of the
        stored rows are dead; then the matrix is compacted on the device with its row caches (hdb_index_gather): one
        pass at HBM speed, nothing re-uploaded, no cache rebuilt.  The reference re-stacks the host array on every
        removal (:721-728) and re-normalises all rows on the next query."""
        n_docs = len(self.documents)
        drop = sorted({int(i) % n_docs if -n_docs <= int(i) < 0 else int(i) for i in np.atleast_1d(np.asarray(index)).tolist()})
        if drop and (drop[0] < 0 or drop[-1] >= n_docs):
            raise IndexError("pop index out of range")                  # what self.documents.pop(idx) raises (:717)
        if not drop:
            return
        gone = set(drop)
        drops_a_nan = False
        if self._index is not None:
            dropped_rows = self._live_rows()[np.asarray(drop, dtype=np.int64)]
            # only a removal that takes a NaN row away can clear the matrix's NaN flag: look at the dropped rows themselves
            # (O(dropped x d) on the device) instead of compacting the whole matrix on every removal while the flag is up
            drops_a_nan = self._index.has_nan and self._rows_have_nan(dropped_rows)
            self._dead = np.union1d(self._dead, dropped_rows)
        self.documents = [d for i, d in enumerate(self.documents) if i not in gone]
        kept_src = [s_ for s_ in self.source_indices if s_ not in gone]     # :737-745
        drop_arr = np.asarray(drop)
        self.source_indices = [int(s_ - np.searchsorted(drop_arr, s_)) for s_ in kept_src]
        self._invalidate_rows()
        if self._index is None:
            return
        if not self.documents:
            self._index.close()
            self._index, self._dead = None, np.zeros(0, dtype=np.int64)
        elif self._dead.size * 4 > self._index.n or drops_a_nan:
            # (a stored NaN: the library's flag covers tombstoned rows too, and the reference answers normally as soon as the
            # NaN document is gone -- compact at once, the gather recomputes the flag over the rows it keeps)
            self._index.compact(self._live_rows())
            self._dead = np.zeros(0, dtype=np.int64)

    def _rows_have_nan(self, rows):
        """Does any of the device rows `rows` hold a NaN?  (lifecycle helper of remove_document, not on the query path)"""
        rows = np.asarray(rows, dtype=np.int64)
        for sh, lo, hi in self._shards():
            mine = rows[(rows >= lo) & (rows < hi)] - lo
            if mine.size and bool(torch.isnan(sh.V[torch.from_numpy(mine).to(sh.device)]).any().item()):
                return True
        return False

    def invalidate_rows(self):
        """Call after mutating stored documents IN PLACE (metadata values, timestamps): the device row masks of filters and
        the recency biases are cached per filter / (key, filter, recency_bias) until the document list changes through
        add / remove_document, whereas the reference re-evaluates filters and timestamps on every uncached query
        (hyperdb.py:1492-1493, :1555).  Also clears the LRU query cache, as add / remove do (:566, :766)."""
        self._invalidate_rows()

    @property
    def vectors(self):
        """The stored matrix as a host array, copied back from HBM on demand (the reference keeps it in self.vectors;
        here the only resident copy is the device one)."""
        if self._index is None:
            return None
        host = self._index.host_matrix()
        return host if self._dead.size == 0 else host[self._live_rows()]

    def size(self, with_chunks=False, metadata=None):
        """Number of documents, optionally only those matching a metadata filter (hyperdb.py:410-442); one vector per
        document here, so ``with_chunks`` changes nothing."""
        if metadata:
            return len(self.dict(metadata=metadata))
        return len(self.documents)

    # ---------------------------------------------------------------- LRU cache (hyperdb.py:1368-1427)
    def clear_cache(self):
        self._cache.clear()
        self.cache_hits = self.cache_misses = 0

    def get_cache_size_and_info(self):
        return {'cache_info': {'hits': self.cache_hits, 'misses': self.cache_misses, 'maxsize': self._cache_size,
                               'currsize': len(self._cache)}}

    @staticmethod
    def _hashable_key(query_input, top_k, return_similarities, filters, recency_bias, timestamp_key, metric, ann_percent):
        if isinstance(query_input, np.ndarray):               # the array's bytes (1 us) instead of a tuple of its values (8 us at d=384)
            query_input = (query_input.tobytes(), query_input.dtype.str, query_input.shape)
        elif isinstance(query_input, list):
            query_input = tuple(np.asarray(query_input).ravel().tolist())
        return (query_input, top_k, return_similarities, HyperDB._filter_key(filters), recency_bias, timestamp_key, metric, ann_percent)

    @staticmethod
    def _filter_key(filters):
        def freeze(p):
            if isinstance(p, dict):
                return tuple(sorted(p.items()))
            if isinstance(p, np.ndarray):                       # the additive ("mask", bool_array) filter
                return (p.shape, p.dtype.str, p.tobytes())
            if isinstance(p, (list, tuple)):
                return tuple(p)
            return p
        return None if not filters else tuple((name, freeze(p)) for name, p in filters)

    # ---------------------------------------------------------------- helpers around the ranking call
    @staticmethod
    def get_nested_value(dictionary, keys):
        value = dictionary
        for key in keys:
            if isinstance(value, dict):
                value = value.get(key)
            else:
                return None
        return value

    def _timestamps(self, timestamp_key):
        """float64 timestamps of all documents under ``timestamp_key`` (NaN where missing), extracted once per key and
        reused until the document list changes (the reference walks every document on every query, hyperdb.py:1333-1334)."""
        ts = self._ts_cache.get(timestamp_key)
        if ts is None:
            nested = timestamp_key.split('.') if '.' in timestamp_key else [timestamp_key]
            vals = [self.get_nested_value(doc, nested) for doc in self.documents]
            ts = np.array([np.nan if v is None else v for v in vals], dtype=float)
            self._ts_cache[timestamp_key] = ts
        return ts

    def _handle_timestamps(self, recency_bias, timestamp_key, keep=None):
        """First application of the decay over the FILTERED documents: rb * exp(ts - max ts) (hyperdb.py:1310-1346,
        called with filtered_documents at :1555).  Returns one value per kept document, or None."""
        if recency_bias == 0:
            return None
        if timestamp_key is None:
            timestamp_key = "timestamp"
        if timestamp_key not in self.metadata_keys:
            raise ValueError(f"The timestamp_key '{timestamp_key}' must be present in metadata_keys when recency_bias is not 0.")
        ts = self._timestamps(timestamp_key)
        if keep is not None:
            ts = ts[keep]
        if np.isnan(ts).any():
            raise ValueError("All timestamps must be populated when recency_bias is not 0 or timestamp_key is provided.")
        return recency_bias * np.exp(-np.max(ts) + ts)

    def _query_vectors(self, query_input):
        """hyperdb.py:1178-1216: strings are embedded, arrays validated; returns (nq, d) float array."""
        if isinstance(query_input, str):
            q = np.asarray(self._embed([query_input]))
        elif isinstance(query_input, (list, np.ndarray, tuple)):
            q = np.array(query_input)
            if q.dtype.kind not in "fiuc":                     # (np.issubdtype(q.dtype, np.number) at a tenth of its cost)
                raise ValueError("Numeric array-like query_input expected.")
        else:
            raise ValueError("query_input must be either a string or a numeric array-like object.")
        if q.ndim > 2:
            raise ValueError("query_input must be a 1D or 2D array.")
        if q.ndim == 1:
            q = q.reshape(1, -1)
        if q.size == 0:
            raise ValueError("The generated query vector is empty.")
        if q.shape[1] != self._index.d:
            raise ValueError(f"The dimension of the query_vector ({q.shape[1]}) must match the dimension of the vectors "
                             f"in the database ({self._index.d}).")
        return q

    def _row_mask(self, filters):
        """skip_doc (hyperdb.py:1119-1134), metadata equality and caller-supplied ``mask`` filters -> boolean array
        over self.documents, or None."""
        if not filters:
            return None
        n = len(self.documents)
        keep = np.ones(n, dtype=bool)
        for name, params in filters:
            if name == 'skip_doc':
                if abs(params) >= n:
                    print(f"The absolute value of skip_doc ({abs(params)}) is equal or greater than the total number of documents ({n}).")
                    raise Exception("The absolute value of skip_doc is equal or greater than the total number of documents")
                if params > 0:
                    keep[:params] = False
                elif params < 0:
                    keep[n + params:] = False
            elif name == 'metadata':
                for key, want in params.items():
                    nested = key.split('.')
                    keep &= np.array([self.get_nested_value(doc, nested) == want if isinstance(doc, dict) else False
                                      for doc in self.documents])
            elif name == 'mask':                                                # additive: the caller's own predicate
                m = np.asarray(params, dtype=bool).reshape(-1)
                if m.size != n:
                    raise ValueError(f"mask filter needs one boolean per document ({n}), got {m.size}")
                keep &= m
            elif name == 'sentence':
                # whole-word text search (hyperdb.py:1136-1176): a document passes a filter when ONE string inside it holds
                # every word of the filter; several filters must all pass.  The strings of every document are tokenised once
                # (until the document list changes) instead of once per query.
                wanted = [self.tokenize(w) for w in (params if isinstance(params, (list, tuple)) else [params])]
                if self._str_tokens is None:
                    self._str_tokens = [self._string_token_sets(doc) for doc in self.documents]
                keep &= np.fromiter((all(any(w <= toks for toks in sets) for w in wanted) for sets in self._str_tokens),
                                    dtype=bool, count=n)
            elif name == 'key':
                raise NotImplementedError("filter 'key' re-embeds sub-documents (hyperdb.py:1087): needs the embedding "
                                          "pipeline, which is out of scope here")
            else:
                raise ValueError(f"Invalid filter name {name}")                # hyperdb.py:1279-1280
        return keep

    _PUNCT = str.maketrans('', '', string.punctuation)
    _WORD = re.compile(r'\w+')

    @classmethod
    def tokenize(cls, text):
        """The set of lower-cased words of a string, punctuation dropped first (what hyperdb.py:1136-1141 compares)."""
        return set(cls._WORD.findall(text.translate(cls._PUNCT).lower()))

    @classmethod
    def _string_token_sets(cls, doc):
        """Token sets of every string reachable inside a document through dicts and lists (explicit stack, no recursion
        limit); other leaves never match (hyperdb.py:1143-1158)."""
        out, todo = [], [doc]
        while todo:
            obj = todo.pop()
            if isinstance(obj, str):
                out.append(cls.tokenize(obj))
            elif isinstance(obj, dict):
                todo.extend(obj.values())
            elif isinstance(obj, list):
                todo.extend(obj)
        return out

    # ---------------------------------------------------------------- persistence (hyperdb.py:769-1005)
    def _data_dict(self):
        return {"vectors": [v.tolist() for v in self.vectors], "documents": self.documents,
                "source_indices": self.source_indices, "split_info": getattr(self, "split_info", {}),
                "metadata_index": getattr(self, "_metadata_index", {}),
                "vectors_normalized": getattr(self, "vectors_normalized", False)}

    def save(self, storage_file, format='pickle', save_ann_index=True):
        """Write the database in the reference's own layout (hyperdb.py:769-899): a dict with the keys ``vectors``
        (list of lists), ``documents``, ``source_indices``, ``split_info``, ``metadata_index``, ``vectors_normalized`` as
        pickle (gzip when the name ends in .gz), json, or the sqlite tables of ``_save_sqlite``.  There is no ANN index
        to save (``save_ann_index`` is accepted and ignored)."""
        if self.vectors is None or len(self.vectors) == 0 or not self.documents:
            print("Nothing to save. Exit.")
            return
        data = self._data_dict()
        if format == 'pickle':
            opener = gzip.open if str(storage_file).endswith(".gz") else open
            with opener(storage_file, "wb") as f:
                pickle.dump(data, f)
        elif format == 'json':
            with open(storage_file, "w") as f:
                json.dump(data, f)
        elif format == 'sqlite':
            with closing(sqlite3.connect(storage_file)) as conn:
                cur = conn.cursor()
                cur.execute('CREATE TABLE IF NOT EXISTS documents (id INTEGER PRIMARY KEY, data TEXT)')
                cur.execute('CREATE TABLE IF NOT EXISTS vectors (id INTEGER PRIMARY KEY, document_id INTEGER, vector BLOB)')
                cur.execute('CREATE TABLE IF NOT EXISTS source_indices (id INTEGER PRIMARY KEY, value INTEGER)')
                cur.execute('CREATE TABLE IF NOT EXISTS split_info (id INTEGER PRIMARY KEY, value TEXT)')
                cur.execute('CREATE TABLE IF NOT EXISTS metadata_index (key TEXT PRIMARY KEY, value TEXT)')
                cur.execute('CREATE TABLE IF NOT EXISTS settings (name TEXT PRIMARY KEY, value TEXT)')
                cur.executemany('INSERT INTO documents (data) VALUES (?)', [(json.dumps(d),) for d in data["documents"]])
                first = cur.lastrowid - len(data["documents"]) + 1
                cur.executemany('INSERT INTO vectors (document_id, vector) VALUES (?, ?)',
                                [(first + i, json.dumps(v)) for i, v in enumerate(data["vectors"])])
                cur.executemany('INSERT INTO source_indices (value) VALUES (?)', [(int(i),) for i in data["source_indices"]])
                cur.execute('INSERT INTO split_info (value) VALUES (?)', (json.dumps(data["split_info"]),))
                cur.executemany('INSERT INTO metadata_index (key, value) VALUES (?, ?)',
                                [(k, json.dumps(v)) for k, v in data["metadata_index"].items()])
                cur.execute('INSERT OR REPLACE INTO settings (name, value) VALUES (?, ?)',
                            ('vectors_normalized', json.dumps(data["vectors_normalized"])))
                conn.commit()
        else:
            raise ValueError(f"Unsupported format '{format}'")

    def load(self, storage_file, format='pickle', load_ann_index=True, preload_ann_into_memory=False):
        """Read a database file written by the reference (or by ``save``) and put its matrix in HBM in ``fp_precision``
        (hyperdb.py:901-1005).  Only load pickles you trust.  The ``.ann`` side file, if any, is not used (exact search)."""
        if format == 'pickle':
            try:
                with gzip.open(storage_file, "rb") as f:
                    data = pickle.load(f)
            except OSError:
                with open(storage_file, "rb") as f:
                    data = pickle.load(f)
        elif format == 'json':
            with open(storage_file, "r") as f:
                data = json.load(f)
        elif format == 'sqlite':
            with closing(sqlite3.connect(storage_file)) as conn:
                cur = conn.cursor()
                data = {"documents": [json.loads(r[0]) for r in cur.execute('SELECT data FROM documents')],
                        "vectors": [json.loads(r[0]) for r in cur.execute('SELECT vector FROM vectors ORDER BY document_id')],
                        "source_indices": [r[0] for r in cur.execute('SELECT value FROM source_indices')],
                        "split_info": {}, "metadata_index": {}, "vectors_normalized": False}
                for r in cur.execute('SELECT value FROM split_info'):
                    data["split_info"] = json.loads(r[0])
                for r in cur.execute('SELECT key, value FROM metadata_index'):
                    data["metadata_index"][r[0]] = json.loads(r[1])
                for r in cur.execute('SELECT value FROM settings WHERE name = ?', ('vectors_normalized',)):
                    data["vectors_normalized"] = json.loads(r[0])
        else:
            raise ValueError(f"Unsupported format '{format}'")
        vectors = np.array(data["vectors"], dtype=self.fp_precision)
        if self._index is not None:
            self._index.close()
        self._index, self._dead = None, np.zeros(0, dtype=np.int64)
        self.documents, self.source_indices = [], []
        self._invalidate_rows()
        documents = list(data["documents"])
        if len(documents):
            self.add(documents, vectors)
            stored = list(data.get("source_indices", []))
            if len(stored) == len(documents):
                self.source_indices = stored
        self.split_info = data.get("split_info", {})
        self._metadata_index = data.get("metadata_index", {})
        self.vectors_normalized = data.get("vectors_normalized", False)

    # ---------------------------------------------------------------- query (hyperdb.py:1429-1586)
    # Resident inputs of the ranking call.  Filters and tombstones become ONE row mask per shard, kept on the device per filter
    # key; the timestamps of a key live on the device as a float64 column over the device rows; both decays of a query
    # (hyperdb.py:1344 over the FILTERED documents, then ranking_algorithm.py:183) are one kernel over that column and the
    # mask, and its result is kept per (key, filter, recency_bias).  A query whose inputs are cached builds no O(N) host array
    # and uploads nothing; everything is dropped when the document list changes (_invalidate_rows).
    _ROW_CACHE_SLOTS = 4

    def _mask_for(self, filters):
        """-> (per-shard device masks or None, number of documents kept, bool array over documents or None)."""
        fkey = self._filter_key(filters)
        hit = self._mask_cache.get(fkey)
        if hit is not None:
            self._mask_cache.move_to_end(fkey)
            return hit
        keep = self._row_mask(filters)                          # over documents (host predicates, once per filter key)
        if keep is None and self._dead.size == 0:
            hit = (None, len(self.documents), None)
        else:
            self.host_row_passes += 1
            live = self._live_rows()
            rows_kept = live if keep is None else live[keep]
            row_mask = np.zeros(self._index.n, dtype=np.uint8)
            row_mask[rows_kept] = 1
            parts = [torch.from_numpy(row_mask[lo:hi]).to(sh.device) for sh, lo, hi in self._shards()]
            hit = (parts, int(rows_kept.size), keep)
        self._mask_cache[fkey] = hit
        while len(self._mask_cache) > self._ROW_CACHE_SLOTS:
            self._mask_cache.popitem(last=False)
        return hit

    def _bias_for(self, recency_bias, timestamp_key, filters, masks, keep):
        """Per-shard device bias of both decays, or None when recency_bias == 0 (hyperdb.py:1320-1322)."""
        if recency_bias == 0:
            return None
        if timestamp_key is None:
            timestamp_key = "timestamp"
        if timestamp_key not in self.metadata_keys:
            raise ValueError(f"The timestamp_key '{timestamp_key}' must be present in metadata_keys when recency_bias is not 0.")
        bkey = (timestamp_key, self._filter_key(filters), float(recency_bias))
        hit = self._bias_cache.get(bkey)
        if hit is not None:
            self._bias_cache.move_to_end(bkey)
            return hit
        self.host_row_passes += 1
        ts = self._timestamps(timestamp_key)                    # over documents (host, extracted once per key)
        kept = ts if keep is None else ts[keep]
        if np.isnan(kept).any():
            raise ValueError("All timestamps must be populated when recency_bias is not 0 or timestamp_key is provided.")
        cols = self._ts_dev.get(timestamp_key)
        if cols is None:                                        # documents -> device rows (tombstoned rows keep a 0)
            rows = np.zeros(self._index.n, dtype=np.float64)
            rows[self._live_rows()] = np.nan_to_num(ts)
            cols = self._ts_dev[timestamp_key] = [torch.from_numpy(rows[lo:hi]).to(sh.device) for sh, lo, hi in self._shards()]
        t_max, t_min = float(np.max(kept)), float(np.min(kept))
        hit = [sh.recency_twice(cols[p], None if masks is None else masks[p], recency_bias, t_max, t_min) if sh.n else None
               for p, (sh, lo, hi) in enumerate(self._shards())]
        # The decay kernels run on torch's current stream of each shard's device; a shard group (devices=[...]) queries on
        # private non-blocking streams that do not order against it.  Once per cache miss, never per query: wait here.
        for sh, lo, hi in self._shards():
            if sh.n:
                torch.cuda.current_stream(sh.device).synchronize()
        self._bias_cache[bkey] = hit
        while len(self._bias_cache) > self._ROW_CACHE_SLOTS:
            self._bias_cache.popitem(last=False)
        return hit

    def _execute(self, Q, top_k, return_similarities, filters, recency_bias, timestamp_key, metric):
        if self._index is None or not self.documents:
            raise Exception("The database is empty. Cannot proceed with the query.")
        if metric not in _METRICS:
            raise ValueError(f"Invalid metric '{metric}'. Supported: " + ", ".join(f"'{m}'" for m in _METRICS))
        ranking._validate_metric(metric)
        ix = self._index
        # (one query: a NaN anywhere makes q.q a NaN -- a dot product instead of an isnan pass and a reduction)
        if ix.has_nan or (math.isnan(float(np.dot(Q[0], Q[0]))) if len(Q) == 1 and Q.dtype.kind == "f" else bool(np.isnan(Q).any())):
            raise ValueError(ranking.NAN_MESSAGE)
        masks, n_avail, keep = self._mask_for(filters)
        if n_avail == 0:
            print("INFO: No document matches your query with the brute-force method and the current filters.")
            return [[] for _ in range(len(Q))]
        if top_k > n_avail:
            print(f"Warning: top_k ({top_k}) is greater than the number of filtered documents ({n_avail}). Setting top_k to {n_avail}.")
            top_k = n_avail
        bias = self._bias_for(recency_bias, timestamp_key, filters, masks, keep)
        shards = self._shards()
        try:
            for p, (sh, lo, hi) in enumerate(shards):
                if sh.n:
                    sh.set_row_mask(None if masks is None else masks[p])
                    sh.set_bias(None if bias is None else bias[p])
            idx, sc = ix.topk(Q, int(top_k), METRIC_IDS[metric])
        finally:
            for sh, lo, hi in shards:
                sh.set_row_mask(None)
                sh.set_bias(None)
        if n_avail == 1:
            print("Info: Only one document left.")                               # ranking_algorithm.py:189-191
        out = []
        documents, sources = self.documents, self.source_indices
        for qi in range(len(Q)):
            rows = idx[qi]
            if len(rows) and rows[-1] < 0:                        # fewer than k rows exist: the tail is padded with -1
                rows = rows[rows >= 0]
            docs = self._docs_of_rows(rows).tolist()
            if not return_similarities:
                out.append([documents[r] for r in docs])
            elif n_avail == 1:                                    # one row left: the reference's scores come back 2-D (:191)
                out.append([(documents[r], np.array([float(sc[qi][j])]), sources[r]) for j, r in enumerate(docs)])
            elif len(docs) > 1:                                   # C-level gathers instead of a Python loop over the k results
                pick = itemgetter(*docs)
                out.append(list(zip(pick(documents), sc[qi].tolist(), pick(sources))))
            else:
                out.append([(documents[r], s_, sources[r]) for r, s_ in zip(docs, sc[qi].tolist())])
        return out

    def query(self, query_input, top_k=5, return_similarities=True, filters=None, recency_bias=0, timestamp_key=None,
              metric='cosine_similarity', ann_percent=5):
        """Same signature and return shape as reference HyperDB.query (hyperdb.py:1584): a list of
        ``(document, score, source_index)`` tuples (or documents) for ONE query."""
        key = self._hashable_key(query_input, top_k, return_similarities, filters, recency_bias, timestamp_key, metric, ann_percent)
        if key in self._cache:
            self.cache_hits += 1
            self._cache.move_to_end(key)
            return self._cache[key]
        self.cache_misses += 1
        try:
            if self._index is None or not self.documents:
                raise Exception("The database is empty. Cannot proceed with the query.")
            Q = self._query_vectors(query_input)
            if len(Q) != 1:
                raise ValueError("query() takes one query; use query_batch() for a (Q, d) batch.")
            result = self._execute(Q, top_k, return_similarities, filters, recency_bias, timestamp_key, metric)[0]
        except (ValueError, TypeError) as e:
            print(f"An exception occurred due to invalid input: {e}")
            raise
        self._cache[key] = result
        if len(self._cache) > self._cache_size:
            self._cache.popitem(last=False)
        return result

    def query_batch(self, query_inputs, top_k=5, return_similarities=True, filters=None, recency_bias=0, timestamp_key=None,
                    metric='cosine_similarity'):
        """Additive: one list of results per row of a (Q, d) batch, computed in a single pass over the matrix."""
        Q = self._query_vectors(query_inputs)
        return self._execute(Q, top_k, return_similarities, filters, recency_bias, timestamp_key, metric)
