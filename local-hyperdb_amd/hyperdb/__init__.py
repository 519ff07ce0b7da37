"""MI355X-native drop-in for the brute-force ranking path of AdamCodd/local-hyperDB.

``import hyperdb.ranking_algorithm as ranking`` and ``from hyperdb import HyperDB`` keep the
reference's spelling (reference hyperdb/__init__.py is ``from .hyperdb import *``).
"""
from . import ranking_algorithm  # noqa: F401
from .hyperdb import *  # noqa: F401,F403
