"""The one arithmetic step of the reference's graph store that sits on the dense hot path's kernels: entity de-duplication by
all-pairs cosine (encapsulation/database/graph_db/Base_Neo4j.py:538-583).  The graph database itself is out of scope."""
from .similarity import similar_pairs

__all__ = ["similar_pairs"]
