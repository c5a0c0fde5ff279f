"""Golden vectors for the all-pairs cosine: inputs + what the reference's arithmetic returns for them.

    python tests/golden/make_similar_pairs.py        # needs scikit-learn; writes tests/golden/similar_pairs.json

The reference's step (encapsulation/database/graph_db/Base_Neo4j.py:559-566) is a call into scikit-learn —
`cosine_similarity(np.array(embeddings))` — followed by `for i ... for j in range(i + 1, n): if similarity_matrix[i][j] >=
similarity_threshold`.  The method around it needs a live Neo4j session, so it cannot be run here; the vectors are made by
calling the SAME library function (scikit-learn, unpinned in the reference's requirements; the version used is recorded in the
file) and applying that loop.  Only data is written."""
import json
import os

import numpy as np
import sklearn
from sklearn.metrics.pairwise import cosine_similarity

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "similar_pairs.json")


def reference_pairs(embeddings, similarity_threshold):
    embeddings_array = np.array(embeddings)
    similarity_matrix = cosine_similarity(embeddings_array)
    pairs, scores = [], []
    for i in range(len(embeddings)):
        for j in range(i + 1, len(embeddings)):
            similarity_score = similarity_matrix[i][j]
            if similarity_score >= similarity_threshold:
                pairs.append([i, j])
                scores.append(float(similarity_score))
    return pairs, scores


def main():
    rng = np.random.default_rng(2026)
    cases = []

    def add(name, x, thr):
        x = np.asarray(x, dtype=np.float32).astype(np.float64)      # what an fp32 encoder hands the graph store
        pairs, scores = reference_pairs(x.tolist(), thr)
        cases.append({"name": name, "threshold": thr, "embeddings_hex": [[float(v).hex() for v in row] for row in x],
                      "pairs": pairs, "scores_hex": [s.hex() for s in scores]})

    base = rng.standard_normal((60, 48))
    add("near-duplicate entities at several distances, scaled copies, two zero rows",
        np.concatenate([base, base[:20] + 0.05 * rng.standard_normal((20, 48)), base[20:30] + 0.12 * rng.standard_normal((10, 48)),
                        7.5 * base[3:7], np.zeros((2, 48))]), 0.95)
    add("a low threshold: many pairs", rng.standard_normal((40, 6)), 0.5)
    add("nothing similar", rng.standard_normal((50, 96)), 0.95)
    add("exact duplicates: cosine 1 up to rounding", np.repeat(rng.standard_normal((5, 33)), 3, axis=0), 0.999999)
    add("two entities", [[1.0, 2.0, 3.0], [1.0, 2.0, 3.1]], 0.95)
    json.dump({"made_with": f"scikit-learn {sklearn.__version__} cosine_similarity + the i < j loop of Base_Neo4j.py:561-566",
               "cases": cases}, open(OUT, "w"), indent=0)
    print("written", OUT, [len(c["pairs"]) for c in cases])


if __name__ == "__main__":
    main()
