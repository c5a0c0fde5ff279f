from .spliter import (BREAKPOINT_DEFAULTS, SemanticChunker, calculate_cosine_distances, combine_sentences,
                      cosine_similarity, device_cosine_distances)

__all__ = ["BREAKPOINT_DEFAULTS", "SemanticChunker", "calculate_cosine_distances", "combine_sentences",
           "cosine_similarity", "device_cosine_distances"]
