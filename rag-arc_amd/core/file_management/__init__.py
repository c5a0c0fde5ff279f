"""Callers of the embedding provider next to the retrieval path."""
