"""HyperDB facade for the GPU ranking path (placeholder: filled in by the facade milestone)."""
__all__ = []
