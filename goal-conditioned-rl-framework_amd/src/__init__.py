"""Host-side mirror of the reference's src/ package for the hot path (see ../__init__.py)."""
