"""tensorflow.python package of the eager stub (see ../__init__.py)."""
