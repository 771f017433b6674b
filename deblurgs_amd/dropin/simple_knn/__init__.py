"""Drop-in shim for the reference's `simple_knn` package (see _C.py)."""
