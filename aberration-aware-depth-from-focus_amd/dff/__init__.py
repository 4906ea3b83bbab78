"""Script-compat shims for the consumers of the renderer (reference: dff/).  Only the glue that selects a lens and the
focus distances is mirrored; the DFF networks, metrics and the file-based datasets are out of scope."""
