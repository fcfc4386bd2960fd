"""Drop-in for the hot-path parts of the reference's ``src/utils`` (loss, metric, npy2point)."""
