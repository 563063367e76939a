"""CSV reader for the seven split files (reference: util/loader.py:21-33), array based."""
import numpy as np


class DataLoader(object):
    @staticmethod
    def load_data_set(file):
        """``user,item[,...]`` with one header line -> ``[[int user, int item, 1.0], ...]``."""
        arr = np.loadtxt(file, delimiter=',', skiprows=1, usecols=(0, 1), dtype=np.int64, ndmin=2)
        return [[int(u), int(i), 1.0] for u, i in arr.tolist()]

    @staticmethod
    def load_pairs(file) -> np.ndarray:
        """Same file as an (n, 2) int64 array (what ColdStartDataBuilder consumes fastest)."""
        return np.loadtxt(file, delimiter=',', skiprows=1, usecols=(0, 1), dtype=np.int64, ndmin=2)
