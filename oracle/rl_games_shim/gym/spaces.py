import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is not None:
            low = np.full(shape, low, dtype=dtype)
            high = np.full(shape, high, dtype=dtype)
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = dtype


class Dict(dict):
    pass
