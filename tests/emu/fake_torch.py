"""tests/emu/fake_torch.py -- TEST INFRASTRUCTURE ONLY: the handful of torch calls tests/test_demod_gpu.py makes, on numpy arrays,
so that the very same test functions can drive the demodulator's host twin (tests/test_demod_gpu_on_twin_cpu.py). "Device"
pointers are host pointers here."""
import numpy as np

float32, int8, uint8, int16, int32 = np.float32, np.int8, np.uint8, np.int16, np.int32


class FakeTensor:
    def __init__(self, a):
        self.a = a

    def cuda(self):
        return self

    def cpu(self):
        return self

    def contiguous(self):
        return FakeTensor(np.ascontiguousarray(self.a))

    def numpy(self):
        return self.a.copy()

    def data_ptr(self):
        return self.a.ctypes.data

    def numel(self):
        return self.a.size

    def __getitem__(self, k):
        return FakeTensor(self.a[k])

    def __mul__(self, v):
        return FakeTensor((self.a * np.float32(v)).astype(self.a.dtype))

    def __len__(self):
        return len(self.a)


def from_numpy(a):
    return FakeTensor(np.ascontiguousarray(a))


def zeros(shape, dtype=float32, device=None):
    return FakeTensor(np.zeros(shape, dtype=dtype))


def empty(shape, dtype=float32, device=None):
    return FakeTensor(np.zeros(shape, dtype=dtype))


class Generator:
    def __init__(self, device=None):
        self.rng = np.random.default_rng(0)

    def manual_seed(self, s):
        self.rng = np.random.default_rng(s)


def randn(n, device=None, generator=None):
    return FakeTensor(generator.rng.standard_normal(n).astype(np.float32))


def device(*a):
    return "host"


class cuda:
    @staticmethod
    def is_available():
        return True

    @staticmethod
    def synchronize():
        pass
