"""Device-resident ring buffer with the sampling semantics of the reference's
``ReplayBuffer`` (learning/replay_buffer.py:3-84 under /root/reference/ase): sequential window over a
random permutation of the slots, ``% head`` while the ring is still filling, permutation redrawn
when the window wraps.  ``sample`` returns INDICES (int32, device): the rows themselves are gathered
by the normaliser kernel, so the 131072 x 1400 copies the reference makes per epoch never exist."""
import torch


class ReplayBuffer:
    def __init__(self, buffer_size, device, backend=None, generator=None):
        self._head = 0
        self._total_count = 0
        self._buffer_size = int(buffer_size)
        self._device = device
        self._data = None
        self._be = backend
        self._gen = generator            # shared by everything an agent draws: data-parallel ranks stay in lock-step
        self._sample_idx = torch.randperm(self._buffer_size, device=device, generator=generator)
        self._sample_head = 0

    def reset(self):
        self._head = 0
        self._total_count = 0
        self._reset_sample_idx()

    def get_buffer_size(self):
        return self._buffer_size

    def get_total_count(self):
        return self._total_count

    @property
    def data(self):
        return self._data

    def _ensure(self, width):
        if self._data is None:
            self._data = torch.zeros(self._buffer_size, width, dtype=torch.float32, device=self._device)

    def store(self, src, n=None, idx=None, remap=(0, 0)):
        """Append n rows of src (optionally through a row map: idx / time-major remap)."""
        n = int(src.shape[0] if n is None else n)
        if n == 0:
            return
        assert n <= self._buffer_size
        src2 = src.view(-1, src.shape[-1])
        self._ensure(src2.shape[1])
        self._be.ring_store(src2, src2.shape[1], idx, remap, n, self._data, self._buffer_size, self._head)
        self._head = (self._head + n) % self._buffer_size
        self._total_count += n

    def sample_indices(self, n):
        size = self._buffer_size
        idx = torch.arange(self._sample_head, self._sample_head + n, device=self._device) % size
        rand_idx = self._sample_idx[idx]
        if self._total_count < size:
            rand_idx = rand_idx % self._head
        self._sample_head += n
        if self._sample_head >= size:
            self._reset_sample_idx()
        return rand_idx.to(torch.int32)

    def sample(self, n):
        """Reference-compatible form (materialises the rows)."""
        return {'amp_obs': self._data[self.sample_indices(n).long()]}

    def _reset_sample_idx(self):
        self._sample_idx[:] = torch.randperm(self._buffer_size, device=self._device, generator=self._gen)
        self._sample_head = 0
