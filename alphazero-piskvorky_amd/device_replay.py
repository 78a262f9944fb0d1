"""Device-resident replay ring: packed self-play records stay in HBM between the episode-end
gather and train_step; a batch is sampled and unpacked (encode + symmetry) on the device.

Semantics follow the reference's ReplayBuffer (replay_buffer.py:14-39): FIFO with a capacity counted in EXAMPLES
(each position contributes `aug` examples, self_play.py:146-148), uniform sampling without replacement.
"""
import numpy as np
import torch

from ._capi import AZ_AUG_REFERENCE4


class DeviceReplayBuffer:
    def __init__(self, engine, capacity=40_000, aug=AZ_AUG_REFERENCE4, device="cuda:0", seed=None):
        self.engine = engine
        self.aug = aug
        self.rb = self.engine.record_bytes
        self.n = self.engine.n
        self.cap = max(1, capacity // aug)               # positions
        self.device = torch.device(device)
        self.ring = torch.zeros(self.cap * self.rb, dtype=torch.uint8, device=self.device)
        self.head = 0                                    # next write position
        self.count = 0
        self.rng = np.random.default_rng(seed)

    def __len__(self):
        return self.count * self.aug

    def extend_packed(self, packed, records):
        """Appends `records` packed records (uint8 device tensor); the oldest are overwritten (deque(maxlen))."""
        records = int(records)
        if records > self.cap:                           # only the newest `cap` survive
            packed = packed[(records - self.cap) * self.rb:]
            records = self.cap
        first = min(records, self.cap - self.head)
        self.ring[self.head * self.rb:(self.head + first) * self.rb] = packed[:first * self.rb]
        if records > first:
            self.ring[:(records - first) * self.rb] = packed[first * self.rb:records * self.rb]
        self.head = (self.head + records) % self.cap
        self.count = min(self.cap, self.count + records)

    def sample_batch(self, batch_size):
        """Returns (states f32[B,4,n,n], pis f32[B,n,n], z f32[B]) on the device."""
        total = self.count * self.aug
        b = min(int(batch_size), total)
        ex = self.rng.choice(total, size=b, replace=False)            # example ids, like random.sample over examples
        start = (self.head - self.count) % self.cap                   # oldest record
        idx = torch.as_tensor(((ex // self.aug) + start) % self.cap, dtype=torch.int64, device=self.device)
        sym = torch.as_tensor(ex % self.aug, dtype=torch.int32, device=self.device)
        n = self.n
        states = torch.empty((b, 4, n, n), dtype=torch.float32, device=self.device)
        pis = torch.empty((b, n, n), dtype=torch.float32, device=self.device)
        zs = torch.empty(b, dtype=torch.float32, device=self.device)
        self.engine.examples_gather(self.ring.data_ptr(), idx.data_ptr(), sym.data_ptr(), b,
                                    1 if self.aug == AZ_AUG_REFERENCE4 else 0, states.data_ptr(), pis.data_ptr(), zs.data_ptr())
        return states, pis, zs

    def export_examples(self):
        """All examples currently in the ring as the reference's tuples (state CPU tensor, pi ndarray, int z), oldest
        first -- e.g. to write a reference-compatible buffer.pkl via ReplayBuffer.extend(...).save(path)."""
        n, aug = self.n, self.aug
        start = (self.head - self.count) % self.cap
        order = torch.as_tensor((np.arange(self.count) + start) % self.cap, dtype=torch.int64, device=self.device)
        idx = order.repeat_interleave(aug)
        sym = torch.arange(aug, dtype=torch.int32, device=self.device).repeat(self.count)
        total = self.count * aug
        states = torch.empty((total, 4, n, n), dtype=torch.float32, device=self.device)
        pis = torch.empty((total, n, n), dtype=torch.float32, device=self.device)
        zs = torch.empty(total, dtype=torch.float32, device=self.device)
        if total:
            self.engine.examples_gather(self.ring.data_ptr(), idx.data_ptr(), sym.data_ptr(), total,
                                        1 if aug == AZ_AUG_REFERENCE4 else 0, states.data_ptr(), pis.data_ptr(), zs.data_ptr())
        states, pis, zs = states.cpu(), pis.cpu().numpy(), zs.cpu().numpy().astype(np.int64)
        return list(zip(states.unbind(0), list(pis), zs.tolist()))
