"""FIFO replay buffer with the reference's surface (replay_buffer.py:14-75): deque(maxlen), random batches."""
import pickle
import random
from collections import deque


class ReplayBuffer:
    def __init__(self, capacity=40_000):
        self.buffer = deque(maxlen=capacity)

    def __len__(self):
        return len(self.buffer)

    def extend(self, examples):
        self.buffer.extend(examples)

    def sample_batch(self, batch_size):
        return random.sample(list(self.buffer), min(batch_size, len(self.buffer)))

    def save(self, path):
        with open(path, "wb") as f:
            pickle.dump(list(self.buffer), f)

    def load(self, path):
        with open(path, "rb") as f:             # only for files this class wrote itself
            self.buffer.extend(pickle.load(f))
