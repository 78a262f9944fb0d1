"""FIFO replay buffer with the reference's surface and file format (replay_buffer.py:14-75): a deque(maxlen) of
(state, pi, z) tuples, pickled as a deque, so buffer.pkl files move between the reference and this package."""
import pickle
import random
from collections import deque


class ReplayBuffer:
    def __init__(self, capacity=40_000):
        self.buffer = deque(maxlen=capacity)

    def __len__(self):
        return len(self.buffer)

    def extend(self, game_examples):
        self.buffer.extend(game_examples)

    def sample_batch(self, batch_size):
        if len(self.buffer) < batch_size:
            return self.buffer                       # replay_buffer.py:36-37: everything when there is not enough
        return random.sample(list(self.buffer), batch_size)

    def all(self):
        return list(self.buffer)

    def save(self, filename):
        with open(filename, "wb") as f:
            pickle.dump(self.buffer, f)

    def load(self, filename):
        # a pickle executes code from the file: only load buffers written by this class / your own reference run
        with open(filename, "rb") as f:
            self.buffer = deque(pickle.load(f), maxlen=self.buffer.maxlen)
