"""Per-prompt reward statistics -> advantages (ring-buffer variant).

Behavioural mirror of /root/reference/ddpo/utils/stat_tracking.py:5-35: for every distinct prompt keep the last
`buffer_size` rewards; while a prompt has fewer than `min_count` entries normalise with the statistics of the WHOLE
current batch, afterwards with the prompt's own buffer; std always gets +1e-6.
"""
from collections import deque

import numpy as np


class PerPromptStatTracker:
    def __init__(self, buffer_size, min_count):
        self.buffer_size = buffer_size
        self.min_count = min_count
        self.stats = {}

    def update(self, prompts, rewards):
        prompts = np.asarray(prompts)
        rewards = np.asarray(rewards)
        advantages = np.empty_like(rewards)
        batch_mean, batch_std = np.mean(rewards), np.std(rewards) + 1e-6
        for prompt in np.unique(prompts):
            sel = prompts == prompt
            buf = self.stats.setdefault(prompt, deque(maxlen=self.buffer_size))
            buf.extend(rewards[sel])
            if len(buf) < self.min_count:
                mean, std = batch_mean, batch_std
            else:
                mean, std = np.mean(buf), np.std(buf) + 1e-6
            advantages[sel] = (rewards[sel] - mean) / std
        return advantages

    def get_stats(self):
        return {p: {"mean": np.mean(b), "std": np.std(b), "count": len(b)} for p, b in self.stats.items()}

    # not in the reference (it never resumes): make the tracker part of a resumable checkpoint
    def state_dict(self):
        return {str(p): [np.asarray(x).tolist() for x in b] for p, b in self.stats.items()}

    def load_state_dict(self, state):
        self.stats = {p: deque([np.asarray(x) for x in v], maxlen=self.buffer_size) for p, v in state.items()}
