"""rl_games.common.datasets.PPODataset (1.1.4): the part AMPDataset builds on (amp_datasets.py:4-7)."""


class PPODataset:
    def __init__(self, batch_size, minibatch_size, is_discrete, is_rnn, device, seq_len):
        self.is_rnn = is_rnn
        self.seq_len = seq_len
        self.batch_size = batch_size
        self.minibatch_size = minibatch_size
        self.device = device
        self.length = self.batch_size // self.minibatch_size
        self.is_discrete = is_discrete
        self.is_continuous = not is_discrete
        self.special_names = ['rnn_states']

    def update_values_dict(self, values_dict):
        self.values_dict = values_dict

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return self._get_item(idx)
