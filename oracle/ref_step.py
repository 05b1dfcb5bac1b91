# -*- coding: utf-8 -*-
"""TEST INFRASTRUCTURE.  One training step of THE REFERENCE'S OWN MODULE on the CPU.

Loads ``oracle/_ref/wavenet.py`` -- the reference's ``wavenet_vocoder/nets/wavenet.py``, copied there by
``oracle/build_ref.py`` (git-ignored) -- and drives it with the 13 lines of the reference's training loop
(``wavenet_vocoder/bin/train.py:527-540``; train.py itself exits without CUDA, :516-525).  Used only by
``bench.py``'s ``cpu_baseline`` leg (so that the stated CPU baseline is the reference itself, ``kind: "reference"``) and by
tests that pin the restatement in ``wavenet_oracle.py`` against it.  Never imported by the product.
"""
import importlib.util
import os

import torch

REF_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "wavenet.py")


def available():
    return os.path.exists(REF_FILE)


def load_reference():
    """The reference's wavenet module (WaveNet, initialize, encode_mu_law, ...), or None when the copy is absent."""
    if not available():
        return None
    spec = importlib.util.spec_from_file_location("_reference_wavenet", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class ReferenceTrainer(object):
    """Reference ``WaveNet`` + ``nn.CrossEntropyLoss`` + ``torch.optim.Adam`` (train.py:440-461), one ``step`` = train.py:527-540."""

    def __init__(self, cfg_tuple, state=None, lr=1e-4, weight_decay=0.0, seed=1):
        ref = load_reference()
        if ref is None:
            raise RuntimeError("oracle/_ref/wavenet.py is missing: run oracle/build_ref.py where /root/reference exists")
        torch.manual_seed(seed)                                            # train.py:386,421
        self.model = ref.WaveNet(*cfg_tuple)                               # train.py:440-448
        self.model.apply(ref.initialize)                                   # train.py:450
        if state is not None:
            self.model.load_state_dict(state)
        self.model.train()                                                 # train.py:455
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr, weight_decay=weight_decay)   # train.py:457-460
        self.criterion = torch.nn.CrossEntropyLoss()                       # train.py:461
        self.rf = self.model.receptive_field
        self.Q = cfg_tuple[0]

    def step(self, x, h, t):
        batch_output = self.model(x, h)                                    # train.py:533
        batch_loss = self.criterion(
            batch_output[:, self.rf:].contiguous().view(-1, self.Q),
            t[:, self.rf:].contiguous().view(-1))                          # train.py:534-536
        self.optimizer.zero_grad()                                         # train.py:537
        batch_loss.backward()                                              # train.py:538
        self.optimizer.step()                                              # train.py:539
        return batch_loss.item(), batch_output
