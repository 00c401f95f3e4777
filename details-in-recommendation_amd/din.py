"""din.py -- DIN local-activation pooling module (no reference code: /root/reference/README.md:27 links
arXiv:1706.06978; the unit is defined in include/dir_hip.h A13 / oracle)."""
import math

import torch
from torch import nn

from . import autograd as ag
from . import ops


class DINAttentionPool(nn.Module):
    """history ids [B,T] + lengths [B] + candidate ids [B] -> pooled interest vector [B,K]."""

    def __init__(self, vocab_size, embedding_dim=64, hidden_units=(80, 40), normalize=False):
        super().__init__()
        K, (H1, H2) = embedding_dim, hidden_units
        s = 1.0 / math.sqrt(K)
        self.table = nn.Parameter(nn.init.trunc_normal_(torch.empty(vocab_size, K), std=s, a=-2 * s, b=2 * s))
        self.W1 = nn.Parameter(nn.init.xavier_uniform_(torch.empty(4 * K, H1)))
        self.b1 = nn.Parameter(torch.zeros(H1))
        self.W2 = nn.Parameter(nn.init.xavier_uniform_(torch.empty(H1, H2)))
        self.b2 = nn.Parameter(torch.zeros(H2))
        self.W3 = nn.Parameter(nn.init.xavier_uniform_(torch.empty(H2, 1)).reshape(H2))
        self.b3 = nn.Parameter(torch.zeros(1))
        self.normalize = normalize

    def forward(self, hist, hist_len, cand, want_scores=False):
        if torch.is_grad_enabled() and not want_scores and any(p.requires_grad for p in self.parameters()):
            return ag.din_attention_pool(self.table, hist, hist_len, cand, self.W1, self.b1, self.W2, self.b2, self.W3,
                                         self.b3, normalize=self.normalize)       # sparse table gradient
        return ops.din_attention_pool(self.table.data, hist, hist_len, cand, self.W1.data, self.b1.data, self.W2.data,
                                      self.b2.data, self.W3.data, self.b3.data, normalize=self.normalize,
                                      want_scores=want_scores)
