"""Vocab-parallel embedding and LM head
(python/sglang/srt/layers/vocab_parallel_embedding.py:453-475): each rank holds rows
[vocab_start, vocab_end) of the table; ids outside the shard produce zeros and the partial
results are summed over the TP group."""
import torch
from torch import nn

from ..distributed.parallel_state import (
    get_tensor_model_parallel_rank,
    get_tensor_model_parallel_world_size,
    tensor_model_parallel_all_reduce,
)
from ..ops_backend import get_ops


class VocabParallelEmbedding(nn.Module):
    def __init__(self, num_embeddings: int, embedding_dim: int, dtype, device):
        super().__init__()
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        self.num_embeddings = num_embeddings
        self.tp_size = tp
        # pad the vocabulary so that it splits evenly (the reference pads to a multiple of 64*tp)
        per = (num_embeddings + tp - 1) // tp
        self.vocab_start = rank * per
        self.vocab_end = min(num_embeddings, (rank + 1) * per)
        self.num_embeddings_per_partition = per
        self.weight = nn.Parameter(torch.zeros(per, embedding_dim, dtype=dtype, device=device),
                                   requires_grad=False)

    def shard_from_checkpoint(self, w: torch.Tensor) -> torch.Tensor:
        out = torch.zeros(self.num_embeddings_per_partition, w.shape[1], dtype=w.dtype, device=w.device)
        n = self.vocab_end - self.vocab_start
        if n > 0:
            out[:n] = w[self.vocab_start:self.vocab_end]
        return out

    def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        out = get_ops().embedding(input_ids, self.weight, self.vocab_start,
                                  self.vocab_start + self.num_embeddings_per_partition)
        if self.tp_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        return out


class ParallelLMHead(VocabParallelEmbedding):
    """Output projection sharing the embedding's layout (weight [vocab/tp, hidden])."""
