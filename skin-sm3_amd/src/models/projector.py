"""Per-label projectors of the multi-label heads -- the reference's src/models/projector.py surface.  `MultiLabelProjector4`
(one biased Linear per label, projector.py:65-78) is what run.sh:39-56 trains with; variants 1-3 (deeper per-label MLPs,
projector.py:5-62) are parameter containers only here: the native head path (sm3hip/mlc.py) builds v4."""
import torch.nn as nn


class MultiLabelProjector4(nn.Module):
    def __init__(self, in_dim, proj_dim, num_labels):
        super().__init__()
        self.projectors = nn.ModuleList([nn.Sequential(nn.Linear(in_dim, proj_dim)) for _ in range(num_labels)])

    def forward(self, x):
        return [projector(x) for projector in self.projectors]
