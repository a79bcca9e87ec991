"""Linear-probe heads -- mirrors reference src/models/evaluator.py:135-147 (LogisticRegressMultiHeadEvaluator);
the unused KNNOnlineEvaluator is not provided."""
import torch.nn as nn


class LogisticRegressMultiHeadEvaluator(nn.Module):
    def __init__(self, feat_dim, n_classes_per_label):
        super().__init__()
        self.classifier = nn.ModuleList([nn.Linear(feat_dim, i) for i in n_classes_per_label])
        for head in self.classifier:
            head.weight.data.normal_(mean=0.0, std=0.01)
            head.bias.data.zero_()

    def forward(self, x):
        return [classify(x) for classify in self.classifier]
